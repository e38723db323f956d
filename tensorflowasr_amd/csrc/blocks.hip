// Conformer block kernels for gfx950: fused two-GEMM chains (FFModule, ConvModule tail), LN-prologue
// GEMMs (QKV, pw_conv_1+GLU, out-projection+residual, CTC project, CTC head+argmax), attention core,
// depthwise conv.  Reference semantics: asr/models/conformer_blocks.py:107-265,
// asr/models/layers/multihead_attention.py:151-188 (see DESIGN.md for the kernel <-> reference map).
#include "common.h"
#include "launch.h"

// =====================================================================================================
// chain2: y = res + scale * ( act( pro(x) W1 + b1 ) W2 + b2 )      [optionally followed by LayerNorm]
//   MODE 0 (FFModule, conformer_blocks.py:126-134): pro = LayerNorm, act = swish
//   MODE 1 (ConvModule tail, :214-218): pro = identity (x = depthwise output), act = swish(BN-affine(.)),
//           W1 = SeparableConv1D pointwise kernel, W2 = pw_conv_2
// The hidden activation never leaves registers: GEMM1's accumulator fragment is GEMM2's operand fragment.
// =====================================================================================================
template <int D, int HT, int RT, int CT1, int MODE>
__global__ __launch_bounds__(BLOCK_THREADS) void chain2_kernel(Chain2Args a) {
  constexpr int KB = D / 16;
  static_assert(HT % CT1 == 0, "hidden tiles must split evenly");
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4;
  const int t = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if ((size_t)wid * RT * 16 >= (size_t)a.M) return;

  int tok[RT];
  size_t row[RT];
  f32x4 xs[RT][KB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    tok[rt] = (wid * RT + rt) * 16 + t;
    row[rt] = (size_t)min(tok[rt], a.M - 1) * D;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[rt][kb] = ldg4(a.x + row[rt] + 16 * kb + g4);
  }
  if (MODE == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ln_apply<KB>(xs[rt], a.ln_g, a.ln_b, g4, a.eps);
  }

  f32x4 acc2[RT][KB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < KB; ++nt) acc2[rt][nt] = splat4(0.f);

  const f32x4* __restrict__ w1 = reinterpret_cast<const f32x4*>(a.w1p) + lane;
  const f32x4* __restrict__ w2 = reinterpret_cast<const f32x4*>(a.w2p) + lane;

#pragma unroll 1
  for (int hc = 0; hc < HT / CT1; ++hc) {
    const int h0 = hc * CT1;
    f32x4 acc1[RT][CT1];
#pragma unroll
    for (int nt = 0; nt < CT1; ++nt) {
      f32x4 b = ldg4(a.b1 + 16 * (h0 + nt) + g4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc1[rt][nt] = b;
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int nt = 0; nt < CT1; ++nt) {
        f32x4 w = w1[(size_t)(kb * HT + h0 + nt) * 64];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt][nt] = mma_kblock(w, xs[rt][kb], acc1[rt][nt]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < CT1; ++nt) {
      if (MODE == 1) {
        f32x4 s = ldg4(a.aff_s + 16 * (h0 + nt) + g4);
        f32x4 sh = ldg4(a.aff_t + 16 * (h0 + nt) + g4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt][nt] = swish4(acc1[rt][nt] * s + sh);
      } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt][nt] = swish4(acc1[rt][nt]);
      }
    }
#pragma unroll
    for (int n1 = 0; n1 < CT1; ++n1) {
#pragma unroll
      for (int n2 = 0; n2 < KB; ++n2) {
        f32x4 w = w2[(size_t)((h0 + n1) * KB + n2) * 64];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc2[rt][n2] = mma_kblock(w, acc1[rt][n1], acc2[rt][n2]);
      }
    }
  }

#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const size_t rrow = (size_t)min(tok[rt], a.M - 1) * D;
#pragma unroll
    for (int nt = 0; nt < KB; ++nt) {
      f32x4 r = ldg4(a.res + rrow + 16 * nt + g4);
      f32x4 b = ldg4(a.b2 + 16 * nt + g4);
      acc2[rt][nt] = r + splat4(a.scale) * (acc2[rt][nt] + b);
    }
    if (a.fln_g != nullptr) ln_apply<KB>(acc2[rt], a.fln_g, a.fln_b, g4, a.eps);
    if (tok[rt] < a.M) {
#pragma unroll
      for (int nt = 0; nt < KB; ++nt) stg4(a.y + rrow + 16 * nt + g4, acc2[rt][nt]);
    }
  }
}

template <int D, int HT, int MODE>
static void launch_chain2_t(const Chain2Args& a, hipStream_t s) {
  constexpr int CT1 = (HT % 6 == 0) ? 6 : 4;
  const int tiles = (a.M + 15) / 16;
  // two row tiles per wave halve the weight traffic per MFMA; only worth it when the grid still
  // covers all 1024 SIMDs at least twice.
  if (tiles >= 4096) {
    int waves = (tiles + 1) / 2;
    hipLaunchKernelGGL((chain2_kernel<D, HT, 2, CT1, MODE>), dim3((waves + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  } else {
    hipLaunchKernelGGL((chain2_kernel<D, HT, 1, CT1, MODE>), dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  }
}

int launch_chain2(int D, int mode, const Chain2Args& a, hipStream_t s) {
  if (D == 144 && mode == 0) launch_chain2_t<144, 36, 0>(a, s);
  else if (D == 144 && mode == 1) launch_chain2_t<144, 18, 1>(a, s);
  else if (D == 256 && mode == 0) launch_chain2_t<256, 64, 0>(a, s);
  else if (D == 256 && mode == 1) launch_chain2_t<256, 32, 1>(a, s);
  else return -1;
  return 0;
}

// =====================================================================================================
// gemm_rows: y = epi( pro(x) W + b ) with the whole input row (K = D) resident in registers.
//   EPI_BIAS      y = v                                   (CTC project  conformer_blocks.py:420)
//   EPI_RESIDUAL  y = res + v                             (MHA out-projection + residual :168-169)
//   EPI_QKV       y = v, first `qtiles` column tiles scaled by qscale (query /= sqrt(hs), mha.py:157-158)
//   EPI_GLU       y[c] = v[c] * sigmoid(v[c + N/2])       (pw_conv_1 + GLU :211-212, :18-21)
//   EPI_HEAD      optional logits store + per-token argmax, first max wins (fully_connected + greedy)
// =====================================================================================================
template <int D, int RT, int CT, int EPI, bool LN>
__global__ __launch_bounds__(BLOCK_THREADS) void gemm_rows_kernel(GemmArgs a) {
  constexpr int KB = D / 16;
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4;
  const int t = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if ((size_t)wid * RT * 16 >= (size_t)a.M) return;

  int tok[RT];
  f32x4 xs[RT][KB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    tok[rt] = (wid * RT + rt) * 16 + t;
    const size_t row = (size_t)min(tok[rt], a.M - 1) * D;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[rt][kb] = ldg4(a.x + row + 16 * kb + g4);
    if (LN) ln_apply<KB>(xs[rt], a.ln_g, a.ln_b, g4, a.eps);
  }

  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + lane;
  const int NT = a.NT;  // total column tiles in the packed weight (multiple of CT; GLU: of 2*CT... see host)

  float best_v[RT];
  int best_i[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { best_v[rt] = -INFINITY; best_i[rt] = 0x7fffffff; }

  if (EPI == EPI_GLU) {
    const int half = NT / 2;
#pragma unroll 1
    for (int c0 = blockIdx.y * CT; c0 < half; c0 += gridDim.y * CT) {
      f32x4 acc_a[RT][CT], acc_b[RT][CT];
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        f32x4 ba = ldg4(a.bias + 16 * (c0 + i) + g4);
        f32x4 bb = ldg4(a.bias + 16 * (half + c0 + i) + g4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { acc_a[rt][i] = ba; acc_b[rt][i] = bb; }
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          f32x4 wa = wp[(size_t)(kb * NT + c0 + i) * 64];
          f32x4 wb = wp[(size_t)(kb * NT + half + c0 + i) * 64];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc_a[rt][i] = mma_kblock(wa, xs[rt][kb], acc_a[rt][i]);
            acc_b[rt][i] = mma_kblock(wb, xs[rt][kb], acc_b[rt][i]);
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (tok[rt] < a.M) {
#pragma unroll
          for (int i = 0; i < CT; ++i) {
            f32x4 va = acc_a[rt][i], vb = acc_b[rt][i];
            f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z),
                       va.w * fast_sigmoid(vb.w)};
            stg4(a.y + (size_t)tok[rt] * a.ldy + 16 * (c0 + i) + g4, o);
          }
        }
      }
    }
    return;
  }

#pragma unroll 1
  for (int c0 = blockIdx.y * CT; c0 < NT; c0 += gridDim.y * CT) {
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      f32x4 b = ldg4(a.bias + 16 * (c0 + i) + g4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][i] = b;
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        f32x4 w = wp[(size_t)(kb * NT + c0 + i) * 64];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mma_kblock(w, xs[rt][kb], acc[rt][i]);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool live = tok[rt] < a.M;
      const size_t orow = (size_t)min(tok[rt], a.M - 1) * a.ldy;
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int f0 = 16 * (c0 + i) + g4;  // first of this lane's 4 features
        f32x4 v = acc[rt][i];
        if (EPI == EPI_RESIDUAL) v += ldg4(a.res + (size_t)min(tok[rt], a.M - 1) * a.ldy + f0);
        if (EPI == EPI_QKV) {
          if (c0 + i < a.qtiles) v *= splat4(a.qscale);
        }
        if (EPI == EPI_HEAD) {
          // running argmax in increasing feature order, strict '>' => first maximum wins
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (f0 + j < a.n_valid && vv[j] > best_v[rt]) { best_v[rt] = vv[j]; best_i[rt] = f0 + j; }
          }
          if (a.y != nullptr && live) {
            if (f0 + 3 < a.n_valid && (a.ldy & 3) == 0) {
              stg4(a.y + orow + f0, v);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (f0 + j < a.n_valid) a.y[orow + f0 + j] = vv[j];
            }
          }
        } else if (live) {
          if (f0 + 3 < a.n_valid) stg4(a.y + orow + f0, v);
        }
      }
    }
  }

  if (EPI == EPI_HEAD) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float bv = best_v[rt];
      int bi = best_i[rt];
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        float ov = __shfl_xor(bv, off);
        int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane < 16 && tok[rt] < a.M) {
        a.argmax_out[tok[rt]] = bi;
        if (a.maxval_out != nullptr) a.maxval_out[tok[rt]] = bv;
      }
    }
  }
}

template <int D, int CT, int EPI, bool LN>
static void launch_gemm_rows_t(const GemmArgs& a, int ychunks, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  dim3 grid((tiles + 3) / 4, ychunks);
  hipLaunchKernelGGL((gemm_rows_kernel<D, 1, CT, EPI, LN>), grid, dim3(BLOCK_THREADS), 0, s, a);
}

int launch_gemm_rows(int D, int epi, bool ln, const GemmArgs& a, hipStream_t s) {
  // column-chunk width CT per (D, epilogue); the packed weight's NT is padded to a multiple of it on the host
  // (gemm_ct() below is the single source of truth for that padding).
  const int ct = gemm_ct(D, epi);
  const int chunks = (epi == EPI_GLU) ? (a.NT / 2) / ct : a.NT / ct;
  const int ych = (epi == EPI_HEAD) ? 1 : chunks;
#define GO(DD, CT, EPI, LNF) launch_gemm_rows_t<DD, CT, EPI, LNF>(a, ych, s)
  if (D == 144) {
    if (epi == EPI_BIAS && !ln) GO(144, 9, EPI_BIAS, false);
    else if (epi == EPI_RESIDUAL && !ln) GO(144, 9, EPI_RESIDUAL, false);
    else if (epi == EPI_QKV && ln) GO(144, 9, EPI_QKV, true);
    else if (epi == EPI_GLU && ln) GO(144, 3, EPI_GLU, true);
    else if (epi == EPI_HEAD && !ln) GO(144, 12, EPI_HEAD, false);
    else return -1;
  } else if (D == 256) {
    if (epi == EPI_BIAS && !ln) GO(256, 8, EPI_BIAS, false);
    else if (epi == EPI_RESIDUAL && !ln) GO(256, 8, EPI_RESIDUAL, false);
    else if (epi == EPI_QKV && ln) GO(256, 8, EPI_QKV, true);
    else if (epi == EPI_GLU && ln) GO(256, 4, EPI_GLU, true);
    else if (epi == EPI_HEAD && !ln) GO(256, 12, EPI_HEAD, false);
    else return -1;
  } else {
    return -1;
  }
#undef GO
  return 0;
}

// =====================================================================================================
// attention core: ctx[b, t, h*HS + i] = sum_m softmax_m( q[b,t,h,:] . k[b,m,h,:] ) v[b,m,h,i]
// (multihead_attention.py:160-180; q arrives pre-scaled; no mask, no positional term).
// One wave = 16 queries of one (b, h).  Scores are produced transposed (S^T = K Q^T) so that after exp
// the accumulator fragment *is* the P operand of the P.V MFMA; keys are swept in blocks of 16*KT with an
// online softmax, so any T works.
// =====================================================================================================
template <int HS, int KT>
__global__ __launch_bounds__(BLOCK_THREADS) void attention_kernel(AttnArgs a) {
  constexpr int FB = HS / 16;          // full 16-wide feature blocks
  constexpr int TS = (HS % 16) / 4;    // tail k-steps (feature = 16*FB + 4*ts + g)
  constexpr int OT = (HS + 15) / 16;   // output feature tiles
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int qt = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int T = a.T;
  if (qt * 16 >= T) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = a.ld;  // row stride of the qkv buffer (3*D)
  const int D = a.D;
  const float* __restrict__ base = a.qkv + (size_t)b * T * ld + h * HS;

  const int tq = qt * 16 + c;
  const float* qrow = base + (size_t)min(tq, T - 1) * ld;
  f32x4 q4[FB > 0 ? FB : 1];
  float qs[TS > 0 ? TS : 1];
#pragma unroll
  for (int s = 0; s < FB; ++s) q4[s] = ldg4(qrow + 16 * s + g4);
#pragma unroll
  for (int s = 0; s < TS; ++s) qs[s] = qrow[16 * FB + 4 * s + g];

  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = splat4(0.f);
  float m_run = -INFINITY, l_run = 0.f;

#pragma unroll 1
  for (int k0 = 0; k0 < T; k0 += 16 * KT) {
    f32x4 sc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      sc[kt] = splat4(0.f);
      if (k0 + 16 * kt < T) {   // wave-uniform
        const int tk = min(k0 + 16 * kt + c, T - 1);
        const float* krow = base + D + (size_t)tk * ld;
#pragma unroll
        for (int s = 0; s < FB; ++s) sc[kt] = mma_kblock(ldg4(krow + 16 * s + g4), q4[s], sc[kt]);
#pragma unroll
        for (int s = 0; s < TS; ++s) sc[kt] = mfma4(krow[16 * FB + 4 * s + g], qs[s], sc[kt]);
      }
    }
    // lane holds S^T[key = k0 + 16*kt + 4*g + j][query c]
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int kb = k0 + 16 * kt + g4;
      sc[kt].x = (kb + 0 < T) ? sc[kt].x : -INFINITY;
      sc[kt].y = (kb + 1 < T) ? sc[kt].y : -INFINITY;
      sc[kt].z = (kb + 2 < T) ? sc[kt].z : -INFINITY;
      sc[kt].w = (kb + 3 < T) ? sc[kt].w : -INFINITY;
      mx = fmaxf(mx, fmaxf(fmaxf(sc[kt].x, sc[kt].y), fmaxf(sc[kt].z, sc[kt].w)));
    }
    mx = group_max(mx);
    const float m_new = fmaxf(m_run, mx);          // finite: every block has >= 1 valid key
    const float alpha = __expf(m_run - m_new);      // first block: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      sc[kt].x = __expf(sc[kt].x - m_new);
      sc[kt].y = __expf(sc[kt].y - m_new);
      sc[kt].z = __expf(sc[kt].z - m_new);
      sc[kt].w = __expf(sc[kt].w - m_new);
      psum += (sc[kt].x + sc[kt].y) + (sc[kt].z + sc[kt].w);
    }
    l_run = l_run * alpha + psum;   // per-lane partial; the 4 groups are summed once at the end
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < OT; ++i) o[i] *= splat4(alpha);
    // O^T[i][query] += V^T[i][key] * P^T[key][query]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (k0 + 16 * kt < T) {   // wave-uniform
        const int kb = k0 + 16 * kt + g4;
        const float* v0 = base + 2 * D + (size_t)min(kb + 0, T - 1) * ld;
        const float* v1 = base + 2 * D + (size_t)min(kb + 1, T - 1) * ld;
        const float* v2 = base + 2 * D + (size_t)min(kb + 2, T - 1) * ld;
        const float* v3 = base + 2 * D + (size_t)min(kb + 3, T - 1) * ld;
#pragma unroll
        for (int i = 0; i < OT; ++i) {
          const int f = 16 * i + c;
          const bool ok = (HS % 16 == 0) || (f < HS);
          const int fc = ok ? f : 0;
          float a0 = v0[fc], a1 = v1[fc], a2 = v2[fc], a3 = v3[fc];
          if (!ok) { a0 = a1 = a2 = a3 = 0.f; }
          o[i] = mfma4(a0, sc[kt].x, o[i]);
          o[i] = mfma4(a1, sc[kt].y, o[i]);
          o[i] = mfma4(a2, sc[kt].z, o[i]);
          o[i] = mfma4(a3, sc[kt].w, o[i]);
        }
      }
    }
  }
  const float inv = 1.0f / group_sum(l_run);
  if (tq < T) {
    float* orow = a.ctx + ((size_t)b * T + tq) * D + h * HS;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      if (16 * i + g4 < HS) stg4(orow + 16 * i + g4, o[i] * splat4(inv));
    }
  }
}

int launch_attention(int HS, const AttnArgs& a, hipStream_t s) {
  const int qtiles = (a.T + 15) / 16;
  dim3 grid((qtiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, a.H, a.B);
  if (HS == 36) hipLaunchKernelGGL((attention_kernel<36, 16>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else if (HS == 64) hipLaunchKernelGGL((attention_kernel<64, 16>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else return -1;
  return 0;
}

// =====================================================================================================
// depthwise conv along time (SeparableConv1D depthwise half, conformer_blocks.py:194-197):
//   y[b,t,c] = sum_j u[b, t + j - pad_left, c] * wd[j][c],  zero outside [0,T).
// pad_left = (K-1)/2 for Keras 'same' (K=32 -> 15 left / 16 right), K-1 for 'causal'.
// HBM-bound elementwise kernel: each thread owns 4 channels x TT consecutive frames, window in registers.
// =====================================================================================================
template <int K, int TT>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
  const int c4n = a.D / 4;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int tchunks = (a.T + TT - 1) / TT;
  const int total = a.B * tchunks * c4n;
  if (idx >= total) return;
  const int c4 = (idx % c4n) * 4;
  const int tc = (idx / c4n) % tchunks;
  const int b = idx / (c4n * tchunks);
  const int t0 = tc * TT;
  const float* __restrict__ ub = a.u + (size_t)b * a.T * a.D + c4;
  f32x4 win[TT + K - 1];
#pragma unroll
  for (int i = 0; i < TT + K - 1; ++i) {
    const int tt = t0 + i - a.pad_left;
    win[i] = (tt >= 0 && tt < a.T) ? ldg4(ub + (size_t)tt * a.D) : splat4(0.f);
  }
  f32x4 acc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) acc[i] = splat4(0.f);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const f32x4 w = ldg4(a.wd + (size_t)j * a.D + c4);
#pragma unroll
    for (int i = 0; i < TT; ++i) acc[i] += win[i + j] * w;
  }
  float* yb = a.y + (size_t)b * a.T * a.D + c4;
#pragma unroll
  for (int i = 0; i < TT; ++i)
    if (t0 + i < a.T) stg4(yb + (size_t)(t0 + i) * a.D, acc[i]);
}

int launch_dwconv(int K, const DwArgs& a, hipStream_t s) {
  constexpr int TT = 8;
  const int total = a.B * ((a.T + TT - 1) / TT) * (a.D / 4);
  dim3 grid((total + 255) / 256);
  if (K == 32) hipLaunchKernelGGL((dwconv_kernel<32, TT>), grid, dim3(256), 0, s, a);
  else if (K == 5) hipLaunchKernelGGL((dwconv_kernel<5, TT>), grid, dim3(256), 0, s, a);
  else return -1;
  return 0;
}
