// Conformer block kernels for gfx950: fused two-GEMM chains (FFModule, ConvModule tail), LN-prologue
// GEMMs (QKV, pw_conv_1+GLU, out-projection+residual, CTC project, CTC head+argmax), attention core,
// depthwise conv.  Reference semantics: asr/models/conformer_blocks.py:107-265,
// asr/models/layers/multihead_attention.py:151-188 (see DESIGN.md for the kernel <-> reference map).
#include <cstdlib>

#include "common.h"
#include "env.h"
#include "launch.h"

// =====================================================================================================
// chain2: y = res + scale * ( act( pro(x) W1 + b1 ) W2 + b2 )      [optionally followed by LayerNorm]
//   MODE 0 (FFModule, conformer_blocks.py:126-134): pro = LayerNorm, act = swish
//   MODE 1 (ConvModule tail, :214-218): pro = identity (x = depthwise output), act = swish(BN-affine(.)),
//           W1 = SeparableConv1D pointwise kernel, W2 = pw_conv_2
// The hidden activation never leaves registers: GEMM1's accumulator fragment is GEMM2's operand fragment.
// =====================================================================================================
template <int D, int HT, int NSPLIT, int CT1, int MODE, int RT>
__global__ __launch_bounds__(BLOCK_THREADS, ((RT == 1 && D <= 144) ? 2 : 1)) void chain2_kernel(Chain2Args a) {
  // Work split: a group of RT 16-token tiles is shared by NSPLIT waves of one block, each owning HT/NSPLIT hidden
  // tiles (both GEMMs are sliced along the hidden dimension, so the waves never exchange activations; only the
  // [16*RT x D] partial outputs are summed through LDS at the end).
  // Weight stream: fragments are consumed in batches (one k-block of GEMM1 = CT1 fragments, one hidden tile
  // of GEMM2 = KB fragments).  Batch s+1 is loaded into the other register buffer before the MFMAs of batch
  // s are issued; sched_barrier(0) pins that order (left alone, hipcc sinks every load to its first use and
  // waits vmcnt(0) before each 4-MFMA group).  With RT = 2 every weight fragment feeds two token tiles, which
  // halves the L2 -> CU weight stream per flop (the limiter of the RT = 1 form: ~9 TB/s at 46 % MFMA busy).
  constexpr int KB = D / 16;
  constexpr int HW = HT / NSPLIT;           // hidden tiles per wave
  constexpr int NCH = HW / CT1;             // chunks of CT1 hidden tiles
  constexpr int TPB = WAVES_PER_BLOCK / NSPLIT;   // tile groups per block
  constexpr int NBUF = (CT1 > KB) ? CT1 : KB;
  constexpr int SPC = KB + CT1;             // pipeline steps per chunk
  constexpr int S = NCH * SPC;
  static_assert(HT % NSPLIT == 0 && HW % CT1 == 0 && WAVES_PER_BLOCK % NSPLIT == 0, "bad split");
  static_assert(NCH == 1, "one chunk per wave (keeps the pipeline loop fully unrollable)");
  __shared__ f32x4 red[WAVES_PER_BLOCK][RT][KB][64];

  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4;
  const int t = lane & 15;
  const int wave = threadIdx.x >> 6;
  const int part = wave % NSPLIT;
  const int grp = blockIdx.x * TPB + wave / NSPLIT;   // group of RT token tiles
  const int tiles = (a.M + 15) / 16;
  const int hbase = part * HW;

  int tok[RT];
  size_t row[RT];
  f32x4 xs[RT][KB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    tok[rt] = min(grp * RT + rt, tiles - 1) * 16 + t;
    if (grp * RT + rt >= tiles) tok[rt] = a.M;           // phantom tile: computed on clamped rows, never stored
    row[rt] = (size_t)min(tok[rt], a.M - 1) * D;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[rt][kb] = ldg4(a.x + row[rt] + 16 * kb + g4);
  }

  const f32x4* __restrict__ w1 = reinterpret_cast<const f32x4*>(a.w1p) + lane;
  const f32x4* __restrict__ w2 = reinterpret_cast<const f32x4*>(a.w2p) + lane;
  f32x4 wb[2][NBUF];
  f32x4 acc1[RT][CT1], acc2[RT][KB], b1v[CT1], aff_s[CT1], aff_t[CT1];

  // prefetch of pipeline step s (s is a compile-time constant after unrolling)
  auto prefetch = [&](int s, f32x4(&dst)[NBUF]) {
    const int r = s % SPC;
    const int h0 = hbase;
    if (r < KB) {
#pragma unroll
      for (int i = 0; i < CT1; ++i) dst[i] = w1[(size_t)((r * HT + h0) + i) * 64];
      if (r == 0) {   // bias of this chunk rides with its first batch
#pragma unroll
        for (int i = 0; i < CT1; ++i) b1v[i] = ldg4(a.b1 + 16 * (h0 + i) + g4);
      }
    } else {
      const int n1 = r - KB;
#pragma unroll
      for (int n2 = 0; n2 < KB; ++n2) dst[n2] = w2[(size_t)((h0 + n1) * KB + n2) * 64];
      if (MODE == 1 && n1 == 0) {
#pragma unroll
        for (int i = 0; i < CT1; ++i) {
          aff_s[i] = ldg4(a.aff_s + 16 * (h0 + i) + g4);
          aff_t[i] = ldg4(a.aff_t + 16 * (h0 + i) + g4);
        }
      }
    }
  };

  prefetch(0, wb[0]);
  if (MODE == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ln_apply<KB>(xs[rt], a.ln_g, a.ln_b, g4, a.eps);
  }

#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int r = s % SPC;
    if (s + 1 < S) prefetch(s + 1, wb[(s + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    if (r < KB) {
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < CT1; ++i)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc1[rt][i] = b1v[i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < CT1; ++i)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc1[rt][i] = mfma4(wb[s & 1][i][j], xs[rt][r][j], acc1[rt][i]);
    } else {
      const int n1 = r - KB;
      if (n1 == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int i = 0; i < CT1; ++i)
            acc1[rt][i] = (MODE == 1) ? swish4(acc1[rt][i] * aff_s[i] + aff_t[i]) : swish4(acc1[rt][i]);
          // acc2 starts life here (xs is dead by now, so the allocator can hand its registers over)
#pragma unroll
          for (int nt = 0; nt < KB; ++nt) acc2[rt][nt] = splat4(0.f);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n2 = 0; n2 < KB; ++n2)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc2[rt][n2] = mfma4(wb[s & 1][n2][j], acc1[rt][n1][j], acc2[rt][n2]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // the epilogue wave fetches residual + bias before the barrier so that their latency overlaps the wait
  f32x4 rres[RT][KB], rb2[KB];
  if (part == 0) {
#pragma unroll
    for (int nt = 0; nt < KB; ++nt) {
      rb2[nt] = ldg4(a.b2 + 16 * nt + g4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) rres[rt][nt] = ldg4(a.res + row[rt] + 16 * nt + g4);
    }
  }
  if (NSPLIT > 1) {
    if (part != 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < KB; ++nt) red[wave][rt][nt][lane] = acc2[rt][nt];
    }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int p = 1; p < NSPLIT; ++p)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < KB; ++nt) acc2[rt][nt] += red[wave + p][rt][nt][lane];
  }

#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int nt = 0; nt < KB; ++nt) acc2[rt][nt] = rres[rt][nt] + splat4(a.scale) * (acc2[rt][nt] + rb2[nt]);
    if (a.fln_g != nullptr) ln_apply<KB>(acc2[rt], a.fln_g, a.fln_b, g4, a.eps);
    if (tok[rt] < a.M) {
#pragma unroll
      for (int nt = 0; nt < KB; ++nt) stg4(a.y + row[rt] + 16 * nt + g4, acc2[rt][nt]);
    }
  }
}

template <int D, int HT, int NSPLIT, int CT1, int MODE, bool HAS_RT2>
static void launch_chain2_t(const Chain2Args& a, int rt, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  constexpr int TPB = WAVES_PER_BLOCK / NSPLIT;
  if (HAS_RT2 && rt == 2) {
    const int groups = (tiles + 1) / 2;
    hipLaunchKernelGGL((chain2_kernel<D, HT, NSPLIT, CT1, MODE, (HAS_RT2 ? 2 : 1)>), dim3((groups + TPB - 1) / TPB),
                       dim3(BLOCK_THREADS), 0, s, a);
  } else {
    hipLaunchKernelGGL((chain2_kernel<D, HT, NSPLIT, CT1, MODE, 1>), dim3((tiles + TPB - 1) / TPB), dim3(BLOCK_THREADS),
                       0, s, a);
  }
}

int launch_chain2(int D, int mode, const Chain2Args& a_in, hipStream_t s) {
  const Chain2Args& a = a_in;
  // two token tiles per wave once there are enough tiles to keep every CU busy with tile pairs
  const int tiles = (a.M + 15) / 16;
  const int rt = tiles >= 512 ? 2 : 1;
  if (D == 144 && mode == 0) launch_chain2_t<144, 36, 4, 9, 0, true>(a, rt, s);
  else if (D == 144 && mode == 1) launch_chain2_t<144, 18, 2, 9, 1, true>(a, rt, s);
  else if (D == 256 && mode == 0) launch_chain2_t<256, 64, 4, 16, 0, false>(a, 1, s);
  else if (D == 256 && mode == 1) launch_chain2_t<256, 32, 4, 8, 1, false>(a, 1, s);
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
// (two row tiles per wave do not fit 256 registers: with a two-waves bound hipcc spilled 16-24 B per lane to scratch,
// which also slows the dispatches around the kernel -- profiles/r02_ring_experiments.md)
__global__ __launch_bounds__(BLOCK_THREADS, (RT == 1 ? 2 : 1)) void gemm_rows_kernel(GemmArgs a) {
  constexpr int KB = D / 16;
  constexpr int NF = (EPI == EPI_GLU) ? 2 * CT : CT;   // weight fragments per k-block batch
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4;
  const int t = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if ((size_t)wid * RT * 16 >= (size_t)a.M) return;

  int tok[RT];
  bool live[RT];
  f32x4 xs[RT][KB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    tok[rt] = (wid * RT + rt) * 16 + t;
    live[rt] = tok[rt] < a.M;
    const size_t row = (size_t)min(tok[rt], a.M - 1) * D;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[rt][kb] = ldg4(a.x + row + 16 * kb + g4);
  }

  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + lane;
  const int NT = a.NT;                                  // column tiles in the packed weight
  const int half = NT / 2;                              // GLU: gate tiles start here
  const int NTC = (EPI == EPI_GLU) ? half : NT;         // tiles this kernel sweeps in chunks of CT
  const int cstep = gridDim.y * CT;

  // batch (chunk c0, k-block kb): CT value fragments (+ CT gate fragments for GLU)
  auto fetch = [&](int c0, int kb, f32x4(&dst)[NF]) {
#pragma unroll
    for (int i = 0; i < CT; ++i) dst[i] = wp[(size_t)(kb * NT + c0 + i) * 64];
    if (EPI == EPI_GLU) {
#pragma unroll
      for (int i = 0; i < CT; ++i) dst[CT + i] = wp[(size_t)(kb * NT + half + c0 + i) * 64];
    }
  };

  f32x4 wb[2][NF];
  int c0 = blockIdx.y * CT;
  fetch(c0, 0, wb[0]);
  if (LN) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ln_apply<KB>(xs[rt], a.ln_g, a.ln_b, g4, a.eps);
  }

  float best_v[RT];
  int best_i[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { best_v[rt] = -INFINITY; best_i[rt] = 0x7fffffff; }

#pragma unroll 1
  for (; c0 < NTC; c0 += cstep) {
    f32x4 acc[RT][NF];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const f32x4 bv = ldg4(a.bias + 16 * (c0 + i) + g4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][i] = bv;
      if (EPI == EPI_GLU) {
        const f32x4 bg = ldg4(a.bias + 16 * (half + c0 + i) + g4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][CT + i] = bg;
      }
    }
    const int cn = (c0 + cstep < NTC) ? c0 + cstep : c0;   // next chunk (clamped: the extra fetch is harmless)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (kb + 1 < KB) fetch(c0, kb + 1, wb[(kb + 1) & 1]);
      else fetch(cn, 0, wb[(kb + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(wb[kb & 1][i][j], xs[rt][kb][j], acc[rt][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KB & 1) {   // odd number of steps per chunk: move the prefetched batch back to buffer 0
#pragma unroll
      for (int i = 0; i < NF; ++i) wb[0][i] = wb[1][i];
    }

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (EPI == EPI_GLU) {
        if (live[rt]) {
#pragma unroll
          for (int i = 0; i < CT; ++i) {
            const f32x4 va = acc[rt][i], vb = acc[rt][CT + i];
            f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z),
                       va.w * fast_sigmoid(vb.w)};
            stg4(a.y + (size_t)tok[rt] * a.ldy + 16 * (c0 + i) + g4, o);
          }
        }
      } else {
        const size_t orow = (size_t)min(tok[rt], a.M - 1) * a.ldy;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int f0 = 16 * (c0 + i) + g4;   // first of this lane's 4 features
          f32x4 v = acc[rt][i];
          if (EPI == EPI_RESIDUAL) v += ldg4(a.res + orow + f0);
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
            if (a.y != nullptr && live[rt]) {
              if (f0 + 3 < a.n_valid && (a.ldy & 3) == 0) {
                stg4(a.y + orow + f0, v);
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (f0 + j < a.n_valid) a.y[orow + f0 + j] = vv[j];
              }
            }
          } else if (live[rt]) {
            if (f0 + 3 < a.n_valid) stg4(a.y + orow + f0, v);
          }
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
      if (lane < 16 && live[rt]) {
        a.argmax_out[tok[rt]] = bi;
        if (a.maxval_out != nullptr) a.maxval_out[tok[rt]] = bv;
      }
    }
  }
}

template <int D, int CT, int EPI, bool LN, bool HAS_RT2>
static void launch_gemm_rows_t(const GemmArgs& a, int ychunks, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  // two token tiles per wave halve the L2 weight stream per flop, but at the benchmark shape these short
  // kernels (13-40 us) are bound by ramp-up/tail, and the RT = 2 form measured 10-50 % slower (qkv 40 -> 44 us,
  // out-projection 15 -> 24 us); it pays only with several waves per SIMD left over, i.e. very large batches.
  const int rt = (tiles / 2) * ychunks >= 8192 ? 2 : 1;
  if (HAS_RT2 && rt == 2) {
    dim3 grid(((tiles + 1) / 2 + 3) / 4, ychunks);
    hipLaunchKernelGGL((gemm_rows_kernel<D, (HAS_RT2 ? 2 : 1), CT, EPI, LN>), grid, dim3(BLOCK_THREADS), 0, s, a);
  } else {
    dim3 grid((tiles + 3) / 4, ychunks);
    hipLaunchKernelGGL((gemm_rows_kernel<D, 1, CT, EPI, LN>), grid, dim3(BLOCK_THREADS), 0, s, a);
  }
}

int launch_gemm_rows(int D, int epi, bool ln, const GemmArgs& a, hipStream_t s) {
  // column-chunk width CT per (D, epilogue); the packed weight's NT is padded to a multiple of it on the host
  // (gemm_ct() below is the single source of truth for that padding).
  const int ct = gemm_ct(D, epi);
  const int chunks = (epi == EPI_GLU) ? (a.NT / 2) / ct : a.NT / ct;
  const int ych = (epi == EPI_HEAD) ? 1 : chunks;
#define GO(DD, CT, EPI, LNF) launch_gemm_rows_t<DD, CT, EPI, LNF, (DD == 144 && EPI != EPI_HEAD)>(a, ych, s)
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
// LDSW (round 4, band attention: chunk_conformer_blocks.py:158-176): the four waves of a workgroup are 64 consecutive queries
// of one (b, h), whose bands overlap -- win_front + win_back + 64 keys in all.  The workgroup stages that window of K and V
// once, with 16-byte loads, and every wave takes its fragments from LDS.  Straight from L2 a wave issued ~63 vector loads per
// 64-key block, 48 of them 4 bytes per lane (V is read down its columns); the memory pipe takes one wave instruction per ~15
// cycles whatever its width (tools/ubench/l2_pull.hip), and twelve waves per CU made that 10 of the kernel's 19 us.
constexpr int ATT_WROWS = 128;      // key rows of a staged window
template <int HS, int KT, bool LDSW = false>
__global__ __launch_bounds__(BLOCK_THREADS, ((HS > 36 && KT >= 16) ? 1 : 2)) void attention_kernel(AttnArgs a) {   // <64, 16> spilled at 256 registers
  constexpr int FB = HS / 16;          // full 16-wide feature blocks
  constexpr int TS = (HS % 16) / 4;    // tail k-steps (feature = 16*FB + 4*ts + g)
  constexpr int OT = (HS + 15) / 16;   // output feature tiles
  constexpr int KG = (KT >= 4) ? 4 : KT;   // key tiles per K-fragment batch
  constexpr int VG = (KT >= 2) ? 2 : 1;    // key tiles per V-fragment batch
  constexpr int NKB = KT / KG, NVB = KT / VG;
  static_assert(NKB == 1 || NKB % 2 == 0, "buffer parity must repeat per key block");
  static_assert(NVB == 1 || NVB % 2 == 0, "buffer parity must repeat per key block");
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int qt = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int T = a.Tk, TQ = a.Tq;       // keys / queries per utterance
  if (!LDSW && qt * 16 >= TQ) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = a.ldk;  // row stride of the k / v buffers
  const int D = a.D;
  const float* __restrict__ kbase = a.k + (size_t)b * T * ld + h * HS;
  const float* __restrict__ vbase = a.v + (size_t)b * T * ld + h * HS;
  __shared__ __attribute__((aligned(16))) float kw[LDSW ? ATT_WROWS * HS : 4], vw[LDSW ? ATT_WROWS * HS : 4];
  int wbeg = 0;                        // first key row of the staged window
  if constexpr (LDSW) {
    static_assert(HS % 4 == 0, "rows are staged in 16-byte pieces");
    // the window starts where the first query tile's sweep starts; ALL ATT_WROWS rows are staged (rows past T: the last row --
    // finite values: a masked key's probability is an exact 0, and 0 x garbage must stay 0); the launcher guarantees that
    // every key some query of the workgroup can see lies inside
    const int i0 = a.q_off + (int)blockIdx.x * (16 * WAVES_PER_BLOCK);
    const int lo0 = min(max(i0 - a.win_front, 0), T - a.win_back);
    wbeg = (max(lo0, 0) / 16) * 16;
    constexpr int RQ = HS / 4;
    for (int i = threadIdx.x; i < ATT_WROWS * RQ; i += BLOCK_THREADS) {
      const int r = i / RQ, q = i - r * RQ;
      const size_t src = (size_t)min(wbeg + r, T - 1) * ld + 4 * q;
      *reinterpret_cast<f32x4*>(kw + r * HS + 4 * q) = ldg4(kbase + src);
      *reinterpret_cast<f32x4*>(vw + r * HS + 4 * q) = ldg4(vbase + src);
    }
    __syncthreads();
    if (qt * 16 >= TQ) return;
  }

  const int tq = qt * 16 + c;
  const float* qrow = a.q + ((size_t)b * TQ + min(tq, TQ - 1)) * a.ldq + h * HS;
  f32x4 q4[FB > 0 ? FB : 1];
  float qs[TS > 0 ? TS : 1];
#pragma unroll
  for (int s = 0; s < FB; ++s) q4[s] = ldg4(qrow + 16 * s + g4);
#pragma unroll
  for (int s = 0; s < TS; ++s) qs[s] = qrow[16 * FB + 4 * s + g];

  // fragment buffers: K batches of KG key tiles, V batches of VG key tiles, double buffered.  The next batch is
  // always in flight while the MFMAs of the current one issue (sched_barrier pins the order).
  f32x4 kb4[2][KG][FB > 0 ? FB : 1];
  float kbs[2][KG][TS > 0 ? TS : 1];
  float vb[2][VG][OT][4];

  auto load_k = [&](int k0, int batch, int buf) {
#pragma unroll
    for (int tt = 0; tt < KG; ++tt) {
      const int tk = min(k0 + 16 * (batch * KG + tt) + c, T - 1);
      if constexpr (LDSW) {
        const float* krow = kw + min(tk - wbeg, ATT_WROWS - 1) * HS;          // (rows past the window: masked below, any value does)
#pragma unroll
        for (int s = 0; s < FB; ++s) kb4[buf][tt][s] = *reinterpret_cast<const f32x4*>(krow + 16 * s + g4);
#pragma unroll
        for (int s = 0; s < TS; ++s) kbs[buf][tt][s] = krow[16 * FB + 4 * s + g];
      } else {
        const float* krow = kbase + (size_t)tk * ld;
#pragma unroll
        for (int s = 0; s < FB; ++s) kb4[buf][tt][s] = ldg4(krow + 16 * s + g4);
#pragma unroll
        for (int s = 0; s < TS; ++s) kbs[buf][tt][s] = krow[16 * FB + 4 * s + g];
      }
    }
  };
  auto load_v = [&](int k0, int batch, int buf) {
#pragma unroll
    for (int tt = 0; tt < VG; ++tt) {
      const int kb = k0 + 16 * (batch * VG + tt) + g4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* vrow = LDSW ? vw + min(min(kb + j, T - 1) - wbeg, ATT_WROWS - 1) * HS : vbase + (size_t)min(kb + j, T - 1) * ld;
#pragma unroll
        for (int i = 0; i < OT; ++i) {
          const int f = 16 * i + c;
          const bool ok = (HS % 16 == 0) || (f < HS);
          const float v = vrow[ok ? f : 0];
          vb[buf][tt][i][j] = ok ? v : 0.f;
        }
      }
    }
  };

  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = splat4(0.f);
  float m_run = -INFINITY, l_run = 0.f;

  // visible key range of this lane's query, and the key blocks the query tile has to sweep
  const bool band = a.win_front >= 0;
  int klo = 0, khi = T - 1, kbeg = 0, kend = T;
  if (band) {
    const int iq = a.q_off + min(tq, TQ - 1);
    klo = min(max(iq - a.win_front, 0), T - a.win_back);
    khi = max(min(iq + a.win_back, T), a.win_back);
    const int i0 = a.q_off + qt * 16, i1 = a.q_off + min(qt * 16 + 15, TQ - 1);
    const int lo0 = min(max(i0 - a.win_front, 0), T - a.win_back);        // lo() and hi() are non-decreasing in i
    const int hi1 = min(max(min(i1 + a.win_back, T), a.win_back), T - 1);
    kbeg = (max(lo0, 0) / 16) * 16;
    kend = hi1 + 1;
  }

  load_k(kbeg, 0, 0);
#pragma unroll 1
  for (int k0 = kbeg; k0 < kend; k0 += 16 * KT) {
    f32x4 sc[KT];
    // ---- S^T = K Q^T, KG key tiles per step
#pragma unroll
    for (int s = 0; s < NKB; ++s) {
      if (s + 1 < NKB) load_k(k0, s + 1, (s + 1) & 1);
      else load_v(k0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // k-step-major issue order: consecutive MFMAs go to different key tiles (independent accumulators)
#pragma unroll
      for (int tt = 0; tt < KG; ++tt) sc[s * KG + tt] = splat4(0.f);
#pragma unroll
      for (int f = 0; f < FB; ++f) {
#pragma unroll
        for (int tt = 0; tt < KG; ++tt) sc[s * KG + tt] = mfma4(kb4[s & 1][tt][f].x, q4[f].x, sc[s * KG + tt]);
#pragma unroll
        for (int tt = 0; tt < KG; ++tt) sc[s * KG + tt] = mfma4(kb4[s & 1][tt][f].y, q4[f].y, sc[s * KG + tt]);
#pragma unroll
        for (int tt = 0; tt < KG; ++tt) sc[s * KG + tt] = mfma4(kb4[s & 1][tt][f].z, q4[f].z, sc[s * KG + tt]);
#pragma unroll
        for (int tt = 0; tt < KG; ++tt) sc[s * KG + tt] = mfma4(kb4[s & 1][tt][f].w, q4[f].w, sc[s * KG + tt]);
      }
#pragma unroll
      for (int f = 0; f < TS; ++f)
#pragma unroll
        for (int tt = 0; tt < KG; ++tt) sc[s * KG + tt] = mfma4(kbs[s & 1][tt][f], qs[f], sc[s * KG + tt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane holds S^T[key = k0 + 16*kt + 4*g + j][query c]
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int kb = k0 + 16 * kt + g4;
      // keys outside [klo, khi] (band mask; Keras adds -1e9 there, i.e. exp() == 0 in fp32) or beyond T
      sc[kt].x = (kb + 0 < T && kb + 0 >= klo && kb + 0 <= khi) ? sc[kt].x : -INFINITY;
      sc[kt].y = (kb + 1 < T && kb + 1 >= klo && kb + 1 <= khi) ? sc[kt].y : -INFINITY;
      sc[kt].z = (kb + 2 < T && kb + 2 >= klo && kb + 2 <= khi) ? sc[kt].z : -INFINITY;
      sc[kt].w = (kb + 3 < T && kb + 3 >= klo && kb + 3 <= khi) ? sc[kt].w : -INFINITY;
      mx = fmaxf(mx, fmaxf(fmaxf(sc[kt].x, sc[kt].y), fmaxf(sc[kt].z, sc[kt].w)));
    }
    mx = group_max(mx);
    // a key block can be entirely outside one query's band: keep the running max finite-or-zero so that
    // exp(-inf - m) is 0 instead of exp(-inf + inf) = NaN
    const float m_cand = fmaxf(m_run, mx);
    const float m_new = (m_cand == -INFINITY) ? 0.f : m_cand;
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
    m_run = m_cand;
#pragma unroll
    for (int i = 0; i < OT; ++i) o[i] *= splat4(alpha);
    // ---- O^T[i][query] += V^T[i][key] * P^T[key][query], VG key tiles per step
    const int k0n = (k0 + 16 * KT < kend) ? k0 + 16 * KT : k0;
#pragma unroll
    for (int s = 0; s < NVB; ++s) {
      if (s + 1 < NVB) load_v(k0, s + 1, (s + 1) & 1);
      else load_k(k0n, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < VG; ++tt) {
        const f32x4 p = sc[s * VG + tt];
#pragma unroll
        for (int i = 0; i < OT; ++i) o[i] = mfma4(vb[s & 1][tt][i][0], p.x, o[i]);
#pragma unroll
        for (int i = 0; i < OT; ++i) o[i] = mfma4(vb[s & 1][tt][i][1], p.y, o[i]);
#pragma unroll
        for (int i = 0; i < OT; ++i) o[i] = mfma4(vb[s & 1][tt][i][2], p.z, o[i]);
#pragma unroll
        for (int i = 0; i < OT; ++i) o[i] = mfma4(vb[s & 1][tt][i][3], p.w, o[i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float inv = 1.0f / group_sum(l_run);
  if (tq < TQ) {
    float* orow = a.ctx + ((size_t)b * TQ + tq) * D + h * HS;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      if (16 * i + g4 < HS) stg4(orow + 16 * i + g4, o[i] * splat4(inv));
    }
  }
}

template <int HS>
static void launch_attention_t(const AttnArgs& a, hipStream_t s) {
  const int qtiles = (a.Tq + 15) / 16;
  dim3 grid((qtiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, a.H, a.B);
  // keys are swept in blocks of 16*KT; short sequences (streaming blocks: T = 13) and band attention
  // (win_front + win_back + 16 keys per query tile) use small blocks
  const int span = a.win_front >= 0 ? min(a.Tk, a.win_front + a.win_back + 31) : a.Tk;
  // band attention whose window for 64 queries fits ATT_WROWS key rows: K / V staged once per workgroup (MI355ASR_ATTN_BAND_LDS=0: from L2)
  static const bool band_lds = mi355_env("MI355ASR_ATTN_BAND_LDS", 1) != 0;
  if (span <= 16) hipLaunchKernelGGL((attention_kernel<HS, 1>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else if (band_lds && HS % 4 == 0 && HS <= 36 && a.win_front >= 0 && span <= 96 && a.win_front + a.win_back + 16 * WAVES_PER_BLOCK + 15 <= ATT_WROWS)
    hipLaunchKernelGGL((attention_kernel<HS, 4, true>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else if (span <= 96) hipLaunchKernelGGL((attention_kernel<HS, 4>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else hipLaunchKernelGGL((attention_kernel<HS, 16>), grid, dim3(BLOCK_THREADS), 0, s, a);
}

// would launch_attention hand this launch to the two-term attention_split_kernel (the one kernel that reads head-major operands)?
bool attention_takes_head_major(int HS, const AttnArgs& a) {
  static const bool lds_env = mi355_env("MI355ASR_ATTN_LDS", 1) != 0;
  static const bool split_env = mi355_env("MI355ASR_ATTN_SPLIT", 1) != 0;
  return lds_env && split_env && attention_split_two_term(HS, a);
}

int launch_attention(int HS, const AttnArgs& a, hipStream_t s) {
  // short full-attention utterances (offline ConformerCTC): K / V^T staged in LDS (attention_lds.hip)
  // MI355ASR_ATTN_SPLIT=0: the fp32-MFMA LDS kernel of round 1 instead of the split-bf16 one (attention_split.hip);
  // MI355ASR_ATTN_LDS=0: neither (online-softmax kernel with K / V from L2)
  static const bool lds_env = mi355_env("MI355ASR_ATTN_LDS", 1) != 0;
  static const bool split_env = mi355_env("MI355ASR_ATTN_SPLIT", 1) != 0;
  if (lds_env && split_env && attention_split_applicable(HS, a))
    return launch_attention_split(HS, a, s);
  if (lds_env && split_env && attention_split64_applicable(HS, a))      // round 5: head size 64, operand bounds known, <= 288 keys
    return launch_attention_split64(HS, a, s);
  if (a.head_major) return -1;         // head-major q / k / v (round 5) are read by attention_split_kernel only
  note_scheme(SCHEME_F32);
  if (lds_env && attention_lds_applicable(HS, a)) return launch_attention_lds(HS, a, s);
  if (HS == 36) launch_attention_t<36>(a, s);
  else if (HS == 64) launch_attention_t<64>(a, s);
  else {
    // round 6: the other head sizes a dmodel of 144 / 256 / 512 factors into (multiples of four up to 128): the online-softmax kernel
    // with one key-block size each (slow path: a configuration outside the shipped YAMLs must run, not fail)
    const int qtiles = (a.Tq + 15) / 16;
    const dim3 grid((qtiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, a.H, a.B), blk(BLOCK_THREADS);
    switch (HS) {
      case 12: hipLaunchKernelGGL((attention_kernel<12, 4>), grid, blk, 0, s, a); break;
      case 16: hipLaunchKernelGGL((attention_kernel<16, 4>), grid, blk, 0, s, a); break;
      case 24: hipLaunchKernelGGL((attention_kernel<24, 4>), grid, blk, 0, s, a); break;
      case 32: hipLaunchKernelGGL((attention_kernel<32, 4>), grid, blk, 0, s, a); break;
      case 48: hipLaunchKernelGGL((attention_kernel<48, 4>), grid, blk, 0, s, a); break;
      case 72: hipLaunchKernelGGL((attention_kernel<72, 2>), grid, blk, 0, s, a); break;
      case 128: hipLaunchKernelGGL((attention_kernel<128, 1>), grid, blk, 0, s, a); break;
      default: return -1;
    }
  }
  return 0;
}
bool attention_head_size_ok(int HS) { return HS == 36 || HS == 64 || HS == 12 || HS == 16 || HS == 24 || HS == 32 || HS == 48 || HS == 72 || HS == 128; }

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

// Any kernel size (round 6: conformer_blocks.py:278-294 takes any; a configuration the tuned kernels do not cover must run, not
// fail): one thread per (frame, four channels), the taps walked at run time.  Slow path: every tap is two L2 / L1 reads.
__global__ __launch_bounds__(256) void dwconv_any_kernel(DwArgs a, int K) {
  const int c4n = a.D / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)a.B * a.T * c4n) return;
  const int c4 = (int)(idx % c4n) * 4;
  const int t = (int)((idx / c4n) % a.T);
  const int b = (int)(idx / ((size_t)c4n * a.T));
  const float* __restrict__ ub = a.u + (size_t)b * a.T * a.D + c4;
  f32x4 acc = splat4(0.f);
  for (int j = 0; j < K; ++j) {
    const int tt = t + j - a.pad_left;
    if (tt >= 0 && tt < a.T) acc += ldg4(ub + (size_t)tt * a.D) * ldg4(a.wd + (size_t)j * a.D + c4);
  }
  stg4(a.y + ((size_t)b * a.T + t) * a.D + c4, acc);
}

// LDS-tiled depthwise conv: a workgroup owns 64 consecutive frames of one utterance x CB channels; the 64 + K - 1 input
// rows it needs are staged once (coalesced, zero-filled outside the utterance), then each thread computes 8 frames x 4
// channels from a register window read from LDS.  The version above re-reads its 39-row window per thread straight from
// L2 (4.9x read amplification, 39 dependent-latency loads per thread): 15 us at 64 x 250 x 144; this one is bound by the
// 9 MB in + 9 MB out.
template <int K, int CB>
__global__ __launch_bounds__(320) void dwconv_tile_kernel(DwArgs a) {
  constexpr int TT = 64, TS = 8, C4 = CB / 4, ROWS = TT + K - 1;
  __shared__ __attribute__((aligned(16))) float tile[ROWS * CB];
  __shared__ __attribute__((aligned(16))) float wt[K * CB];        // the K x CB taps of this channel block
  const int b = blockIdx.y, t0 = blockIdx.x * TT, c0 = blockIdx.z * CB;
  const float* __restrict__ ub = a.u + (size_t)b * a.T * a.D + c0;
  // all global loads first, then the LDS writes (a load -> write loop costs one L2 round trip per iteration)
  constexpr int NT = ((CB / 4) * 8 + 63) / 64 * 64, NL = (ROWS * C4 + NT - 1) / NT, NW = (K * C4 + NT - 1) / NT;
  f32x4 stage[NL], wstage[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int i = min((int)threadIdx.x + k * NT, K * C4 - 1);
    const int j = i / C4, c = i - j * C4;
    wstage[k] = ldg4(a.wd + (size_t)j * a.D + c0 + 4 * c);
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int i = threadIdx.x + k * NT;
    const int r = i / C4, c = i - r * C4;
    const int tt = t0 + r - a.pad_left;
    stage[k] = (i < ROWS * C4 && tt >= 0 && tt < a.T) ? ldg4(ub + (size_t)tt * a.D + 4 * c) : splat4(0.f);
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < ROWS * C4) *reinterpret_cast<f32x4*>(&tile[4 * i]) = stage[k];
  }
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < K * C4) *reinterpret_cast<f32x4*>(&wt[4 * i]) = wstage[k];
  }
  __syncthreads();
  const int c = threadIdx.x % C4, tg = threadIdx.x / C4;
  if (tg >= TT / TS) return;
  // taps in groups of GJ: per group GJ weights + TS + GJ - 1 window rows in registers (the whole TS + K - 1 row window
  // of K = 32 is 156 registers and spilled to scratch under the 256-register cap of a 320-thread workgroup; a kernel
  // with scratch also slows the launches around it).  The group loop is NOT unrolled, its body is.
  constexpr int GJ = (K % 8 == 0) ? 8 : K;
  const float* __restrict__ trow = &tile[(tg * TS) * CB + 4 * c];
  const float* __restrict__ wrow = &wt[4 * c];
  f32x4 acc[TS];
#pragma unroll
  for (int i = 0; i < TS; ++i) acc[i] = splat4(0.f);
#pragma unroll 1
  for (int j0 = 0; j0 < K; j0 += GJ) {
    f32x4 win[TS + GJ - 1], w[GJ];
#pragma unroll
    for (int r = 0; r < TS + GJ - 1; ++r) win[r] = *reinterpret_cast<const f32x4*>(trow + (j0 + r) * CB);
#pragma unroll
    for (int j = 0; j < GJ; ++j) w[j] = *reinterpret_cast<const f32x4*>(wrow + (j0 + j) * CB);
#pragma unroll
    for (int j = 0; j < GJ; ++j)
#pragma unroll
      for (int i = 0; i < TS; ++i) acc[i] += win[i + j] * w[j];
  }
  float* yb = a.y + (size_t)b * a.T * a.D + c0 + 4 * c;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    const int t = t0 + tg * TS + i;
    if (t < a.T) stg4(yb + (size_t)t * a.D, acc[i]);
  }
}

template <int K, int CB>
static int launch_dwconv_tile(const DwArgs& a, hipStream_t s) {
  const int threads = ((CB / 4) * 8 + 63) / 64 * 64;
  hipLaunchKernelGGL((dwconv_tile_kernel<K, CB>), dim3((a.T + 63) / 64, a.B, a.D / CB), dim3(threads), 0, s, a);
  return 0;
}

int launch_dwconv(int K, const DwArgs& a, hipStream_t s) {
  if (a.T * a.B >= 2048) {                  // LDS-tiled kernel; below: the first-generation kernel (window re-read from L2 per thread)
    if (K == 32 && a.D == 144) return launch_dwconv_tile<32, 144>(a, s);
    if (K == 32 && a.D % 128 == 0) return launch_dwconv_tile<32, 128>(a, s);
    if (K == 5 && a.D % 128 == 0) return launch_dwconv_tile<5, 128>(a, s);
    if (K == 5 && a.D == 144) return launch_dwconv_tile<5, 144>(a, s);
  }
  constexpr int TT = 8;
  const int total = a.B * ((a.T + TT - 1) / TT) * (a.D / 4);
  dim3 grid((total + 255) / 256);
  if (K == 32) hipLaunchKernelGGL((dwconv_kernel<32, TT>), grid, dim3(256), 0, s, a);
  else if (K == 5) hipLaunchKernelGGL((dwconv_kernel<5, TT>), grid, dim3(256), 0, s, a);
  else if (K >= 1) {
    const size_t n = (size_t)a.B * a.T * (a.D / 4);
    hipLaunchKernelGGL(dwconv_any_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, K);
  } else return -1;
  return 0;
}
