// bf16-MFMA variant of the dense layers (BASELINE config 3: "StreamingConformerCTC ... bf16 MFMA"): bf16 inputs to
// every GEMM (weights rounded once on the host, activations rounded to nearest-even at the operand), fp32
// accumulation, fp32 LayerNorm / softmax / swish / GLU / residuals -- the split SURVEY 8d prescribes for that config.
//
// One generic kernel: Y[M, N] = epilogue( LN?(X)[M, K] . W[K, N] + b ),  any K that is a multiple of 16.
// Same transposed-product layout as the fp32 kernels (common.h), with v_mfma_f32_16x16x16_bf16: the lane that held
// four fp32 k-steps of a token (k = 16 kb + 4 g + j) holds the same four values as one packed bf16x4 operand, so
// one MFMA (16 cycles) replaces four v_mfma_f32_16x16x4_f32 (128 cycles) per (k-block, column tile), and the
// weight fragments are half the bytes ("P16" order, 4 bf16 per lane).
// A wave owns 16 rows x CT column tiles; X is streamed from global/L2 per k-block (not held in registers), so K is
// free (the 5120-wide subsampling Dense and the 1024-wide FFN hidden use the same code).  The streaming shapes this
// exists for are tiny (832 rows): columns are split over grid.y so that every launch has several hundred waves,
// and the layers of a block run as separate launches (the hidden activation goes through HBM: 3.4 MB).
#include "common.h"
#include "env.h"
#include "launch.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

DEV s16x4 to_bf16x4(f32x4 v) {      // 2 x v_cvt_pk_bf16_f32 (round to nearest even)
  f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
  u32x2 p = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t)),
             __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_t))};
  return __builtin_bit_cast(s16x4, p);
}
DEV f32x4 mfma_bf16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// operand precision of the kernel: bf16 (one 16x16x16 MFMA per k-block and column tile) or fp32 (four 16x16x4 MFMAs;
// the layer-at-a-time path for dmodel values the fused / chained fp32 kernels are not instantiated for, e.g. 512)
struct PBf16 {
  typedef s16x4 W;
  typedef s16x4 X;
  static DEV X cvt(f32x4 v) { return to_bf16x4(v); }
  static DEV f32x4 mma(W w, X x, f32x4 acc) { return mfma_bf16(w, x, acc); }
};
struct PF32 {
  typedef f32x4 W;
  typedef f32x4 X;
  static DEV X cvt(f32x4 v) { return v; }
  static DEV f32x4 mma(W w, X x, f32x4 acc) { return mma_kblock(w, x, acc); }
};

// NF weight fragments per k-block: CT column tiles (+ CT gate tiles for GLU).
// KS = 4: the four waves of a workgroup share one row tile and split K between them (partial sums meet in LDS, wave 0
// runs the epilogue) -- for the streaming shapes, where 52 row tiles would otherwise be 52 long serial waves.
// KS = 1: one row tile per wave, as everywhere else.
// The k-loop fetches U k-blocks (operand + weight fragments) at a time, one group ahead.
template <class P, int CT, int EPI, bool LN, int KS>
__global__ __launch_bounds__(BLOCK_THREADS, ((sizeof(typename P::W) > 8 && CT == 9 && LN && KS == 4) ? 1 : 2)) void gemm16_kernel(Gemm16Args a) {   // that one instantiation spilled at 256 registers
  typedef typename P::W WF;
  typedef typename P::X XF;
  constexpr int NF = (EPI == E16_GLU) ? 2 * CT : CT;
  constexpr int U0 = NF <= 4 ? 4 : (NF <= 8 ? 2 : 1);
  constexpr int U = (sizeof(WF) > 8 && U0 > 1) ? U0 / 2 : U0;     // fp32 fragments are twice the registers
  __shared__ f32x4 red[KS > 1 ? (KS - 1) * NF * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int wave = threadIdx.x >> 6;
  const int wid = KS > 1 ? blockIdx.x : blockIdx.x * WAVES_PER_BLOCK + wave;
  const int ks = KS > 1 ? wave : 0;
  if ((size_t)wid * 16 >= (size_t)a.M) return;          // uniform per workgroup when KS > 1
  const int tok = wid * 16 + c;
  const bool live = tok < a.M;
  const int rowi = min(tok, a.M - 1);
  const float* __restrict__ xr = (a.rpb > 0 ? a.x + (size_t)(rowi / a.rpb) * a.bstride + (size_t)(rowi % a.rpb) * a.ldx
                                            : a.x + (size_t)rowi * a.ldx) + g4;
  const int KB = a.K / 16, NT = a.NT;
  const int KBs = (KB + KS - 1) / KS;
  const int kbeg = ks * KBs, kend = min(KB, kbeg + KBs);
  const int half = NT / 2;
  const int NTC = (EPI == E16_GLU) ? half : NT;           // tiles swept in chunks of CT
  const WF* __restrict__ wp = reinterpret_cast<const WF*>(a.wp) + lane;

  // prologue LayerNorm statistics (two-pass, biased variance, eps inside the sqrt: Keras semantics)
  float mean = 0.f, rstd = 1.f;
  // (the register-hungry instantiations -- fp32 fragments, more than eight accumulators -- keep the round-1 code: they are not
  // the streaming shapes, and zero scratch is a hard requirement of this library)
  constexpr bool LEAN = sizeof(WF) <= 8 && NF <= 8;
  if (LN && LEAN && KB <= 16) {
    // round 3: rows of up to 256 features (dmodel 256, the streaming shapes) are requested whole -- ONE memory round trip for
    // both passes of the statistics instead of four; at 832 rows the prologue's latency chain was half of the kernel
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ldg4(xr + 16 * min(i, KB - 1));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < KB) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = group_sum(s) / (float)a.K;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < KB) {
        const f32x4 d = v[i] - splat4(mean);
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      }
    rstd = 1.0f / sqrtf(group_sum(q) / (float)a.K + a.eps);
  } else if (LN) {
    // eight row chunks requested at a time: one load per trip costs a memory latency each (32 trips for dmodel 256 --
    // the whole kernel at streaming sizes, where nothing else covers it)
    float s = 0.f;
    for (int kb0 = 0; kb0 < KB; kb0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ldg4(xr + 16 * min(kb0 + i, KB - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (kb0 + i < KB) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    mean = group_sum(s) / (float)a.K;
    float q = 0.f;
    for (int kb0 = 0; kb0 < KB; kb0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ldg4(xr + 16 * min(kb0 + i, KB - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (kb0 + i < KB) {
          const f32x4 d = v[i] - splat4(mean);
          q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    rstd = 1.0f / sqrtf(group_sum(q) / (float)a.K + a.eps);
  }
  auto xfrag = [&](int kb) -> XF {
    f32x4 v = ldg4(xr + 16 * kb);
    if (LN) v = (v - splat4(mean)) * splat4(rstd) * ldg4(a.ln_g + 16 * kb + g4) + ldg4(a.ln_b + 16 * kb + g4);
    return P::cvt(v);
  };

  float best_v = -INFINITY;
  int best_i = 0;
  // EPI_HEAD sweeps every column chunk inside the wave (the argmax needs all classes); the others take the chunk
  // blockIdx.y selects
  const int c_begin = (EPI == E16_HEAD) ? 0 : blockIdx.y * CT;
  const int c_end = (EPI == E16_HEAD) ? NTC : c_begin + CT;
  for (int c0 = c_begin; c0 < c_end; c0 += CT) {
    f32x4 acc[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = splat4(0.f);
    WF wb[2][U][NF];
    XF xb[2][U];
    auto load_group = [&](int kb0, WF (&w)[U][NF], XF (&x)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kb = min(kb0 + u, kend - 1);
        x[u] = xfrag(kb);
#pragma unroll
        for (int i = 0; i < CT; ++i) w[u][i] = wp[(size_t)(kb * NT + c0 + i) * 64];
        if (EPI == E16_GLU) {
#pragma unroll
          for (int i = 0; i < CT; ++i) w[u][CT + i] = wp[(size_t)(kb * NT + half + c0 + i) * 64];
        }
      }
    };
    auto mma_group = [&](int kb0, const WF (&w)[U][NF], const XF (&x)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (kb0 + u < kend) {
#pragma unroll
          for (int i = 0; i < NF; ++i) acc[i] = P::mma(w[u][i], x[u], acc[i]);
        }
      }
    };
    // round 3: the epilogue's operands (bias, residual rows) are requested BEFORE the k-loop -- at streaming sizes every
    // dependent memory round trip is a visible part of the kernel -- and a group past the end is not fetched at all
    f32x4 pbias[NF], pres[EPI == E16_RES ? CT : 1];
    if (LEAN && (KS == 1 || wave == 0)) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int f0 = 16 * (c0 + i) + g4;
        pbias[i] = ldg4(a.bias + f0);
        if (EPI == E16_GLU) pbias[CT + i] = ldg4(a.bias + 16 * half + f0);
        if constexpr (EPI == E16_RES) pres[i] = ldg4(a.res + (size_t)min(tok, a.M - 1) * a.ldy + f0);
      }
    }
    if (kbeg < kend) {
      load_group(kbeg, wb[0], xb[0]);
      for (int kb0 = kbeg; kb0 < kend; kb0 += 2 * U) {
        if (!LEAN || kb0 + U < kend) load_group(kb0 + U, wb[1], xb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(kb0, wb[0], xb[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (!LEAN || kb0 + 2 * U < kend) load_group(kb0 + 2 * U, wb[0], xb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(kb0 + U, wb[1], xb[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (KS > 1) {
      // partial sums of waves 1..KS-1 -> LDS -> wave 0
      if (c0 != c_begin) __syncthreads();                 // the previous chunk's partial sums have been consumed
      if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NF; ++i) red[((wave - 1) * NF + i) * 64 + lane] = acc[i];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w2 = 0; w2 < KS - 1; ++w2)
#pragma unroll
          for (int i = 0; i < NF; ++i) acc[i] += red[(w2 * NF + i) * 64 + lane];
      }
    }
    if (KS > 1 && wave != 0) continue;                    // only wave 0 holds the full sums
    // ---- epilogue of this chunk: lane holds Y[token c][feature 16*(c0+i) + g4 + 0..3]
    float* yrow = a.y ? a.y + (size_t)min(tok, a.M - 1) * a.ldy : nullptr;
    if (EPI == E16_GLU) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int f0 = 16 * (c0 + i) + g4;
        const f32x4 va = acc[i] + (LEAN ? pbias[i] : ldg4(a.bias + f0)), vb = acc[CT + i] + (LEAN ? pbias[CT + i] : ldg4(a.bias + 16 * half + f0));
        f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
        if (live) stg4(yrow + f0, o);
      }
    } else if (EPI == E16_RES) {
      // y = res + scale * (acc + bias), optionally followed by a LayerNorm over the row (needs CT == NT)
      f32x4 v[CT];
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int f0 = 16 * (c0 + i) + g4;
        if constexpr (LEAN) v[i] = pres[i] + splat4(a.scale) * (acc[i] + pbias[i]);
        else v[i] = ldg4(a.res + (size_t)min(tok, a.M - 1) * a.ldy + f0) + splat4(a.scale) * (acc[i] + ldg4(a.bias + f0));
      }
      if (a.fln_g) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mu = group_sum(s) / (float)(16 * CT);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const f32x4 d = v[i] - splat4(mu);
          q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        const float rs = 1.0f / sqrtf(group_sum(q) / (float)(16 * CT) + a.eps);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int f0 = 16 * (c0 + i) + g4;
          v[i] = (v[i] - splat4(mu)) * splat4(rs) * ldg4(a.fln_g + f0) + ldg4(a.fln_b + f0);
        }
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < CT; ++i) stg4(yrow + 16 * (c0 + i) + g4, v[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int f0 = 16 * (c0 + i) + g4;
        f32x4 v = acc[i] + (LEAN ? pbias[i] : ldg4(a.bias + f0));
        if (EPI == E16_SWISH) v = swish4(v);
        if (EPI == E16_AFFSWISH) v = swish4(v * ldg4(a.aff_s + f0) + ldg4(a.aff_t + f0));
        if (EPI == E16_QKV) { if (c0 + i < a.qtiles) v *= splat4(a.qscale); }
        if (EPI == E16_HEAD) {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (f0 + j < a.n_valid && vv[j] > best_v) { best_v = vv[j]; best_i = f0 + j; }   // first maximum wins
          if (yrow && live) {
            if (f0 + 3 < a.n_valid && (a.ldy & 3) == 0) stg4(yrow + f0, v);
            else
#pragma unroll
              for (int j = 0; j < 4; ++j) if (f0 + j < a.n_valid) yrow[f0 + j] = vv[j];
          }
        } else if (live && f0 + 3 < a.n_valid) {
          stg4(yrow + f0, v);
        }
      }
    }
  }
  if (EPI == E16_HEAD && a.argmax_out && (KS == 1 || wave == 0)) {
    // the 4 lane groups of a token hold disjoint feature sets: max over groups, lowest index on ties
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      const float ov = __shfl_xor(best_v, off);
      const int oi = __shfl_xor(best_i, off);
      if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
    }
    if (live && lane < 16) a.argmax_out[tok] = best_i;
  }
}

void launch_layernorm_rows(float* y, const float* g, const float* b, int M, int N, int ld, float eps, hipStream_t s);

template <class P, int CT, int EPI, bool LN>
void go(const Gemm16Args& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  const int ntc = (EPI == E16_GLU) ? a.NT / 2 : a.NT;
  const int ny = EPI == E16_HEAD ? 1 : ntc / CT;
  // fewer than ~2 waves per SIMD with one wave per row tile -> split K over the four waves of a workgroup
  if ((size_t)tiles * ny < 2048) {
    hipLaunchKernelGGL((gemm16_kernel<P, CT, EPI, LN, 4>), dim3(tiles, ny), dim3(BLOCK_THREADS), 0, s, a);
  } else {
    hipLaunchKernelGGL((gemm16_kernel<P, CT, EPI, LN, 1>), dim3((tiles + 3) / 4, ny), dim3(BLOCK_THREADS), 0, s, a);
  }
}

// Column tiles per wave: the full row (CT = NT) when the epilogue needs it (residual + LayerNorm: N = dmodel = 144 /
// 256 / 512) or when there are enough row tiles to fill the chip; otherwise 3 (dmodel 144) / 4 so that small M still
// gives several hundred waves.
template <class P>
int dispatch(int epi, bool ln, const Gemm16Args& a, hipStream_t s) {
  if (a.K % 16 != 0 || a.K < 16) return -1;
  const int tiles = (a.M + 15) / 16;
  const int ntc = (epi == E16_GLU) ? a.NT / 2 : a.NT;
  const bool d144 = (ntc % 9 == 0) || (ntc % 4 != 0 && ntc % 3 == 0);     // column tiles in chunks of 3 / 9
  if (ntc % 4 != 0 && ntc % 3 != 0 && epi != E16_HEAD) return -1;
#define CASE(E, L) \
  if (epi == E && ln == L) { \
    if constexpr (E == E16_RES) { \
      if (ntc == 9) go<P, 9, E, L>(a, s); \
      /* round 3: few rows (the streaming shapes) and no row LayerNorm: column chunks of four tiles -- four times the */ \
      /* workgroups (52 row tiles alone leave 200 CUs idle) on the lean code path */ \
      else if (ntc == 16 && !a.fln_g && tiles < 512) go<P, 4, E, L>(a, s); \
      else if (ntc == 16) go<P, 16, E, L>(a, s); \
      else if (ntc % 8 == 0) { \
        /* wide rows (dmodel 512): column chunks of 8 tiles, the row LayerNorm as a second pass over y */ \
        Gemm16Args b = a; b.fln_g = nullptr; b.fln_b = nullptr; go<P, 8, E, L>(b, s); \
        if (a.fln_g) launch_layernorm_rows(a.y, a.fln_g, a.fln_b, a.M, 16 * ntc, a.ldy, a.eps, s); \
      } else return -1; \
    } else if constexpr (E == E16_HEAD) { if (a.NT % 12 != 0) return -1; go<P, 12, E, L>(a, s); } \
    else if (d144) { if (tiles >= 1024 && ntc % 9 == 0 && E != E16_GLU) go<P, 9, E, L>(a, s); else go<P, 3, E, L>(a, s); } \
    else { if (tiles >= 1024 && ntc % 8 == 0) go<P, 8, E, L>(a, s); else go<P, 4, E, L>(a, s); } \
    return 0; }
  CASE(E16_BIAS, false)
  CASE(E16_SWISH, true)
  CASE(E16_RES, false)
  CASE(E16_QKV, true)
  CASE(E16_GLU, true)
  CASE(E16_AFFSWISH, false)
  CASE(E16_HEAD, false)
#undef CASE
  return -1;
}

// y[row] = LayerNorm(y[row]) in place, one wave per row (two-pass statistics, Keras semantics)
__global__ __launch_bounds__(BLOCK_THREADS) void layernorm_rows_kernel(float* y, const float* __restrict__ g,
                                                                      const float* __restrict__ b, int M, int N, int ld, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  float* p = y + (size_t)row * ld;
  float s = 0.f;
  for (int i = lane * 4; i < N; i += 256) { const f32x4 v = ldg4(p + i); s += (v.x + v.y) + (v.z + v.w); }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)N;
  float q = 0.f;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 d = ldg4(p + i) - splat4(mean);
    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) q += __shfl_xor(q, off);
  const float rstd = 1.0f / sqrtf(q / (float)N + eps);
  for (int i = lane * 4; i < N; i += 256)
    stg4(p + i, (ldg4(p + i) - splat4(mean)) * splat4(rstd) * ldg4(g + i) + ldg4(b + i));
}
void launch_layernorm_rows(float* y, const float* g, const float* b, int M, int N, int ld, float eps, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3((M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK_THREADS), 0, s,
                     y, g, b, M, N, ld, eps);
}

// =====================================================================================================
// chain256 (round 4): a whole FFModule / ConvModule tail of dmodel 256 in ONE launch in bf16 mode,
//     y = res + scale * ( act( pro(x) W1 + b1 ) W2 + b2 )   [+ LayerNorm]            (conformer_blocks.py:126-134, :214-218)
// MODE 0: pro = LayerNorm, act = swish (HT = 64 hidden tiles); MODE 1: pro = identity, act = swish(BN-affine(.)) (HT = 32).
// ONE workgroup of eight waves owns RT 16-token tiles: every wave requests the first batches of its share of W1 at once (the
// weights do not depend on the rows), computes HT / 8 hidden tiles per row tile, leaves them in LDS as bf16 operand fragments
// (GEMM1's accumulator layout IS GEMM2's operand layout: nothing is transposed), and after one barrier computes TWO 16-column
// tiles of the output over the whole hidden dimension -- no partial sums to reduce, the hidden activation (fp32 in HBM
// between the two gemm16 / gemm_ring launches before: 136 MB per FFModule at 16 640 rows) never leaves the CU.  Weight
// fragments travel in batches of eight 1 KB fragments (16 bytes per lane), two batches ahead of their MFMAs (the first batches of
// W2 across the barrier).  Same arithmetic as the two launches: bf16 operands rounded to nearest even, fp32 accumulation; the
// trailing LayerNorm's statistics cross the waves through LDS in a fixed order.
// What bounds it: one workgroup needs all of a module's weights (1 MB for an FFModule) and a CU pulls ~145 GB/s from L2 with
// 16-byte loads (tools/ubench/l2_pull.hip): 7 us + the fixed part of a launch.  Measured 17.0 us per FFModule at 832 rows (two
// gemm16 launches: 12.7 + 10 us on 832 workgroups each), 46 us at 16 640 rows with four row tiles per workgroup (two ring
// launches: 102-133 us); 1 / 2 / 4 row tiles share every weight fragment, chosen by the row count.  Tried at 832 rows and not
// kept: four workgroups per row tile with a quarter of the hidden dimension each, partial outputs summed in workgroup order by
// the one that arrives last at a per-tile counter -- with device-scope fences 27 us (a release walks the L2), with relaxed
// device-scope atomics and hand-counted waits 21 us against 25 for the kernel as it then was: 2 % of the config-3 step for a
// protocol outside the language's memory model.
constexpr int C_NW = 8;         // waves per workgroup, two 16-column tiles of dmodel 256 each
constexpr int C_KB = 16;        // 16-wide k-blocks of a row
constexpr int C_BF = 8;         // weight fragments (1 KB: 16 bytes per lane) per batch
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
DEV f32x4 mfma32_bf16(u32x4_t a, u32x4_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// Weights: the one-term slab-ring packs of gemm_ring.hip (api.hip: put_ring with ring_terms = 1) -- [N / 128 chunks][K / 32
// steps][8 tiles][64 lanes][8 bf16], a lane's eight values of a 32-wide k-step being the two 16-blocks side by side, which is
// how two accumulator-layout tiles (or two operand k-blocks) sit next to each other -- read straight from L2 with 16-byte
// loads: a CU pulls ~145 GB/s that way and half of it with 8-byte loads (tools/ubench/l2_pull.hip; the first version of this
// kernel read the P16 bf16 arena, 8 bytes per lane, 48 loads in flight: 45 GB/s, 23 us per FFModule at 832 rows).
// RT row tiles per workgroup share every weight fragment; HWP hidden tiles per wave and PHASE: with RT = 4 the hidden
// dimension goes through LDS in phases of 8 HWP = 32 tiles (GEMM1 of the phase, barrier, its k-steps of GEMM2, barrier),
// so that accumulators (RT x HWP tiles) and LDS (RT x 32 tiles) stay the size they have at RT = 2.  PF weight batches are in
// flight ahead of the one being multiplied.
DEV unsigned ring_frag(unsigned tile, unsigned step, unsigned steps) { return (((tile >> 3) * steps + step) * 8 + (tile & 7)) * 64; }
template <int HT, int MODE, int RT, int HWP, int PF>
__global__ __launch_bounds__(C_NW * 64) void chain256_bf16_kernel(Chain2Args a) {
  constexpr int PT = C_NW * HWP;      // hidden tiles per phase
  constexpr int NPH = HT / PT;        // phases
  constexpr int G1 = C_BF / HWP;      // 32-wide k-steps per fragment batch of GEMM1
  constexpr int NB1 = (C_KB / 2) / G1, NB2 = (PT / 2) / 4, NBP = NB1 + NB2, NB = NPH * NBP, NBUF = PF + 1;
  static_assert(HT % PT == 0 && C_BF % HWP == 0 && PT % 8 == 0, "phases of whole batches");
  __shared__ __attribute__((aligned(16))) u32x4_t hid[RT][PT / 2][64];      // pairs of hidden tiles: one GEMM2 operand each
  __shared__ __attribute__((aligned(16))) u32x4_t xl[RT > 1 ? RT : 1][RT > 1 ? C_KB / 2 : 1][64];      // RT > 1: the operand rows, converted once by waves 0 .. RT - 1
  __shared__ float stat[2][C_NW][RT][16];
  // unsigned index arithmetic throughout: scalar base + 32-bit vector offset addressing (a signed index costs a sign-extended
  // 64-bit address pair per load)
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned g4 = (lane >> 4) * 4;
  const int c = (int)(lane & 15);
  const u32x4_t* __restrict__ w1 = reinterpret_cast<const u32x4_t*>(a.w1p);
  const u32x4_t* __restrict__ w2 = reinterpret_cast<const u32x4_t*>(a.w2p);
  // 32-bit element offsets (the launcher bounds M): scalar base + one offset register per row tile, nothing to keep in pairs
  auto row_of = [&](int rt) -> unsigned { return (unsigned)min((int)(blockIdx.x * RT + rt) * 16 + c, a.M - 1) * (16u * C_KB); };
  u32x4_t wa[NBUF][C_BF];
  // batch t of the stream: phase t / NBP; inside it NB1 batches of W1 (G1 k-steps for this wave's HWP hidden tiles of the
  // phase), then NB2 batches of W2 (four k-steps = four hidden tile pairs of the phase, for this wave's two column tiles)
  auto load = [&](int t, u32x4_t (&w)[C_BF]) {
    const int ph = t / NBP, r = t % NBP;
    if (r < NB1) {
#pragma unroll
      for (int u = 0; u < G1; ++u)
#pragma unroll
        for (int i = 0; i < HWP; ++i) w[u * HWP + i] = w1[ring_frag(ph * PT + wave * HWP + i, r * G1 + u, C_KB / 2) + lane];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 2; ++j) w[u * 2 + j] = w2[ring_frag(2 * wave + j, ph * (PT / 2) + (r - NB1) * 4 + u, HT / 2) + lane];
    }
  };
  u32x4_t xb[C_KB / 2];
  if (RT == 1 || wave < RT) {
    const unsigned row = row_of(RT == 1 ? 0 : wave);
    f32x4 xr[C_KB];
#pragma unroll
    for (int kb = 0; kb < C_KB; ++kb) xr[kb] = ldg4(a.x + (row + 16u * kb + g4));
    if (RT == 1) {
#pragma unroll
      for (int t = 0; t < PF; ++t) load(t, wa[t]);
    }
    if (MODE == 0) {
      // two-pass statistics of the whole row (Keras semantics), as gemm16_kernel's
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < C_KB; ++i) sm += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
      const float mean = group_sum(sm) / (float)(16 * C_KB);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < C_KB; ++i) {
        const f32x4 d = xr[i] - splat4(mean);
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      }
      const float rstd = 1.0f / sqrtf(group_sum(q) / (float)(16 * C_KB) + a.eps);
#pragma unroll
      for (int i = 0; i < C_KB; ++i) xr[i] = (xr[i] - splat4(mean)) * splat4(rstd) * ldg4(a.ln_g + (16u * i + g4)) + ldg4(a.ln_b + (16u * i + g4));
    }
#pragma unroll
    for (int i = 0; i < C_KB / 2; ++i) {
      const s16x4 lo = to_bf16x4(xr[2 * i]), hi = to_bf16x4(xr[2 * i + 1]);
      const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
      xb[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
    }
    if (RT > 1) {
#pragma unroll
      for (int i = 0; i < C_KB / 2; ++i) xl[wave][i][lane] = xb[i];
    }
  }
  if (RT > 1) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < PF; ++t) load(t, wa[t]);          // (waves 0 .. RT - 1: after their rows, the registers are free then)
    __syncthreads();
  }
  f32x4 acc1[RT][HWP], acc2[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc2[rt][0] = acc2[rt][1] = splat4(0.f);
  f32x4 rres[RT][2], rb2[2];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int ph = t / NBP, r = t % NBP;
    if (t == (RT < 4 ? NB - NB2 : NB - 1)) {
      // the epilogue's operands: requested when the last run of GEMM2 batches starts (RT = 4: its last batch) -- not earlier,
      // they are RT x 8 registers
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rb2[j] = ldg4(a.b2 + (16u * (2 * wave + j) + g4));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rres[rt][j] = ldg4(a.res + (row_of(rt) + 16u * (2 * wave + j) + g4));
      }
    }
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < HWP; ++i) {
        const f32x4 bv = ldg4(a.b1 + (16u * (ph * PT + i) + wave * (16u * HWP) + g4));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt][i] = bv;
      }
    }
    if (t + PF < NB) load(t + PF, wa[(t + PF) % NBUF]);
    __builtin_amdgcn_sched_barrier(0);
    if (r < NB1) {
#pragma unroll
      for (int u = 0; u < G1; ++u) {
        u32x4_t xv[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xv[rt] = RT > 1 ? xl[rt][r * G1 + u][lane] : xb[r * G1 + u];
#pragma unroll
        for (int i = 0; i < HWP; ++i)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc1[rt][i] = mfma32_bf16(wa[t % NBUF][u * HWP + i], xv[rt], acc1[rt][i]);
      }
      if (r == NB1 - 1) {
        // the phase's hidden tiles -> LDS (a later phase: once every wave has read the previous one's), tile 2 p and 2 p + 1
        // side by side: one 16-byte operand of GEMM2
        if (ph > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < HWP; ++i) {
          f32x4 as = splat4(1.f), at = splat4(0.f);
          if (MODE == 1) { as = ldg4(a.aff_s + (16u * (ph * PT + i) + wave * (16u * HWP) + g4)); at = ldg4(a.aff_t + (16u * (ph * PT + i) + wave * (16u * HWP) + g4)); }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            reinterpret_cast<s16x4*>(&hid[rt][(wave * HWP + i) >> 1][lane])[i & 1] = to_bf16x4(swish4(MODE == 1 ? acc1[rt][i] * as + at : acc1[rt][i]));
        }
        __syncthreads();
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x4_t hv = hid[rt][(r - NB1) * 4 + u][lane];
#pragma unroll
          for (int j = 0; j < 2; ++j) acc2[rt][j] = mfma32_bf16(wa[t % NBUF][u * 2 + j], hv, acc2[rt][j]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 v[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 2; ++j) v[rt][j] = rres[rt][j] + splat4(a.scale) * (acc2[rt][j] + rb2[j]);
  if (a.fln_g) {
    // row statistics across the eight waves (each holds 32 of a token's 256 columns), two passes, fixed order
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float sm = ((v[rt][0].x + v[rt][0].y) + (v[rt][0].z + v[rt][0].w)) + ((v[rt][1].x + v[rt][1].y) + (v[rt][1].z + v[rt][1].w));
      sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
      if (lane < 16) stat[0][wave][rt][c] = sm;
    }
    __syncthreads();
    float q[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < C_NW; ++w) tot += stat[0][w][rt][c];
      const float mu = tot / (float)(16 * C_KB);
      q[rt] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[rt][j] = v[rt][j] - splat4(mu);
        q[rt] += (v[rt][j].x * v[rt][j].x + v[rt][j].y * v[rt][j].y) + (v[rt][j].z * v[rt][j].z + v[rt][j].w * v[rt][j].w);
      }
      q[rt] += __shfl_xor(q[rt], 16); q[rt] += __shfl_xor(q[rt], 32);
      if (lane < 16) stat[1][wave][rt][c] = q[rt];
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float qt = 0.f;
#pragma unroll
      for (int w = 0; w < C_NW; ++w) qt += stat[1][w][rt][c];
      const float rs = 1.0f / sqrtf(qt / (float)(16 * C_KB) + a.eps);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        v[rt][j] = v[rt][j] * splat4(rs) * ldg4(a.fln_g + (16u * (2 * wave + j) + g4)) + ldg4(a.fln_b + (16u * (2 * wave + j) + g4));
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if ((int)(blockIdx.x * RT + rt) * 16 + c < a.M) {
#pragma unroll
      for (int j = 0; j < 2; ++j) stg4(a.y + (row_of(rt) + 16u * (2 * wave + j) + g4), v[rt][j]);
    }
  }
}

// =====================================================================================================
// gemm256 (round 6): ONE dense layer of dmodel 256 in bf16 mode at many rows,  Y = epilogue( LN?(X)[M, 256] W[256, N] + b ),
// in chain256's shape: a workgroup of eight waves owns RT 16-token tiles, their rows go through the prologue LayerNorm once and sit
// in LDS as bf16 operand fragments (40 KB at RT = 5), wave w walks the column tiles w, w + 8, ... -- per tile the eight 1 KB
// fragments of its K = 256 from the one-term slab-ring pack, requested a tile ahead, RT MFMAs per fragment -- and stores each
// tile as it finishes.  The ring kernel (gemm_ring.hip) splits the columns over workgroups and re-reads, re-normalises and re-converts
// the rows per 128-column chunk behind a slab DMA ring with a barrier per 32-wide k-step: at K = 256 that is eight steps of
// work under ~40 us of prologue, ring start-up and epilogue (config 3, 16 640 rows: qkv 44, pw_conv_1 + GLU 35, attention out 22,
// CTC projection 26 us).  Same arithmetic: operands rounded to nearest-even bf16, fp32 accumulation along K in 32-wide steps.
// EPI: E16_BIAS, E16_QKV (q columns x qscale), E16_GLU (value tile t with gate tile t + NT / 2), E16_RES (N = 256; optional LayerNorm
// over the output row, statistics across the waves in a fixed order as chain256's).
template <int EPI, bool LN, int RT>
__global__ __launch_bounds__(C_NW * 64) void gemm256_bf16_kernel(Gemm16Args a, const u32x4_t* __restrict__ wr) {
  constexpr int KS = C_KB / 2;                                   // eight 32-wide k-steps
  __shared__ __attribute__((aligned(16))) u32x4_t xl[RT][KS][64];
  __shared__ float stat[2][C_NW][RT][16];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned g4 = (lane >> 4) * 4;
  const int c = (int)(lane & 15);
  auto tok_of = [&](int rt) -> int { return (int)(blockIdx.x * RT + rt) * 16 + c; };
  const int NT = a.NT;                                           // column tiles (GLU: value + gate tiles)
  constexpr bool GLU = EPI == E16_GLU;
  constexpr bool HEAD = EPI == E16_HEAD;
  // what a wave walks: tiles, or (value, gate) pairs; the class head: the ring's whole chunks (tiles past NT are skipped)
  const int units = GLU ? NT / 2 : (HEAD ? (((a.n_valid + 15) / 16 + 7) / 8) * 8 : NT);
  const int per = units / C_NW;                                  // per wave (the launcher guarantees divisibility)
  // fragment of column tile `tile` (GLU: slot 4 + ... holds the gate tiles of a chunk), k-step `st`
  auto frag = [&](int unit, int gate, int st) -> unsigned {
    const unsigned tile = GLU ? (unsigned)((unit >> 2) * 8 + (unit & 3) + 4 * gate) : (unsigned)unit;
    return ring_frag(tile, (unsigned)st, KS) + lane;
  };
  u32x4_t wa[2][GLU ? 2 * KS : KS];
  auto load = [&](int i, u32x4_t (&w)[GLU ? 2 * KS : KS]) {
    const int unit = (int)wave + C_NW * i;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      w[st] = wr[frag(unit, 0, st)];
      if constexpr (GLU) w[KS + st] = wr[frag(unit, 1, st)];
    }
  };
  if (wave < RT) {
    const unsigned row = (unsigned)min(tok_of((int)wave), a.M - 1) * (unsigned)a.ldx;
    f32x4 xr[C_KB];
#pragma unroll
    for (int kb = 0; kb < C_KB; ++kb) xr[kb] = ldg4(a.x + (row + 16u * kb + g4));
    if constexpr (LN) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < C_KB; ++i) sm += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
      const float mean = group_sum(sm) / (float)(16 * C_KB);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < C_KB; ++i) {
        const f32x4 d = xr[i] - splat4(mean);
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      }
      const float rstd = 1.0f / sqrtf(group_sum(q) / (float)(16 * C_KB) + a.eps);
#pragma unroll
      for (int i = 0; i < C_KB; ++i) xr[i] = (xr[i] - splat4(mean)) * splat4(rstd) * ldg4(a.ln_g + (16u * i + g4)) + ldg4(a.ln_b + (16u * i + g4));
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const s16x4 lo = to_bf16x4(xr[2 * i]), hi = to_bf16x4(xr[2 * i + 1]);
      const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
      xl[wave][i][lane] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  load(0, wa[0]);
  __syncthreads();
  f32x4 keep[RT][2];                                             // E16_RES with the row LayerNorm: the wave's two tiles of every row
  float best_v[RT];                                              // E16_HEAD: running arg-max over the classes this lane sees
  int best_i[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { best_v[rt] = -INFINITY; best_i[rt] = 0; }
#pragma unroll 1
  for (int i = 0; i < per; i += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (i + h >= per) break;
      if (i + h + 1 < per) load(i + h + 1, wa[(h + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[RT], accg[GLU ? RT : 1];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { acc[rt] = splat4(0.f); if constexpr (GLU) accg[rt] = splat4(0.f); }
#pragma unroll
      for (int st = 0; st < KS; ++st) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x4_t xv = xl[rt][st][lane];
          acc[rt] = mfma32_bf16(wa[h][st], xv, acc[rt]);
          if constexpr (GLU) accg[rt] = mfma32_bf16(wa[h][KS + st], xv, accg[rt]);
        }
      }
      const int unit = (int)wave + C_NW * (i + h);
      const unsigned f0 = 16u * (unsigned)unit + g4;
      if constexpr (GLU) {
        const f32x4 ba = ldg4(a.bias + f0), bb = ldg4(a.bias + (16u * (unsigned)(NT / 2) + f0));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 va = acc[rt] + ba, vb = accg[rt] + bb;
          const f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
          if (tok_of(rt) < a.M) stg4(a.y + ((size_t)tok_of(rt) * a.ldy + f0), o);
        }
      } else if constexpr (HEAD) {
        if (unit < NT) {                                         // (wave-uniform; the bias is padded to NT tiles)
          const f32x4 bv = ldg4(a.bias + f0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const f32x4 v = acc[rt] + bv;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if ((int)f0 + j < a.n_valid && vv[j] > best_v[rt]) { best_v[rt] = vv[j]; best_i[rt] = (int)f0 + j; }   // first maximum wins
            if (a.y && tok_of(rt) < a.M) {
              float* yrow = a.y + (size_t)tok_of(rt) * a.ldy;
              if ((int)f0 + 3 < a.n_valid) stg4(yrow + f0, v);
              else
#pragma unroll
                for (int j = 0; j < 4; ++j) if ((int)f0 + j < a.n_valid) yrow[f0 + j] = vv[j];
            }
          }
        }
      } else {
        const f32x4 bv = ldg4(a.bias + f0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          f32x4 v = acc[rt] + bv;
          if constexpr (EPI == E16_QKV) { if (unit < a.qtiles) v *= splat4(a.qscale); }
          if constexpr (EPI == E16_RES) v = ldg4(a.res + ((size_t)min(tok_of(rt), a.M - 1) * a.ldy + f0)) + splat4(a.scale) * v;
          if (EPI == E16_RES && a.fln_g) keep[rt][h] = v;
          else if (tok_of(rt) < a.M) stg4(a.y + ((size_t)tok_of(rt) * a.ldy + f0), v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (HEAD) {
    if (a.argmax_out) {
      // a token's classes: four lane groups x eight waves; the larger value, the lower class among equals (as every head here)
      __shared__ int stat_i[C_NW][RT][16];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float bv = best_v[rt];
        int bi = best_i[rt];
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          const float ov = __shfl_xor(bv, off);
          const int oi = __shfl_xor(bi, off);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane < 16) { stat[0][wave][rt][c] = bv; stat_i[wave][rt][c] = bi; }
      }
      __syncthreads();
      if (wave < RT && lane < 16) {
        float bv = stat[0][0][wave][c];
        int bi = stat_i[0][wave][c];
#pragma unroll
        for (int w = 1; w < C_NW; ++w) {
          const float ov = stat[0][w][wave][c];
          const int oi = stat_i[w][wave][c];
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (tok_of((int)wave) < a.M) a.argmax_out[tok_of((int)wave)] = bi;
      }
    }
  }
  if constexpr (EPI == E16_RES) {
    if (a.fln_g) {                                               // (per == 2: the launcher only takes N = 256 with a row LayerNorm)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float sm = ((keep[rt][0].x + keep[rt][0].y) + (keep[rt][0].z + keep[rt][0].w)) + ((keep[rt][1].x + keep[rt][1].y) + (keep[rt][1].z + keep[rt][1].w));
        sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
        if (lane < 16) stat[0][wave][rt][c] = sm;
      }
      __syncthreads();
      float q[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < C_NW; ++w) tot += stat[0][w][rt][c];
        const float mu = tot / (float)(16 * C_KB);
        q[rt] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          keep[rt][j] = keep[rt][j] - splat4(mu);
          q[rt] += (keep[rt][j].x * keep[rt][j].x + keep[rt][j].y * keep[rt][j].y) + (keep[rt][j].z * keep[rt][j].z + keep[rt][j].w * keep[rt][j].w);
        }
        q[rt] += __shfl_xor(q[rt], 16); q[rt] += __shfl_xor(q[rt], 32);
        if (lane < 16) stat[1][wave][rt][c] = q[rt];
      }
      __syncthreads();
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float qt = 0.f;
#pragma unroll
        for (int w = 0; w < C_NW; ++w) qt += stat[1][w][rt][c];
        const float rs = 1.0f / sqrtf(qt / (float)(16 * C_KB) + a.eps);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned f0 = 16u * (wave + C_NW * (unsigned)j) + g4;
          const f32x4 v = keep[rt][j] * splat4(rs) * ldg4(a.fln_g + f0) + ldg4(a.fln_b + f0);
          if (tok_of(rt) < a.M) stg4(a.y + ((size_t)tok_of(rt) * a.ldy + f0), v);
        }
      }
    }
  }
}

__global__ void to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n) {
  for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    const s16x4 v = to_bf16x4(ldg4(src + i));
    *reinterpret_cast<s16x4*>(dst + i) = v;
  }
}

}  // namespace

// whole-arena conversion: a packed fp32 matrix at float offset o is the same fragment order at element offset o
int launch_to_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
  if (n % 4 != 0) return -1;
  hipLaunchKernelGGL(to_bf16_kernel, dim3(1024), dim3(256), 0, s, src, (unsigned short*)dst, n);
  return 0;
}

// FFModule (mode 0) / ConvModule tail (mode 1) of dmodel 256 as one launch; w1p / w2p = the one-term slab-ring packs.  -1: no such kernel
int launch_chain256_bf16(int mode, const Chain2Args& a, hipStream_t s) {
  if (a.M <= 0 || a.M > (1 << 22)) return -1;         // 32-bit element offsets of a row
  note_scheme(SCHEME_BF16);
  const int tiles = (a.M + 15) / 16;
  const dim3 block(C_NW * 64);
  // more row tiles per workgroup as soon as every CU has a workgroup anyway: a half / a quarter of the weight traffic per row
  // (MI355ASR_CHAIN256_RT = 1 / 2 / 4 / 5 forces one)
  static const int rt_env = (int)mi355_env("MI355ASR_CHAIN256_RT", 0);
  int rt = rt_env ? rt_env : (tiles >= 1024 ? 4 : (tiles >= 512 ? 2 : 1));
  // One workgroup per CU (100 KB of LDS at RT = 4), and a workgroup's time is mostly its weight stream: 1 040 tiles (config 3's
  // 64 x 260 rows) are 260 workgroups of four tiles = TWO rounds over 256 CUs, the second with four workgroups.  Five tiles per
  // workgroup (208 workgroups) when that saves a round.
  static const int ncu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
  if (!rt_env && rt == 4 && ((tiles + 3) / 4 + ncu - 1) / ncu > ((tiles + 4) / 5 + ncu - 1) / ncu) rt = 5;
  if (rt == 5) {
    const dim3 grid((tiles + 4) / 5);
    if (mode == 0) hipLaunchKernelGGL((chain256_bf16_kernel<64, 0, 5, 2, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((chain256_bf16_kernel<32, 1, 5, 2, 2>), grid, block, 0, s, a);
  } else if (rt == 4) {
    const dim3 grid((tiles + 3) / 4);
    if (mode == 0) hipLaunchKernelGGL((chain256_bf16_kernel<64, 0, 4, 4, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((chain256_bf16_kernel<32, 1, 4, 4, 2>), grid, block, 0, s, a);
  } else if (rt == 2) {
    const dim3 grid((tiles + 1) / 2);
    if (mode == 0) hipLaunchKernelGGL((chain256_bf16_kernel<64, 0, 2, 8, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((chain256_bf16_kernel<32, 1, 2, 4, 2>), grid, block, 0, s, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL((chain256_bf16_kernel<64, 0, 1, 8, 2>), dim3(tiles), block, 0, s, a);
    else hipLaunchKernelGGL((chain256_bf16_kernel<32, 1, 1, 4, 2>), dim3(tiles), block, 0, s, a);
  }
  return 0;
}
template <int EPI, bool LN>
static int go256(const Gemm16Args& a, const void* ring, hipStream_t s) {
  static const int ncu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
  const int tiles = (a.M + 15) / 16;
  // four row tiles per workgroup, five where that saves a round over the chip (as launch_chain256_bf16)
  const bool five = ((tiles + 3) / 4 + ncu - 1) / ncu > ((tiles + 4) / 5 + ncu - 1) / ncu;
  note_scheme(SCHEME_BF16);
  if (five) hipLaunchKernelGGL((gemm256_bf16_kernel<EPI, LN, 5>), dim3((tiles + 4) / 5), dim3(C_NW * 64), 0, s, a, (const u32x4_t*)ring);
  else hipLaunchKernelGGL((gemm256_bf16_kernel<EPI, LN, 4>), dim3((tiles + 3) / 4), dim3(C_NW * 64), 0, s, a, (const u32x4_t*)ring);
  return 0;
}
// One dense layer of dmodel 256 (K = 256) in bf16 mode from 8 192 rows on; ring = the layer's one-term slab-ring pack.  -1: not this
// kernel's shape (nothing launched).  MI355ASR_GEMM256=0: the ring kernel as before round 6.
int launch_gemm256_bf16(int epi, bool ln, const Gemm16Args& a, const void* ring, hipStream_t s) {
  static const bool on = mi355_env("MI355ASR_GEMM256", 1) != 0;
  static const long min_m = mi355_env("MI355ASR_GEMM256_MIN_M", 8192);
  if (!on || !ring || a.K != 256 || a.M < min_m || a.M > (1 << 22) || a.rpb != 0 || (a.ldx & 3)) return -1;
  if (epi == E16_HEAD) {                                         // logits optional; the ring holds whole chunks of eight tiles
    if (ln || a.NT < 1 || a.n_valid < 1 || a.n_valid > 16 * a.NT || !(a.y || a.argmax_out) || (a.y && (a.ldy & 3))) return -1;
    return go256<E16_HEAD, false>(a, ring, s);
  }
  if (!a.y || (a.ldy & 3)) return -1;
  const int units = epi == E16_GLU ? a.NT / 2 : a.NT;
  if (a.NT <= 0 || units % C_NW != 0 || a.n_valid != (epi == E16_GLU ? 16 * units : 16 * a.NT)) return -1;
  if (epi == E16_GLU && (a.NT / 2) % 4 != 0) return -1;          // the ring holds four value + four gate tiles per chunk
  if (epi != E16_GLU && a.NT % 8 != 0) return -1;
  switch (epi) {
    case E16_BIAS: return ln ? -1 : go256<E16_BIAS, false>(a, ring, s);
    case E16_QKV: return ln ? go256<E16_QKV, true>(a, ring, s) : -1;
    case E16_GLU: return ln ? go256<E16_GLU, true>(a, ring, s) : -1;
    case E16_RES:
      if (ln || (a.fln_g && a.NT != 2 * C_NW)) return -1;
      return go256<E16_RES, false>(a, ring, s);
    default: return -1;
  }
}
int launch_gemm16_bf16(int epi, bool ln, const Gemm16Args& a, hipStream_t s) { note_scheme(SCHEME_BF16); return dispatch<PBf16>(epi, ln, a, s); }
// same kernel with fp32 operands: wp = the fp32 P16 weights
int launch_gemm16_f32(int epi, bool ln, const Gemm16Args& a, hipStream_t s) { note_scheme(SCHEME_F32); return dispatch<PF32>(epi, ln, a, s); }
