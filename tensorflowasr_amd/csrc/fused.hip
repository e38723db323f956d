// Block-level fused kernels (dmodel 144): whole runs of token-local layers of a ConformerBlock in one launch.
//
// Why: at the benchmark shape one layer is 1-5 GFLOP = 13-57 us, and a pure-MFMA kernel of that size already
// loses ~35 % to launch / ramp-up / drain (tools/ubench/mfma_stream.hip: 4000 waves x 650 MFMAs -> 101 TFLOP/s vs
// 150 for long waves).  Because the transposed-chain layout keeps a token tile's activations in registers across
// any number of GEMMs, the token-local layers between the two layers that mix tokens (attention, depthwise conv)
// can be one kernel each:
//   ff1_qkv_kernel    x0 -> x1 = x0 + fc*FFN1(LN(x0))  ;  qkv = LN(x1) Wqkv (+b), q scaled      (3 GEMMs)
//   out_glu_kernel    x2 = x1 + ctx Wo + bo            ;  u = GLU(LN(x2) Wpw1 + b)                (2 GEMMs)
//   tail_ff2_kernel   x3 = x2 + pw2(swish(BN(dw Wpc + b))) + b ; y = LN(x3 + fc*FFN2(LN(x3)))    (4 GEMMs)
// One wave = 16 tokens, no LDS, no barriers.  Weight fragments are fetched one k-block ahead and the fetches are
// interleaved with the MFMAs at tile-pair granularity (2 loads, 8 MFMAs, fenced): the micro-benchmark shows that
// schedule holding 131-145 TFLOP/s whether one or several waves share a SIMD, where batch-granular prefetch
// (all loads, then all MFMAs) falls to ~100 as soon as two waves are co-resident.
// Reference semantics: asr/models/conformer_blocks.py:126-134 (FFModule), :164-170 + multihead_attention.py:151-188
// (MHSA), :209-219 (ConvModule), :259-265 (block).
#include "common.h"
#include "launch.h"

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;   // 9

// The weight stream of a kernel is one sequence of 9-fragment batches that runs across GEMM and stage
// boundaries: every GEMM routine enters with its first batch already in `wc` (fetched by whoever ran before it)
// and leaves with the batch at `next` in `wc`, fetched during its own last k-step -- so a stage never starts with
// an exposed load.  Loads are interleaved with the MFMAs at tile-pair granularity (2 loads, 8 MFMAs, fenced).
constexpr int NB = KB;   // fragments per batch (dmodel 144: every batch on the path is 9 fragments)

DEV void load_batch(f32x4 (&w)[NB], const f32x4* __restrict__ p) {
#pragma unroll
  for (int i = 0; i < NB; ++i) w[i] = p[(size_t)i * 64];
}

// acc[i] += W[kb][n0 + i]^T * x[kb], i < 9, kb < KBI;  W packed [KBI][NT] fragments.
template <int KBI>
DEV void wave_gemm(f32x4 (&acc)[NB], const f32x4 (&x)[KBI], const f32x4* __restrict__ wp, int NT, int n0,
                   f32x4 (&wc)[NB], const f32x4* __restrict__ next) {
  f32x4 wn[NB];
#pragma unroll
  for (int kb = 0; kb < KBI; ++kb) {
    const f32x4* src = (kb + 1 < KBI) ? wp + (size_t)((kb + 1) * NT + n0) * 64 : next;
#pragma unroll
    for (int i0 = 0; i0 < NB; i0 += 2) {
#pragma unroll
      for (int i = i0; i < i0 + 2 && i < NB; ++i) wn[i] = src[(size_t)i * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < NB; ++i) acc[i] = mfma4(wc[i][j], x[kb][j], acc[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) wc[i] = wn[i];
  }
}

// acc2[n2] += W2[h0 + n1][n2]^T * act(h[n1])  for n1 < 9 hidden tiles, n2 < 9 output tiles (second GEMM of a chain).
// The activation of hidden tile n1+1 is evaluated inside the fenced MFMA region of tile n1, so its VALU work
// (v_exp, v_rcp) issues under the matrix pipe's shadow instead of between the two GEMMs.
template <class ACT>
DEV void wave_gemm2(f32x4 (&acc2)[NB], f32x4 (&h)[NB], const f32x4* __restrict__ wp, int h0, f32x4 (&wc)[NB],
                    const f32x4* __restrict__ next, ACT act) {
  f32x4 wn[NB];
  h[0] = act(0, h[0]);
#pragma unroll
  for (int n1 = 0; n1 < NB; ++n1) {
    const f32x4* src = (n1 + 1 < NB) ? wp + (size_t)((h0 + n1 + 1) * KB) * 64 : next;
#pragma unroll
    for (int i0 = 0; i0 < NB; i0 += 2) {
#pragma unroll
      for (int i = i0; i < i0 + 2 && i < NB; ++i) wn[i] = src[(size_t)i * 64];
      __builtin_amdgcn_sched_barrier(0);
      if (i0 == 0 && n1 + 1 < NB) h[n1 + 1] = act(n1 + 1, h[n1 + 1]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < NB; ++i) acc2[i] = mfma4(wc[i][j], h[n1][j], acc2[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) wc[i] = wn[i];
  }
}

// y = W2 act( W1 xin + b1 )  with the hidden dimension swept in chunks of 9 tiles.
//   AFF: act = swish(s * . + t) (folded BatchNorm), else act = swish
// enters with W1's first batch (k-block 0, hidden tiles 0..8) in wc; leaves with the batch at `next` in wc.
template <int HT, bool AFF>
DEV void wave_chain(f32x4 (&y)[KB], const f32x4 (&xin)[KB], const f32x4* __restrict__ w1, const float* __restrict__ b1,
                    const float* __restrict__ aff_s, const float* __restrict__ aff_t, const f32x4* __restrict__ w2,
                    int g4, f32x4 (&wc)[NB], const f32x4* __restrict__ next) {
  static_assert(HT % NB == 0, "hidden tiles must split evenly");
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = splat4(0.f);
#pragma unroll 1
  for (int c = 0; c < HT / NB; ++c) {
    const int h0 = c * NB;
    f32x4 h[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) h[i] = ldg4(b1 + 16 * (h0 + i) + g4);
    f32x4 as[AFF ? NB : 1], at[AFF ? NB : 1];           // folded-BN scale/shift: fetched now, used after GEMM1
    if (AFF) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        as[i] = ldg4(aff_s + 16 * (h0 + i) + g4);
        at[i] = ldg4(aff_t + 16 * (h0 + i) + g4);
      }
    }
    // GEMM1 of this chunk, then GEMM2; the stream continues into the next chunk's GEMM1 (or `next`)
    wave_gemm<KB>(h, xin, w1, HT, h0, wc, w2 + (size_t)(h0 * KB) * 64);
    const f32x4* after = (c + 1 < HT / NB) ? w1 + (size_t)(0 * HT + h0 + NB) * 64 : next;
    auto act = [&](int i, f32x4 v) -> f32x4 {
      if (AFF) return swish4(v * as[AFF ? i : 0] + at[AFF ? i : 0]);
      return swish4(v);
    };
    wave_gemm2(y, h, w2, h0, wc, after, act);
  }
}

struct WaveCtx {
  int lane, g4, t, tok;
  size_t row;
  bool live;
};
DEV WaveCtx wave_ctx(int M) {
  WaveCtx c;
  c.lane = threadIdx.x & 63;
  c.g4 = (c.lane >> 4) * 4;
  c.t = c.lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  c.tok = wid * 16 + c.t;
  c.live = c.tok < M;
  c.row = (size_t)min(c.tok, M - 1) * D;
  return c;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS) void ff1_qkv_kernel(Ff1QkvArgs a) {
  if ((size_t)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16 >= (size_t)a.M) return;
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.ff_w1p) + c.lane;
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.ff_w2p) + c.lane;
  const f32x4* wq = reinterpret_cast<const f32x4*>(a.qkv_wp) + c.lane;
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x0 + c.row + 16 * kb + c.g4);
  f32x4 wc[NB];
  load_batch(wc, w1);                                   // first batch rides under the LayerNorm
  ln_apply<KB>(xs, a.ff_ln_g, a.ff_ln_b, c.g4, a.eps);
  wave_chain<4 * KB, false>(y, xs, w1, a.ff_b1, nullptr, nullptr, w2, c.g4, wc, wq);
  // x1 = x0 + fc * (ffn + b2)
#pragma unroll
  for (int i = 0; i < KB; ++i)
    xs[i] = ldg4(a.x0 + c.row + 16 * i + c.g4) + splat4(a.fc) * (y[i] + ldg4(a.ff_b2 + 16 * i + c.g4));
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x1 + c.row + 16 * i + c.g4, xs[i]);
  }
  // qkv = LN(x1) Wqkv + b, query tiles scaled
  ln_apply<KB>(xs, a.att_ln_g, a.att_ln_b, c.g4, a.eps);
  float* qrow = a.qkv + (size_t)min(c.tok, a.M - 1) * (3 * D);
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = ldg4(a.qkv_b + 16 * (q * KB + i) + c.g4);
    wave_gemm<KB>(acc, xs, wq, 3 * KB, q * KB, wc, wq + (size_t)(min(q + 1, 2) * KB) * 64);
    const float sc = (q == 0) ? a.qscale : 1.0f;
    if (c.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(qrow + 16 * (q * KB + i) + c.g4, acc[i] * splat4(sc));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS) void out_glu_kernel(OutGluArgs a) {
  if ((size_t)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16 >= (size_t)a.M) return;
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* wo = reinterpret_cast<const f32x4*>(a.out_wp) + c.lane;
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.pw1_wp) + c.lane;
  f32x4 xs[KB], acc[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.ctx + c.row + 16 * kb + c.g4);
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] = ldg4(a.out_b + 16 * i + c.g4);
  f32x4 wc[NB];
  load_batch(wc, wo);
  wave_gemm<KB>(acc, xs, wo, KB, 0, wc, wg);
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = ldg4(a.x1 + c.row + 16 * i + c.g4) + acc[i];     // x2 = x1 + attention
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x2 + c.row + 16 * i + c.g4, xs[i]);
  }
  ln_apply<KB>(xs, a.cv_ln_g, a.cv_ln_b, c.g4, a.eps);
  f32x4 gate[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    acc[i] = ldg4(a.pw1_b + 16 * i + c.g4);
    gate[i] = ldg4(a.pw1_b + 16 * (KB + i) + c.g4);
  }
  wave_gemm<KB>(acc, xs, wg, 2 * KB, 0, wc, wg + (size_t)KB * 64);
  wave_gemm<KB>(gate, xs, wg, 2 * KB, KB, wc, wg);
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const f32x4 va = acc[i], vb = gate[i];
      f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
      stg4(a.u + c.row + 16 * i + c.g4, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS) void tail_ff2_kernel(TailFf2Args a) {
  if ((size_t)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16 >= (size_t)a.M) return;
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* wpc = reinterpret_cast<const f32x4*>(a.pc_w1p) + c.lane;
  const f32x4* wp2 = reinterpret_cast<const f32x4*>(a.pw2_wp) + c.lane;
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.ff_w1p) + c.lane;
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.ff_w2p) + c.lane;
  f32x4 xs[KB], y[KB], x3[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
  f32x4 wc[NB];
  load_batch(wc, wpc);
  wave_chain<2 * KB, true>(y, xs, wpc, a.pc_b1, a.bn_s, a.bn_t, wp2, c.g4, wc, w1);
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    x3[i] = ldg4(a.x2 + c.row + 16 * i + c.g4) + (y[i] + ldg4(a.pw2_b + 16 * i + c.g4));   // conv module residual
    xs[i] = x3[i];
  }
  ln_apply<KB>(xs, a.ff_ln_g, a.ff_ln_b, c.g4, a.eps);
  wave_chain<4 * KB, false>(y, xs, w1, a.ff_b1, nullptr, nullptr, w2, c.g4, wc, w1);
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = x3[i] + splat4(a.fc) * (y[i] + ldg4(a.ff_b2 + 16 * i + c.g4));
  ln_apply<KB>(y, a.ln_g, a.ln_b, c.g4, a.eps);                                              // block-final LayerNorm
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.y + c.row + 16 * i + c.g4, y[i]);
  }
}

}  // namespace

int launch_ff1_qkv(const Ff1QkvArgs& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  hipLaunchKernelGGL(ff1_qkv_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
int launch_out_glu(const OutGluArgs& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  hipLaunchKernelGGL(out_glu_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
int launch_tail_ff2(const TailFf2Args& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  hipLaunchKernelGGL(tail_ff2_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
