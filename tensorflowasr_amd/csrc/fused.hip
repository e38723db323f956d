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
// One wave = 16 tokens and (at 16 000 tokens) one wave per SIMD, so nothing hides a stall: every load a wave
// waits for is matrix-pipe idle time.  Hence
//   * weight fragments are fetched one k-block ahead, interleaved with the MFMAs at tile-pair granularity
//     (2 loads, 8 MFMAs, fenced) -- the micro-benchmark holds 124-138 TFLOP/s with that schedule at this shape;
//   * every small parameter vector of the kernel (biases, LayerNorm gamma/beta, folded BatchNorm) is copied to
//     LDS once per workgroup at kernel start and read from there (~100 cycles, issued a fence group ahead of its
//     use) instead of from L2 (~700 cycles, exposed at each of the 8-10 stage boundaries of a kernel);
//   * bias + activation of hidden tile n+1 runs inside the fenced MFMA region of tile n (v_exp / v_rcp in the
//     matrix pipe's shadow), the residual input stays in registers;
//   * the kernels fit 256 registers (__launch_bounds__(256, 2)), so hipcc keeps the accumulators in VGPRs; with
//     the 512-register budget it parks them in AGPRs and moves them back and forth around every MFMA group.
// Reference semantics: asr/models/conformer_blocks.py:126-134 (FFModule), :164-170 + multihead_attention.py:151-188
// (MHSA), :209-219 (ConvModule), :259-265 (block).
#include "common.h"
#include "launch.h"
#include "wstream.h"

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;   // 9
constexpr int NB = KB;       // fragments per batch (dmodel 144: every batch on the path is 9 fragments)

// acc[i] += W[t]^T * x[t] for the 9 batches own(0..8) of one GEMM; next0 / next1 = the two batches that follow it
// in the kernel's stream.  Enters with CUR, leaves with CUR^1 (9 is odd).  hook(T, GI) as in batch_step.
template <int CUR, class OWN, class HOOK>
DEV void wave_gemm(f32x4 (&acc)[NB], const f32x4 (&x)[KB], WStream<NB>& s, OWN&& own, const f32x4* __restrict__ next0,
                   const f32x4* __restrict__ next1, HOOK&& hook) {
  const unsigned l16 = fresh_lane16(s.lane16);
  static_for<0, KB>([&](auto T) {
    constexpr int t = decltype(T)::value;
    const f32x4* p1 = (t + 1 < KB) ? own(t + 1) : next0;
    const f32x4* p2 = (t + 2 < KB) ? own(t + 2) : (t + 2 == KB ? next0 : next1);
    batch_step<(CUR + t) & 1>(acc, x[t], s, l16, p1, p2, [&](auto GI) { hook(T, GI); });
  });
}

// Bias / folded-BatchNorm / swish of one hidden tile, split in two so that the LDS reads are issued one fence
// group before the VALU work that consumes them.
template <bool AFF>
struct Act {
  const float *b1, *as, *at;   // LDS
  int g4;
  f32x4 pb, ps, pt;
  DEV void fetch(int tile) {
    pb = lds4(b1, tile, g4);
    if (AFF) { ps = lds4(as, tile, g4); pt = lds4(at, tile, g4); }
  }
  DEV f32x4 apply(f32x4 v) const {
    v = v + pb;
    if (AFF) v = v * ps + pt;
    return swish4(v);
  }
};

// y += W2 act( W1 xin + b1 )  with the hidden dimension swept in chunks of 9 tiles; b1 / aff_* are LDS pointers.
//   AFF: act = swish(s * (. + b1) + t) (folded BatchNorm), else act = swish(. + b1)
// Stream: per chunk 9 batches of W1 (k-block t, hidden tiles h0..h0+8) then 9 batches of W2 (hidden tile h0+t as
// the k-block); next0 / next1 = the two batches after the chain.  Enters and leaves with CUR = 0.
template <int HT, bool AFF>
DEV void wave_chain(f32x4 (&y)[KB], const f32x4 (&xin)[KB], const f32x4* __restrict__ w1, const float* b1,
                    const float* aff_s, const float* aff_t, const f32x4* __restrict__ w2, int g4, WStream<NB>& s,
                    const f32x4* __restrict__ next0, const f32x4* __restrict__ next1) {
  static_assert(HT % NB == 0, "hidden tiles must split evenly");
  Act<AFF> act{b1, aff_s, aff_t, g4, {}, {}, {}};
#pragma unroll 1
  for (int c = 0; c < HT / NB; ++c) {
    const int h0 = c * NB;
    f32x4 h[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) h[i] = splat4(0.f);
    auto own1 = [&](int t) { return w1 + (size_t)(t * HT + h0) * 64; };
    auto own2 = [&](int t) { return w2 + (size_t)((h0 + t) * KB) * 64; };
    // GEMM1: hidden tile 0 is complete after group 0 of the last batch and is activated two groups later
    wave_gemm<0>(h, xin, s, own1, own2(0), own2(1), [&](auto T, auto GI) {
      if constexpr (decltype(T)::value == KB - 1 && decltype(GI)::value == 1) act.fetch(h0);
      if constexpr (decltype(T)::value == KB - 1 && decltype(GI)::value == 2) h[0] = act.apply(h[0]);
    });
    // GEMM2: hidden tile t+1 is activated inside the fenced MFMA region of tile t
    const bool last = (c + 1 == HT / NB);
    const f32x4* a0 = last ? next0 : w1 + (size_t)(0 * HT + h0 + NB) * 64;
    const f32x4* a1 = last ? next1 : w1 + (size_t)(1 * HT + h0 + NB) * 64;
    wave_gemm<1>(y, h, s, own2, a0, a1, [&](auto T, auto GI) {
      constexpr int t = decltype(T)::value;
      if constexpr (t + 1 < NB && decltype(GI)::value == 0) act.fetch(h0 + t + 1);
      if constexpr (t + 1 < NB && decltype(GI)::value == 1) h[t + 1] = act.apply(h[t + 1]);
    });
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
  c.row = (size_t)min(c.tok, M - 1) * D;     // waves past the end recompute the last token and store nothing
  return c;
}

// LayerNorm with gamma / beta in LDS
DEV void ln_lds(f32x4 (&xs)[KB], const float* ga, const float* be, int g4, float eps) {
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * lds4(ga, kb, g4) + lds4(be, kb, g4);
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS, 2) void ff1_qkv_kernel(Ff1QkvArgs a) {
  __shared__ __attribute__((aligned(16))) float p_ln1g[D], p_ln1b[D], p_b1[4 * D], p_b2[D], p_ln2g[D], p_ln2b[D], p_qb[3 * D];
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.ff_w1p);
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.ff_w2p);
  const f32x4* wq = reinterpret_cast<const f32x4*>(a.qkv_wp);
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x0 + c.row + 16 * kb + c.g4);
  WStream<NB> ws;
  ws.lane16 = (unsigned)c.lane * 16u;
  stream_begin(ws, w1, w1 + (size_t)(4 * KB) * 64);    // first two batches ride under the stash + LayerNorm
  stash(p_ln1g, a.ff_ln_g, D); stash(p_ln1b, a.ff_ln_b, D); stash(p_b1, a.ff_b1, 4 * D); stash(p_b2, a.ff_b2, D);
  stash(p_ln2g, a.att_ln_g, D); stash(p_ln2b, a.att_ln_b, D); stash(p_qb, a.qkv_b, 3 * D);
  __syncthreads();
  // the residual rides in the accumulator: y = x0/fc + b2 + W2 h, x1 = fc*y (fc = 0.5 in every reference
  // config, so the scaling is exact); keeping x0 in registers across the chain would cost 36 VGPRs
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = lds4(p_b2, kb, c.g4) + splat4(inv_fc) * xs[kb];
  ln_lds(xs, p_ln1g, p_ln1b, c.g4, a.eps);
  wave_chain<4 * KB, false>(y, xs, w1, p_b1, nullptr, nullptr, w2, c.g4, ws, wq, wq + (size_t)(3 * KB) * 64);
  // x1 = x0 + fc * (ffn + b2)
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = splat4(a.fc) * y[i];
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x1 + c.row + 16 * i + c.g4, xs[i]);
  }
  // qkv = LN(x1) Wqkv + b, query tiles scaled
  ln_lds(xs, p_ln2g, p_ln2b, c.g4, a.eps);
  float* qrow = a.qkv + (size_t)min(c.tok, a.M - 1) * (3 * D);
  static_for<0, 3>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    constexpr int qn = q < 2 ? q + 1 : 2;                 // after the last GEMM the stream idles on valid addresses
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = lds4(p_qb, q * KB + i, c.g4);
    wave_gemm<q & 1>(acc, xs, ws, [&](int t) { return wq + (size_t)(t * 3 * KB + q * KB) * 64; },
                     wq + (size_t)(qn * KB) * 64, wq + (size_t)(3 * KB + qn * KB) * 64, NoHook());
    const float sc = (q == 0) ? a.qscale : 1.0f;
    if (c.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(qrow + 16 * (q * KB + i) + c.g4, acc[i] * splat4(sc));
    }
  });
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS, 2) void out_glu_kernel(OutGluArgs a) {
  __shared__ __attribute__((aligned(16))) float p_ob[D], p_lng[D], p_lnb[D], p_pb[2 * D];
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* wo = reinterpret_cast<const f32x4*>(a.out_wp);
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.pw1_wp);
  f32x4 xs[KB], acc[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.ctx + c.row + 16 * kb + c.g4);
  WStream<NB> ws;
  ws.lane16 = (unsigned)c.lane * 16u;
  stream_begin(ws, wo, wo + (size_t)KB * 64);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) acc[kb] = ldg4(a.x1 + c.row + 16 * kb + c.g4);  // residual rides in the accumulator
  stash(p_ob, a.out_b, D); stash(p_lng, a.cv_ln_g, D); stash(p_lnb, a.cv_ln_b, D); stash(p_pb, a.pw1_b, 2 * D);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] += lds4(p_ob, i, c.g4);
  wave_gemm<0>(acc, xs, ws, [&](int t) { return wo + (size_t)(t * KB) * 64; }, wg, wg + (size_t)(2 * KB) * 64, NoHook());
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = acc[i];                                      // x2 = x1 + attention
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x2 + c.row + 16 * i + c.g4, xs[i]);
  }
  ln_lds(xs, p_lng, p_lnb, c.g4, a.eps);
  f32x4 gate[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    acc[i] = lds4(p_pb, i, c.g4);
    gate[i] = lds4(p_pb, KB + i, c.g4);
  }
  wave_gemm<1>(acc, xs, ws, [&](int t) { return wg + (size_t)(t * 2 * KB) * 64; }, wg + (size_t)KB * 64,
               wg + (size_t)(3 * KB) * 64, NoHook());
  wave_gemm<0>(gate, xs, ws, [&](int t) { return wg + (size_t)(t * 2 * KB + KB) * 64; }, wg, wg, NoHook());
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
__global__ __launch_bounds__(BLOCK_THREADS, 2) void tail_ff2_kernel(TailFf2Args a) {
  __shared__ __attribute__((aligned(16))) float p_pcb[2 * D], p_bns[2 * D], p_bnt[2 * D], p_pw2b[D], p_lng[D], p_lnb[D], p_b1[4 * D],
      p_b2[D], p_fg[D], p_fb[D];
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* wpc = reinterpret_cast<const f32x4*>(a.pc_w1p);
  const f32x4* wp2 = reinterpret_cast<const f32x4*>(a.pw2_wp);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.ff_w1p);
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.ff_w2p);
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
  WStream<NB> ws;
  ws.lane16 = (unsigned)c.lane * 16u;
  stream_begin(ws, wpc, wpc + (size_t)(2 * KB) * 64);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + c.row + 16 * kb + c.g4);     // residuals ride in the accumulators
  stash(p_pcb, a.pc_b1, 2 * D); stash(p_bns, a.bn_s, 2 * D); stash(p_bnt, a.bn_t, 2 * D); stash(p_pw2b, a.pw2_b, D);
  stash(p_lng, a.ff_ln_g, D); stash(p_lnb, a.ff_ln_b, D); stash(p_b1, a.ff_b1, 4 * D); stash(p_b2, a.ff_b2, D);
  stash(p_fg, a.ln_g, D); stash(p_fb, a.ln_b, D);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] += lds4(p_pw2b, i, c.g4);
  wave_chain<2 * KB, true>(y, xs, wpc, p_pcb, p_bns, p_bnt, wp2, c.g4, ws, w1, w1 + (size_t)(4 * KB) * 64);
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    xs[i] = y[i];                                                                    // x3 = x2 + conv module
    y[i] = lds4(p_b2, i, c.g4) + splat4(inv_fc) * y[i];                              // x3/fc + b2 (+ W2 h): see ff1_qkv
  }
  ln_lds(xs, p_lng, p_lnb, c.g4, a.eps);
  wave_chain<4 * KB, false>(y, xs, w1, p_b1, nullptr, nullptr, w2, c.g4, ws, w1, w1);
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = splat4(a.fc) * y[i];
  ln_lds(y, p_fg, p_fb, c.g4, a.eps);                                                // block-final LayerNorm
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
