// Shared device helpers for the MI355X (gfx950 / CDNA4) Conformer-CTC kernels.
//
// Design in one paragraph (DESIGN.md has the long form): every dense contraction on the path is
// a tall-skinny GEMM  Y[M, N] = X[M, K] * W[K, N]  with M = tokens (B*T, 16 000 at the benchmark
// shape) and K, N in {80..2880}.  Each 64-lane wavefront owns 16*RT tokens and keeps them in
// registers for a whole chain of GEMMs: it issues v_mfma_f32_16x16x4_f32 (exact fp32) as
// D = Wfrag * Xfrag, i.e. the *transposed* product, so the accumulator lane layout
//     lane (g = lane>>4, t = lane&15), reg j  <->  Y[token t][feature 16*nt + 4*g + j]
// is at the same time (a) a float4 of 4 consecutive features of one token -> 16-byte global
// loads/stores, and (b) exactly the B-operand fragment of the next GEMM in the chain
// (k = 16*kb + 4*g + j).  Activations therefore never go through LDS between fused GEMMs, waves
// never synchronise with each other, and the weights -- pre-packed on the host into fragment
// order ("P16": [K/16][N/16][64 lanes][4]) -- stream from L2 as perfectly coalesced 1 KiB
// wave loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64
#define WAVES_PER_BLOCK 4
#define BLOCK_THREADS (WAVE * WAVES_PER_BLOCK)

#define DEV __device__ __forceinline__
// __launch_bounds__(BLOCK_THREADS, 2) on the MFMA kernels: asking for two waves per SIMD makes hipcc allocate
// everything in <= 256 architectural VGPRs (no AGPR half, hence no v_accvgpr_read/write shuffling) and gives
// every SIMD a second wave whose MFMAs cover the first one's address arithmetic and load issue.

// D(16x16) += A(16x4) * B(4x16); lane supplies A[i=lane&15][k=lane>>4] and B[k=lane>>4][j=lane&15];
// result reg r of lane holds D[row = 4*(lane>>4)+r][col = lane&15]  (cdna_hip_programming.md §3).
DEV f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 16-wide k-block: acc += Wblock^T * Xblock  with w = packed weight fragment, x = token fragment.
DEV f32x4 mma_kblock(f32x4 w, f32x4 x, f32x4 acc) {
  acc = mfma4(w.x, x.x, acc);
  acc = mfma4(w.y, x.y, acc);
  acc = mfma4(w.z, x.z, acc);
  acc = mfma4(w.w, x.w, acc);
  return acc;
}

// Batch form: acc[i] += W_i-block^T * X-block for N independent accumulators, issued k-step-major so that
// consecutive MFMAs never hit the same accumulator (v_mfma_f32_16x16x4_f32 issues every 32 cycles but a
// dependent one waits 40: four back-to-back MFMAs on one accumulator cap the pipe at 80 %).
template <int N>
DEV void mma_batch(f32x4 (&acc)[N], const f32x4 (&w)[N], f32x4 x) {
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].x, x.x, acc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].y, x.y, acc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].z, x.z, acc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].w, x.w, acc[i]);
}
// same, using the first N entries of a larger fragment buffer
template <int N, int NA, int NW>
DEV void mma_batch_n(f32x4 (&acc)[NA], const f32x4 (&w)[NW], f32x4 x) {
  static_assert(N <= NA && N <= NW, "batch larger than its buffers");
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].x, x.x, acc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].y, x.y, acc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].z, x.z, acc[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = mfma4(w[i].w, x.w, acc[i]);
}
template <int RT, int N>
DEV void mma_batch_rt(f32x4 (&acc)[RT][N], const f32x4 (&w)[N], const f32x4 (&x)[RT]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(w[i].x, x[rt].x, acc[rt][i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(w[i].y, x[rt].y, acc[rt][i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(w[i].z, x[rt].z, acc[rt][i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(w[i].w, x[rt].w, acc[rt][i]);
}

DEV f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEV void stg4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
DEV f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }

// sum / max over the 4 lane groups that share a token (lanes t, t+16, t+32, t+48)
// (Round 5 measured gfx950's v_permlane16_swap_b32 / v_permlane32_swap_b32 here -- handed the same value twice they return, in every
// lane, the pair a __shfl_xor(v, 16) / (v, 32) butterfly combines, on the VALU instead of an LDS round trip; the primitive is bit for
// bit the butterfly (tools/ubench/group_reduce_cmp.hip), a tail_ff1 launch 1.3 us shorter by events, the step 0.15 % on the clock,
// and hipcc contracts the surrounding arithmetic of the pair-pipelined kernels differently, so the library's outputs move in the
// last bit: not kept.  One trap on the way: __builtin_bit_cast(float, r.y) on the ELEMENT of the returned vector reads element 0.)
DEV float group_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
DEV float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

DEV float fast_sigmoid(float x) {
  // 1/(1+e^-x): v_exp_f32 + v_rcp_f32 (both ~1 ulp); saturates cleanly for |x| large.
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
DEV float swishf(float x) { return x * fast_sigmoid(x); }
DEV f32x4 swish4(f32x4 v) {
  f32x4 r = {swishf(v.x), swishf(v.y), swishf(v.z), swishf(v.w)};
  return r;
}

// LayerNorm statistics of one token whose D = 16*KB features live as xs[kb] in the 4 lanes of its group.
// Two-pass (mean, then centred second moment), biased variance, eps inside the sqrt -- Keras semantics.
template <int KB>
DEV void ln_stats(const f32x4 (&xs)[KB], float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) s += (xs[kb].x + xs[kb].y) + (xs[kb].z + xs[kb].w);
  s = group_sum(s);
  mean = s * (1.0f / (16 * KB));
  float q = 0.f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    f32x4 d = xs[kb] - splat4(mean);
    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
  q = group_sum(q);
  rstd = 1.0f / sqrtf(q * (1.0f / (16 * KB)) + eps);
}

template <int KB>
DEV void ln_apply(f32x4 (&xs)[KB], const float* __restrict__ gamma, const float* __restrict__ beta, int g4,
                  float eps) {
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    f32x4 ga = ldg4(gamma + 16 * kb + g4);
    f32x4 be = ldg4(beta + 16 * kb + g4);
    xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * ga + be;
  }
}

// Software-pipelined k-sweep for kernels whose K extent is a run-time value:
//   acc[rt][i] += W[kb][c0 + i]^T * X_rt[kb]   for kb in [0, KBT), KBT even, rt < RT row tiles.
// The fragments of step kb+1 are fetched while the MFMAs of step kb issue, and the fetches are *interleaved*
// with the MFMA groups: one group = GT column tiles = GT weight loads followed by GT*4*RT MFMAs, fenced by
// sched_barrier(0).  A single clump of ~200 address/load instructions between two MFMA batches would leave
// the matrix pipe idle for ~1500 cycles per step when only one wave is resident on the SIMD; spread out, each
// group's handful of non-MFMA instructions hides under its own MFMAs (32 cycles each).  Without the fences
// hipcc sinks every load to its first use.  With RT = 2 one weight fragment feeds two MFMA groups, halving
// the L2 weight stream per flop.
//   wp : packed weights + lane, NT = column tiles per k-block, c0 = first column tile of this wave
//   xp : functor (rt, kb) -> f32x4: the raw loads of this lane's operand fragment (branch-free, NO arithmetic on
//        the loaded values: anything that consumes them would be waited for at issue time)
//   fx : functor (rt, kb, raw) -> f32x4: masking / normalisation of the raw fragment, applied at use time
template <int RT, int CT, int GT, class XP, class FX>
DEV void sweep_step(f32x4 (&acc)[RT][CT], const f32x4 (&wc)[CT], f32x4 (&xc)[RT], f32x4 (&wn)[CT], f32x4 (&xn)[RT],
                    const f32x4* __restrict__ wnext, int kb_cur, int kb_next, XP& xp, FX& fx) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) xc[rt] = fx(rt, kb_cur, xc[rt]);
#pragma unroll
  for (int g0 = 0; g0 < CT; g0 += GT) {
#pragma unroll
    for (int i = g0; i < g0 + GT && i < CT; ++i) wn[i] = wnext[(size_t)i * 64];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (rt * GT >= g0 && rt * GT < g0 + GT) xn[rt] = xp(rt, kb_next);   // operand loads ride with the first groups
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = g0; i < g0 + GT && i < CT; ++i)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(wc[i][j], xc[rt][j], acc[rt][i]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RT, int CT, class XP, class FX>
DEV void sweep_k(f32x4 (&acc)[RT][CT], const f32x4* __restrict__ wp, int NT, int c0, int KBT, XP xp, FX fx) {
  constexpr int GT = (RT >= 2) ? 1 : 2;   // >= 2 independent accumulators per MFMA group (40-cycle dependent latency)
  static_assert(RT * GT <= CT || RT == 1, "operand loads must fit in the group sequence");
  f32x4 w0[CT], w1[CT], x0[RT], x1[RT];
#pragma unroll
  for (int i = 0; i < CT; ++i) w0[i] = wp[(size_t)(0 * NT + c0 + i) * 64];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) x0[rt] = xp(rt, 0);
#pragma unroll 1
  for (int kb = 0; kb < KBT; kb += 2) {
    sweep_step<RT, CT, GT>(acc, w0, x0, w1, x1, wp + (size_t)((kb + 1) * NT + c0) * 64, kb, kb + 1, xp, fx);
    const int kn = (kb + 2 < KBT) ? kb + 2 : kb;   // clamped: the last prefetch is redundant but harmless
    sweep_step<RT, CT, GT>(acc, w1, x1, w0, x0, wp + (size_t)(kn * NT + c0) * 64, kb + 1, kn, xp, fx);
  }
}
