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

DEV f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEV void stg4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
DEV f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }

// sum / max over the 4 lane groups that share a token (lanes t, t+16, t+32, t+48)
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
//   acc[rt][i] += W[kb][c0 + i]^T * X_rt[kb]   for kb in [0, KBT), KBT even, rt < RT row tiles,
// with the fragments of step kb+1 (CT weight fragments + RT operand fragments) in flight while the MFMAs of
// step kb issue.  sched_barrier(0) keeps hipcc from sinking the loads back to their first use.  With RT = 2 one
// weight fragment feeds two MFMA groups, halving the L2 weight stream per flop.
//   wp : packed weights + lane, NT = column tiles per k-block, c0 = first column tile of this wave
//   xp : functor (rt, kb) -> f32x4: the raw loads of this lane's operand fragment (branch-free, NO arithmetic on
//        the loaded values: anything that consumes them would be waited for at issue time)
//   fx : functor (rt, kb, raw) -> f32x4: masking / normalisation of the raw fragment, applied at use time
template <int RT, int CT, class XP, class FX>
DEV void sweep_k(f32x4 (&acc)[RT][CT], const f32x4* __restrict__ wp, int NT, int c0, int KBT, XP xp, FX fx) {
  f32x4 w0[CT], w1[CT], x0[RT], x1[RT];
#pragma unroll
  for (int i = 0; i < CT; ++i) w0[i] = wp[(size_t)(0 * NT + c0 + i) * 64];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) x0[rt] = xp(rt, 0);
#pragma unroll 1
  for (int kb = 0; kb < KBT; kb += 2) {
#pragma unroll
    for (int i = 0; i < CT; ++i) w1[i] = wp[(size_t)((kb + 1) * NT + c0 + i) * 64];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x1[rt] = xp(rt, kb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x0[rt] = fx(rt, kb, x0[rt]);
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mma_kblock(w0[i], x0[rt], acc[rt][i]);
    __builtin_amdgcn_sched_barrier(0);
    const int kn = (kb + 2 < KBT) ? kb + 2 : kb;   // clamped: the last prefetch is redundant but harmless
#pragma unroll
    for (int i = 0; i < CT; ++i) w0[i] = wp[(size_t)(kn * NT + c0 + i) * 64];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x0[rt] = xp(rt, kn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x1[rt] = fx(rt, kb + 1, x1[rt]);
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mma_kblock(w1[i], x1[rt], acc[rt][i]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
