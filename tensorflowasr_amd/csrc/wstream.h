// Register-resident weight stream for the one-wave-per-16-rows GEMM kernels (fused.hip, subconv in frontend.hip).
#pragma once
#include <type_traits>

#include "common.h"

// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

// ---- parameter stash --------------------------------------------------------------------------------------
// Small per-kernel parameter vectors (biases, LayerNorm gamma/beta, conv1 taps) are copied to LDS once per
// workgroup and read from there (~100 cycles) instead of from L2 (~700 cycles, exposed at every stage boundary).
DEV void stash(float* dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += BLOCK_THREADS) dst[i] = src[i];
}
// this lane's float4 of tile `tile` of a stashed vector (lanes of one 16-lane group read the same address)
DEV f32x4 lds4(const float* v, int tile, int g4) { return *reinterpret_cast<const f32x4*>(v + 16 * tile + g4); }

// ---- weight stream ----------------------------------------------------------------------------------------
// The weight stream of a kernel is one sequence of NB-fragment batches (one k-block x NB column tiles) that runs
// across GEMM and stage boundaries.  Two register buffers wb[0], wb[1] hold it: while batch t is consumed from
// wb[CUR] in fenced groups of two tiles (2+2+2+2+1 for NB = 9), the slots a group has just finished with are
// refilled with the same fragments of batch t+2, and wb[CUR^1] holds batch t+1.  A fragment is therefore requested
// ~2 NB - 2 fragments (64 MFMAs, ~2000 cycles at NB = 9) before its first use, with no more registers than plain
// double buffering; fetching only one batch ahead (~1100 cycles) left 17 % of the wave's cycles in s_waitcnt
// (SQ_WAIT_ANY / SQ_WAVE_CYCLES) because one wave per SIMD has nothing else to run while it waits on L2.
// Invariant on entry to batch t with CUR: wb[CUR] = batch t (all issued), wb[CUR^1] = batch t+1 minus its last group.
// Addresses are (uniform batch pointer in SGPRs) + (lane * 16 bytes in one VGPR): no per-load vector address math.
template <int NB>
struct WStream {
  f32x4 wb[2][NB];
  unsigned lane16;   // lane * 16 bytes
};
DEV f32x4 ldw(const f32x4* __restrict__ batch, int frag, unsigned lane16) {
  // pin the (uniform) fragment address to an SGPR pair so that the load is `global_load v, v_lane16, s[..]`;
  // left alone, hipcc folds the lane offset into a 64-bit VGPR base and spends two VALU adds per fragment
  unsigned long long p = reinterpret_cast<unsigned long long>(batch + frag * 64);
  asm("" : "+s"(p));     // opaque SGPR pair: keeps LLVM from re-associating the lane offset into the base
  typedef const __attribute__((address_space(1))) char* gptr;
  gptr sp = (gptr)p;
  return *(const __attribute__((address_space(1))) f32x4*)(sp + lane16);
}
// The 32-bit lane offset has to be re-materialised (opaquely) in the basic block that uses it: the
// `saddr + zext(voffset)` addressing mode is only selected when the zero-extension is visible in the same block.
DEV unsigned fresh_lane16(unsigned lane16) {
  asm volatile("" : "+v"(lane16));
  return lane16;
}
template <int NB>
DEV void stream_begin(WStream<NB>& s, const f32x4* __restrict__ b0, const f32x4* __restrict__ b1) {
  constexpr int LAST0 = 2 * ((NB + 1) / 2 - 1);   // first slot of the last group
#pragma unroll
  for (int i = 0; i < NB; ++i) s.wb[0][i] = ldw(b0, i, s.lane16);
#pragma unroll
  for (int i = 0; i < LAST0; ++i) s.wb[1][i] = ldw(b1, i, s.lane16);
}

// acc[i] += wb[CUR][i]^T * x  (one k-block, NB column tiles); p1 / p2 = addresses of batches t+1 / t+2.
// hook(gi) runs inside fence group gi: a place for VALU / LDS work that should issue under the MFMAs.
template <int CUR, int NB, class HOOK>
DEV void batch_step(f32x4 (&acc)[NB], const f32x4 x, WStream<NB>& s, unsigned l16, const f32x4* __restrict__ p1,
                    const f32x4* __restrict__ p2, HOOK&& hook) {
  constexpr int G = (NB + 1) / 2;
  static_for<0, G>([&](auto GI) {
    constexpr int gi = decltype(GI)::value;
    if constexpr (gi == 0) {
#pragma unroll
      for (int i = 2 * (G - 1); i < NB; ++i) s.wb[CUR ^ 1][i] = ldw(p1, i, l16);
    } else {
      s.wb[CUR][2 * gi - 2] = ldw(p2, 2 * gi - 2, l16);
      s.wb[CUR][2 * gi - 1] = ldw(p2, 2 * gi - 1, l16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 2 * gi; i < 2 * gi + 2 && i < NB; ++i) acc[i] = mfma4(s.wb[CUR][i][j], x[j], acc[i]);
    hook(GI);
    __builtin_amdgcn_sched_barrier(0);
  });
}

struct NoHook {
  template <class T, class G> DEV void operator()(T, G) const {}
  template <class G> DEV void operator()(G) const {}
};
