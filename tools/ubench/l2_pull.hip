// Micro-benchmark (round 4): what ONE compute unit pulls from L2 -- every workgroup streams the same buffer (L2 resident: 2 MB,
// the size of a Conformer block's two-term weight stream) front to back, with W waves and U 16-byte loads per lane in flight
// per wave, into registers (global_load_dwordx4) or straight into LDS (global_load_lds_dwordx4, the slab rings' path).
// Prints GB/s per workgroup (= per CU while workgroups <= 256 and W large enough that two do not share a CU) and in total.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/l2_pull.hip -o tools/ubench/l2_pull.bin && tools/ubench/l2_pull.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 8 bytes per lane (the bf16 weight fragments of bf16.hip): U loads of 512 B per wave, as chain256_bf16_kernel issues them
template <int U>
__global__ __launch_bounds__(1024) void pull8(const u32x2* __restrict__ src, unsigned n8, int reps, unsigned* out) {
  const unsigned nt = blockDim.x, tid = threadIdx.x;
  u32x2 acc = {0, 0};
  for (int r = 0; r < reps; ++r) {
    u32x2 a[U], b[U];
    unsigned i = tid;
#pragma unroll
    for (int u = 0; u < U; ++u, i += nt) a[u] = src[min(i, n8 - 1)];
    while (i < n8) {
#pragma unroll
      for (int u = 0; u < U; ++u, i += nt) b[u] = src[min(i, n8 - 1)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= a[u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u, i += nt) a[u] = src[min(i, n8 - 1)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= b[u];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= a[u];
  }
  if ((acc.x ^ acc.y) == 0x12345678u) out[0] = 1;
}
template <int U>
static void run8(const void* src, size_t bytes, int wgs, int waves, unsigned* out) {
  const unsigned n8 = (unsigned)(bytes / 8);
  const int reps = 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((pull8<U>), dim3(wgs), dim3(waves * 64), 0, 0, (const u32x2*)src, n8, 2, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((pull8<U>), dim3(wgs), dim3(waves * 64), 0, 0, (const u32x2*)src, n8, reps, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per = (double)bytes * reps / (ms * 1e-3) / 1e9;
  printf("regs8 %4.1f MB  wgs %4d  waves %2d  in flight %2d x  8 B per lane (%5.0f KB per workgroup)  %7.1f GB/s per workgroup  %6.2f TB/s total\n",
         bytes / 1048576.0, wgs, waves, 2 * U, waves * 64.0 * 2 * U * 8 / 1024, per, per * wgs / 1e3);
}

template <int U, bool DMA>
__global__ __launch_bounds__(1024) void pull(const u32x4* __restrict__ src, unsigned n16, int reps, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];      // DMA: U slots of 1 KB per wave
  const unsigned nt = blockDim.x, tid = threadIdx.x;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    if constexpr (DMA) {
      // each wave keeps 2 U DMAs in flight: group g of U instructions is waited for while group g + 1 is outstanding
      unsigned i = tid;
#pragma unroll
      for (int u = 0; u < U; ++u, i += nt)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                         (__attribute__((address_space(3))) void*)(lds + (wave * 2 * U + u) * 64), 16, 0, 0);
      int par = 1;
      for (; i < n16; par ^= 1) {
#pragma unroll
        for (int u = 0; u < U; ++u, i += nt)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + min(i, n16 - 1)),
                                           (__attribute__((address_space(3))) void*)(lds + (wave * 2 * U + par * U + u) * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 a[U], b[U];
      unsigned i = tid;
#pragma unroll
      for (int u = 0; u < U; ++u, i += nt) a[u] = src[min(i, n16 - 1)];
      while (i < n16) {
#pragma unroll
        for (int u = 0; u < U; ++u, i += nt) b[u] = src[min(i, n16 - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= a[u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u, i += nt) a[u] = src[min(i, n16 - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= b[u];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= a[u];
    }
  }
  if constexpr (DMA) acc = lds[tid & 63];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int U, bool DMA>
static void run(const u32x4* src, size_t bytes, int wgs, int waves, unsigned* out) {
  const unsigned n16 = (unsigned)(bytes / 16);
  const int reps = 8;
  const size_t dyn = DMA ? (size_t)waves * 2 * U * 1024 : 0;
  if (dyn > 160 * 1024) return;
  if (DMA) (void)hipFuncSetAttribute((const void*)pull<U, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((pull<U, DMA>), dim3(wgs), dim3(waves * 64), dyn, 0, src, n16, 2, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((pull<U, DMA>), dim3(wgs), dim3(waves * 64), dyn, 0, src, n16, reps, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
  const double per = (double)bytes * reps / (ms * 1e-3) / 1e9;
  printf("%-4s %4.1f MB  wgs %4d  waves %2d  in flight %2d x 16 B per lane (%5.0f KB per workgroup)  %7.1f GB/s per workgroup  %6.2f TB/s total\n",
         DMA ? "dma" : "regs", bytes / 1048576.0, wgs, waves, 2 * U, waves * 64.0 * 2 * U * 16 / 1024, per, per * wgs / 1e3);
}

int main() {
  const size_t maxb = 32u << 20;
  u32x4* src; unsigned* out;
  (void)hipMalloc((void**)&src, maxb); (void)hipMalloc((void**)&out, 64);
  (void)hipMemset(src, 1, maxb); (void)hipMemset(out, 0, 64);
  for (int wgs : {52, 256})
    for (int waves : {8}) {
      run8<8>(src, 1u << 20, wgs, waves, out);
      run8<16>(src, 1u << 20, wgs, waves, out);
      run8<24>(src, 1u << 20, wgs, waves, out);
    }
  for (size_t mb : {2u, 16u}) {
    for (int wgs : {52, 256}) {
      for (int waves : {4, 8, 16}) {
        run<2, false>(src, mb << 20, wgs, waves, out);
        run<4, false>(src, mb << 20, wgs, waves, out);
        run<8, false>(src, mb << 20, wgs, waves, out);
        run<2, true>(src, mb << 20, wgs, waves, out);
        run<4, true>(src, mb << 20, wgs, waves, out);
        run<8, true>(src, mb << 20, wgs, waves, out);
      }
    }
  }
  // two workgroups per CU
  run<4, false>(src, 2u << 20, 512, 8, out);
  run<4, true>(src, 2u << 20, 512, 4, out);
  return 0;
}
