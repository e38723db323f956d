// Micro-benchmark (round 2): how many single-issue instructions hide behind ONE wave's v_mfma_f32_16x16x32_bf16 stream?
// One wave per SIMD (256 workgroups x 4 waves), three rotating accumulators (a dependent MFMA every third issue, as in
// slab_step_p), NF filler instructions after every MFMA, placed with inline asm so that the issue order is the source
// order.  Prints wave cycles per MFMA (s_memtime) for each variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_fill.hip -o /tmp/mfma_fill && /tmp/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
// KIND 0: v_fma_f32 (independent chains)  1: v_and_b32 / v_sub pairs  2: v_exp_f32  3: ds_read_b128 (to a dead register)
template <int KIND>
__device__ __forceinline__ void filler(float& x, float c, unsigned lds_addr, u32x4& sink) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
  if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(lds_addr));
}

template <int NF, int KIND, int NACC>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  __shared__ u32x4 lds[1024];
  lds[threadIdx.x] = u32x4{threadIdx.x, 1, 2, 3};
  __syncthreads();
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a, sink = a;
  f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[4] = {1.f + threadIdx.x, 2.f, 3.f, 4.f};
  const float c = 0.5f;
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + (threadIdx.x & 63));
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      MFMA(acc[j % NACC]);
#pragma unroll
      for (int f = 0; f < NF; ++f) filler<KIND>(x[f & 3], c, addr, sink);
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sink));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = acc[0].x + acc[1].y + acc[2].z + x[0] + x[1] + x[2] + x[3] + __builtin_bit_cast(float, sink.x);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 123.456f) out[1] = 1;
}

template <int NF, int KIND, int NACC>
void run(const char* name, unsigned long long* d) {
  const int iters = 200;
  hipLaunchKernelGGL((k<NF, KIND, NACC>), dim3(256), dim3(256), 0, 0, d, iters);
  hipLaunchKernelGGL((k<NF, KIND, NACC>), dim3(256), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-28s NF=%d NACC=%d: %.2f cycles per MFMA (s_memtime ticks)\n", name, NF, NACC, (double)h / (iters * 18.0));
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  run<0, 0, 3>("bare", d);
  run<0, 0, 1>("bare, one accumulator", d);
  run<0, 0, 9>("bare, nine accumulators", d);
  run<1, 0, 3>("v_fma", d);
  run<2, 0, 3>("v_fma", d);
  run<3, 0, 3>("v_fma", d);
  run<4, 0, 3>("v_fma", d);
  run<6, 0, 3>("v_fma", d);
  run<1, 1, 3>("v_and", d);
  run<2, 1, 3>("v_and", d);
  run<3, 1, 3>("v_and", d);
  run<1, 2, 3>("v_exp", d);
  run<2, 2, 3>("v_exp", d);
  run<1, 3, 3>("ds_read_b128", d);
  run<2, 3, 3>("ds_read_b128", d);
  run<2, 0, 9>("v_fma, nine accumulators", d);
  run<4, 0, 9>("v_fma, nine accumulators", d);
  return 0;
}
