// Micro-benchmark (round 2): one consumer wave per SIMD running the slab stream of fused.hip's loader-wave kernels from a
// static LDS slab -- 54 v_mfma_f32_16x16x32_bf16 per slab in 3 groups of (3 + 6 + 9), nine ds_read_b128 per group refilled
// in place, s_waitcnt lgkmcnt(6) before the first use of a term -- with and without filler VALU work behind the MFMAs.
// No DMA, no barrier: isolates the matrix pipe / LDS read / VALU issue interplay.   cycles per slab, ideal = 54 x 16.5 = 891.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/slab_stream.hip -o /tmp/slab_stream && /tmp/slab_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MMA(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(W), "v"(X))
#define WMMA(ACC, W0, W1, W2, X) asm volatile("s_waitcnt lgkmcnt(6)\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0" : "+v"(ACC), "+v"(W0), "+v"(W1), "+v"(W2) : "v"(X))
#define RD(W, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(W) : "v"(ADDR), "n"(OFF))
// NF fillers after an MFMA: alternate v_and / v_sub on private registers
template <int NF>
__device__ __forceinline__ void fill(float& a, float& b) {
  if (NF >= 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a));
  if (NF >= 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(b) : "v"(a));
  if (NF >= 3) asm volatile("v_and_b32 %0, 0xfffffff0, %0" : "+v"(a));
}
template <int NF, bool LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned long long* out, int iters) {
  __shared__ u32x4 slab[2 * 1792];
  for (int i = threadIdx.x; i < 2 * 1792; i += blockDim.x) slab[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32x4 w[3][3], x0 = {0x3f803f80u, 0, 0, 0}, x1 = x0, x2 = x0;
  for (int i = 0; i < 3; ++i) for (int t = 0; t < 3; ++t) w[i][t] = x0;
  f32x4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float fa = 1.f + lane, fb = 2.f;
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(slab + lane);
  if (LDS) {
    RD(w[0][2], addr, 2 * 1024); RD(w[1][2], addr, 5 * 1024); RD(w[2][2], addr, 8 * 1024);
    RD(w[0][1], addr, 1 * 1024); RD(w[1][1], addr, 4 * 1024); RD(w[2][1], addr, 7 * 1024);
    RD(w[0][0], addr, 0 * 1024); RD(w[1][0], addr, 3 * 1024); RD(w[2][0], addr, 6 * 1024);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
      f32x4* a = acc + 3 * grp;
      const int o = ((grp + 1) % 3) * 9 * 1024;      // next group's fragments
      if (LDS) WMMA(a[0], w[0][2], w[1][2], w[2][2], x0); else MMA(a[0], w[0][2], x0);
      fill<NF>(fa, fb); MMA(a[1], w[1][2], x0); fill<NF>(fa, fb); MMA(a[2], w[2][2], x0); fill<NF>(fa, fb);
      if (LDS) { RD(w[0][2], addr + o, 2 * 1024); RD(w[1][2], addr + o, 5 * 1024); RD(w[2][2], addr + o, 8 * 1024); }
      if (LDS) WMMA(a[0], w[0][1], w[1][1], w[2][1], x1); else MMA(a[0], w[0][1], x1);
      fill<NF>(fa, fb); MMA(a[1], w[1][1], x1); fill<NF>(fa, fb); MMA(a[2], w[2][1], x1); fill<NF>(fa, fb);
      MMA(a[0], w[0][1], x0); fill<NF>(fa, fb); MMA(a[1], w[1][1], x0); fill<NF>(fa, fb); MMA(a[2], w[2][1], x0); fill<NF>(fa, fb);
      if (LDS) { RD(w[0][1], addr + o, 1 * 1024); RD(w[1][1], addr + o, 4 * 1024); RD(w[2][1], addr + o, 7 * 1024); }
      if (LDS) WMMA(a[0], w[0][0], w[1][0], w[2][0], x2); else MMA(a[0], w[0][0], x2);
      fill<NF>(fa, fb); MMA(a[1], w[1][0], x2); fill<NF>(fa, fb); MMA(a[2], w[2][0], x2); fill<NF>(fa, fb);
      MMA(a[0], w[0][0], x1); fill<NF>(fa, fb); MMA(a[1], w[1][0], x1); fill<NF>(fa, fb); MMA(a[2], w[2][0], x1); fill<NF>(fa, fb);
      MMA(a[0], w[0][0], x0); fill<NF>(fa, fb); MMA(a[1], w[1][0], x0); fill<NF>(fa, fb); MMA(a[2], w[2][0], x0); fill<NF>(fa, fb);
      if (LDS) { RD(w[0][0], addr + o, 0 * 1024); RD(w[1][0], addr + o, 3 * 1024); RD(w[2][0], addr + o, 6 * 1024); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = fa + fb;
  for (int i = 0; i < 9; ++i) s += acc[i].x;
  for (int i = 0; i < 3; ++i) for (int t = 0; t < 3; ++t) s += __builtin_bit_cast(float, w[i][t].x);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 123.456f) out[1] = 1;
}
template <int NF, bool LDS, int WAVES>
void run(const char* name, unsigned long long* d) {
  const int iters = 120;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NF, LDS, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-44s fillers/MFMA=%d waves/CU=%d: %7.1f cycles per slab (54 MFMAs), %.2f per MFMA\n", name, NF, WAVES, (double)h / iters, (double)h / iters / 54);
}
int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 64);
  run<0, false, 4>("MFMA only", d);
  run<2, false, 4>("MFMA + fillers", d);
  run<0, true, 4>("fragment pipeline (in-place refill)", d);
  run<1, true, 4>("fragment pipeline + fillers", d);
  run<2, true, 4>("fragment pipeline + fillers", d);
  run<3, true, 4>("fragment pipeline + fillers", d);
  run<0, true, 1>("fragment pipeline, ONE wave per CU", d);
  run<2, true, 1>("fragment pipeline + fillers, ONE wave per CU", d);
  return 0;
}
