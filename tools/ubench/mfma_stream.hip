// Micro-benchmark behind DESIGN.md "what the measurements taught": how fast can ONE wave per SIMD (and 2, 3) issue
// v_mfma_f32_16x16x4_f32 when the MFMA batches are interleaved with the weight-fragment loads the kernels use?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_stream.hip -o /tmp/mfma_stream && /tmp/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// VARIANT 0: MFMA only (operands in registers)            -> pipe ceiling for this instruction mix
// VARIANT 1: + double-buffered fragment loads, hard fences (what the kernels do)
// VARIANT 2: same loads, no fences (compiler's own schedule)
// VARIANT 3: loads interleaved one per 4*RT MFMAs with fences around each group
template <int VARIANT, int NT, int RT, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float* __restrict__ w, float* __restrict__ out, int steps) {
  const int lane = threadIdx.x & 63;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(w) + lane;
  f32x4 acc[RT][NT];
  f32x4 x[RT];
  for (int rt = 0; rt < RT; ++rt) {
    x[rt] = f32x4{1.f + lane, 2.f, 3.f, 4.f + rt};
    for (int i = 0; i < NT; ++i) acc[rt][i] = f32x4{0, 0, 0, 0};
  }
  f32x4 wb[2][NT];
  for (int i = 0; i < NT; ++i) wb[0][i] = wp[i * 64];
  for (int i = 0; i < NT; ++i) wb[1][i] = wp[(NT + i) * 64];
#pragma unroll 1
  for (int s = 0; s < steps; s += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (VARIANT == 1 || VARIANT == 2) {
#pragma unroll
        for (int i = 0; i < NT; ++i) wb[h ^ 1][i] = wp[(size_t)(((s + h + 1) & 15) * NT + i) * 64];
      }
      if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
      if (VARIANT == 3) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          wb[h ^ 1][i] = wp[(size_t)(((s + h + 1) & 15) * NT + i) * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(wb[h][i][j], x[rt][j], acc[rt][i]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma4(wb[h][i][j], x[rt][j], acc[rt][i]);
      }
      if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 r = {0, 0, 0, 0};
  for (int rt = 0; rt < RT; ++rt)
    for (int i = 0; i < NT; ++i) r += acc[rt][i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r.x + r.y + r.z + r.w;
}

template <int VARIANT, int NT, int RT, int WPS>
void run(const char* name, const float* w, float* out, int blocks, int steps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VARIANT, NT, RT, WPS>), dim3(blocks), dim3(256), 0, 0, w, out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<VARIANT, NT, RT, WPS>), dim3(blocks), dim3(256), 0, 0, w, out, steps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double mfmas = (double)blocks * 4 * steps * NT * RT * 4;
  const double tf = mfmas * 2048 / (ms * 1e-3) / 1e12;
  printf("%-34s NT=%2d RT=%d WPS=%d blocks=%5d steps=%4d  %8.3f ms  %7.1f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.2GHz)\n", name, NT, RT, WPS,
         blocks, steps, ms, tf, ms * 1e-3 * 2.2e9 / (mfmas / 1024));
}

int main() {
  float *w, *out;
  hipMalloc(&w, 16 * 16 * 64 * 16 * 4);
  hipMemset(w, 0, 16 * 16 * 64 * 16 * 4);
  hipMalloc(&out, 8192 * 256 * 4);
  const int S = 512;
  for (int blocks : {256, 512, 1024}) {
    run<0, 9, 1, 1>("mfma only", w, out, blocks, S);
    run<1, 9, 1, 1>("loads + hard fences", w, out, blocks, S);
    run<2, 9, 1, 1>("loads, compiler schedule", w, out, blocks, S);
    run<3, 9, 1, 1>("loads interleaved per tile", w, out, blocks, S);
    run<0, 9, 2, 1>("mfma only", w, out, blocks, S);
    run<1, 9, 2, 1>("loads + hard fences", w, out, blocks, S);
    run<3, 9, 2, 1>("loads interleaved per tile", w, out, blocks, S);
    run<1, 9, 1, 2>("loads + hard fences (2 w/SIMD)", w, out, blocks, S);
    run<3, 9, 1, 2>("interleaved (2 w/SIMD)", w, out, blocks, S);
    printf("\n");
  }
  // short waves like the FFN: 18 steps per wave, 4000 waves
  run<1, 9, 1, 2>("FFN-like: 18 steps, 1000 blocks", w, out, 1000, 18);
  run<1, 9, 2, 1>("FFN-like RT=2: 18 steps, 500 blk", w, out, 500, 18);
  run<0, 9, 1, 2>("mfma only: 18 steps, 1000 blocks", w, out, 1000, 18);
  // the fused block kernels: 1000 waves (one per SIMD), ~3600 MFMAs each
  run<0, 9, 1, 1>("fused-like: mfma only, 250 blk", w, out, 250, 100);
  run<3, 9, 1, 1>("fused-like: interleaved, 250 blk", w, out, 250, 100);
  run<1, 9, 1, 1>("fused-like: hard fences, 250 blk", w, out, 250, 100);
  run<0, 9, 1, 1>("fused-like: mfma only, 256 blk", w, out, 256, 100);
  run<0, 9, 1, 1>("tiny: 1 step, 250 blk", w, out, 250, 2);
  return 0;
}
