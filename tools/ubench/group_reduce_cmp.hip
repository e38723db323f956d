// sum / max over lanes t, t + 16, t + 32, t + 48 through v_permlane16_swap / v_permlane32_swap (gfx950) against the __shfl_xor
// butterflies of common.h on random floats: bit for bit?  (round 5: yes -- 0 of 262 144 lane-values differ; see common.h for why
// the library keeps the butterflies)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/group_reduce_cmp.hip -o /tmp/grc && /tmp/grc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x2_perm __attribute__((ext_vector_type(2)));
__device__ inline void rows_pair16(float v, float& a, float& b) {
  const u32x2_perm r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned rx = r.x, ry = r.y;        // (not __builtin_bit_cast(float, r.y): that reads element 0)
  a = __builtin_bit_cast(float, rx); b = __builtin_bit_cast(float, ry);
}
__device__ inline void rows_pair32(float v, float& a, float& b) {
  const u32x2_perm r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned rx = r.x, ry = r.y;
  a = __builtin_bit_cast(float, rx); b = __builtin_bit_cast(float, ry);
}
__device__ inline float group_sum(float v) { float a, b; rows_pair16(v, a, b); v = a + b; rows_pair32(v, a, b); return a + b; }
__device__ inline float group_max(float v) { float a, b; rows_pair16(v, a, b); v = fmaxf(a, b); rows_pair32(v, a, b); return fmaxf(a, b); }
__global__ void k(const float* x, int* out, int n) {
  int bad_s = 0, bad_m = 0;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float v = x[i];
    float s0 = v + __shfl_xor(v, 16); s0 = s0 + __shfl_xor(s0, 32);
    float m0 = fmaxf(v, __shfl_xor(v, 16)); m0 = fmaxf(m0, __shfl_xor(m0, 32));
    const float s1 = group_sum(v), m1 = group_max(v);
    bad_s += __float_as_uint(s0) != __float_as_uint(s1);
    bad_m += __float_as_uint(m0) != __float_as_uint(m1);
  }
  atomicAdd(out, bad_s); atomicAdd(out + 1, bad_m);
}
int main() {
  const int n = 64 * 4096;
  float* h = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = ((rand() % 2000001) - 1000000) * 1e-3f * ((i % 7) ? 1.f : 1e-6f);
  float* d; int* o; hipMalloc(&d, n * 4); hipMalloc(&o, 8); hipMemset(o, 0, 8);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  int r[2]; hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
  printf("values per lane %d: group_sum differs in %d, group_max in %d lane-values\n", n / 64, r[0], r[1]);
  return 0;
}
