// What v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) return, lane by lane, next to the __shfl_xor butterflies
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/permlane_swap.hip -o /tmp/permlane_swap && /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const unsigned x = 100 + lane;
  const u2 a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const u2 b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[lane * 6 + 0] = a.x; out[lane * 6 + 1] = a.y; out[lane * 6 + 2] = b.x; out[lane * 6 + 3] = b.y;
  out[lane * 6 + 4] = __shfl_xor((int)x, 16); out[lane * 6 + 5] = __shfl_xor((int)x, 32);
}
int main() {
  int* d; hipMalloc(&d, 64 * 6 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[64 * 6]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 8) printf("lane %2d: p16 (%d, %d)  p32 (%d, %d)  xor16 %d xor32 %d\n", l, h[l*6], h[l*6+1], h[l*6+2], h[l*6+3], h[l*6+4], h[l*6+5]);
  return 0;
}
