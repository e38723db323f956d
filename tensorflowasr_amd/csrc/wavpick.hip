// add_wav_info branch: WavePickModel (asr/models/wav_model.py:108-146) -- a strided 1-D conv stack on the raw waveform
// whose [B, L / hop, dmodel] output is added to the subsampled frontend features (conformer_blocks.py:344-348).
// The channels-last Conv1D layers are GEMMs on overlapping rows of a padded copy of their input (row t = the k*Cin
// contiguous floats starting at frame t*stride), so they run on gemm16_kernel<PF32> (bf16.hip) with a row stride of
// stride*Cin; the two small kernels here are the first layer (SeparableConv1D on one channel) and the padded copy
// (zero or reflect padding, LeakyReLU, optional sum of the two branches of the preceding residual stack).
#include <algorithm>

#include "common.h"
#include "launch.h"

namespace {

__global__ __launch_bounds__(256) void wp_sepconv_kernel(WpSepConvArgs a) {
  const size_t total = (size_t)a.B * a.T0 * 8;       // 8 float4 = 32 channels
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i & 7) * 4;
    const size_t bt = i >> 3;
    const int t = (int)(bt % a.T0), b = (int)(bt / a.T0);
    const float* x = a.wav + (size_t)b * a.L;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int s = t * a.stride + j - a.pad_left;
      d += (s >= 0 && s < a.L) ? a.dw[j] * x[s] : 0.f;
    }
    f32x4 v = splat4(d) * ldg4(a.pw + c4) + ldg4(a.bias + c4);
    v.x = v.x >= 0.f ? v.x : a.slope * v.x; v.y = v.y >= 0.f ? v.y : a.slope * v.y;
    v.z = v.z >= 0.f ? v.z : a.slope * v.z; v.w = v.w >= 0.f ? v.w : a.slope * v.w;
    stg4(a.out + bt * 32 + c4, v);
  }
}

__global__ __launch_bounds__(256) void wp_pad_act_kernel(WpPadActArgs a) {
  const int c4n = a.C / 4, Tp = a.T + a.lo + a.hi;
  const size_t total = (size_t)a.B * Tp * c4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n) * 4;
    const size_t bt = i / c4n;
    const int tp = (int)(bt % Tp), b = (int)(bt / Tp);
    int t = tp - a.lo;
    f32x4 v = splat4(0.f);
    bool inside = t >= 0 && t < a.T;
    if (!inside && a.reflect) { t = t < 0 ? -t : 2 * (a.T - 1) - t; inside = t >= 0 && t < a.T; }   // tf.pad REFLECT
    if (inside) {
      const size_t o = ((size_t)b * a.T + t) * a.C + c4;
      v = ldg4(a.src + o);
      if (a.src2) v += ldg4(a.src2 + o);
      v.x = v.x >= 0.f ? v.x : a.slope * v.x; v.y = v.y >= 0.f ? v.y : a.slope * v.y;
      v.z = v.z >= 0.f ? v.z : a.slope * v.z; v.w = v.w >= 0.f ? v.w : a.slope * v.w;
    }
    stg4(a.dst + bt * a.C + c4, v);
  }
}

}  // namespace

int launch_wp_sepconv(const WpSepConvArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * a.T0 * 8;
  if (total == 0) return 0;
  hipLaunchKernelGGL(wp_sepconv_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, a);
  return 0;
}
int launch_wp_pad_act(const WpPadActArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * (a.T + a.lo + a.hi) * (a.C / 4);
  if (total == 0 || a.C % 4 != 0) return a.C % 4 != 0 ? -1 : 0;
  hipLaunchKernelGGL(wp_pad_act_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, a);
  return 0;
}
