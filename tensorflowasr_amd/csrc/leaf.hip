// LEAF frontend (mel_layer_type 'leaf'): leaf_audio/frontend.py:170-194 as ConformerEncoder builds it
// (conformer_blocks.py:315-317) -- pre-emphasis Conv1D(k=2) -> Gabor complex conv (80 filters x (re, im), k = 401,
// stride 1, SAME) -> squared modulus -> per-channel Gaussian low-pass pooling (k = 401, stride = hop, SAME) ->
// max(., 1e-5) -> PCEN -> instance normalisation over time.
//
// leaf_conv_pool_kernel: the Gabor convolution is a GEMM  [L positions] x [K = 401 taps (26 k-blocks)] x [160 channels],
// 20.5 GFLOP per 10 s utterance -- more than the rest of the ConformerCTC(S) path together -- and its [L, 160] output
// (102 MB per utterance) must never reach HBM.  One workgroup = 128 positions = 4 waves x 2 row tiles (one wave per
// SIMD, two workgroups per CU so that one's barriers fall under the other's MFMAs):
//   * the pre-emphasised, zero-padded signal window of the tile (128 + 416 samples) is staged in LDS once;
//   * the 10 KB weight slab of each k-block goes through a double-buffered LDS stage shared by the four waves
//     (one barrier per k-block), so L2 sees each slab once per workgroup instead of once per wave;
//   * squared modulus in registers: interleaving (re, im) per filter puts both parts of a filter in one lane;
//   * every position contributes to at most three pooled frames; the Gaussian weights are evaluated on the fly
//     (one v_exp per value), reduced over the 16 positions of a tile with shuffles, over the waves through LDS, and
//     written as four partial sums per position tile -- no atomics, the result is deterministic.
// leaf_pcen_norm_kernel: one thread per (utterance, channel): sum of the tile partials that fall on a frame, floor, the
// EMA recurrence over frames, PCEN, then instance-norm statistics and normalisation in a second sweep.
#include "common.h"
#include "launch.h"

namespace {

constexpr int LW = 4;                       // waves per workgroup
constexpr int LTH = LW * 64;
constexpr int KTAPS = 401, KBL = 26;        // taps, k-blocks (416 rows, zero padded)
constexpr int NTL = 10;                     // 160 channels
constexpr int HOP_TILE = 128;               // positions per workgroup (= 4 waves x 2 row tiles of 16)
constexpr int XWIN = HOP_TILE + 16 * KBL;   // staged samples per hop

__global__ __launch_bounds__(LTH, 2) void leaf_conv_pool_kernel(LeafConvArgs a) {
  __shared__ __attribute__((aligned(16))) f32x4 wlds[2][NTL * 64];
  __shared__ float xlds[XWIN + 8];
  __shared__ float red[LW][4][80];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y;
  const int L = a.L, padl = (KTAPS - 1) / 2;
  const float* __restrict__ x = a.wav + (size_t)b * L;
  // ---- stage xp[s] = p0 x[s] + p1 x[s+1] (0 outside [0, L)) for s = 160 h - 200 .. + XWIN
  const int s_base = HOP_TILE * h - padl;
  for (int i = threadIdx.x; i < XWIN; i += LTH) {
    const int s = s_base + i;
    float v = 0.f;
    if (s >= 0 && s < L) v = a.p0 * x[s] + (s + 1 < L ? a.p1 * x[s + 1] : 0.f);
    xlds[i] = v;
  }
  const f32x4* __restrict__ wg = reinterpret_cast<const f32x4*>(a.wp);
  // first weight slab
  for (int i = threadIdx.x; i < NTL * 64; i += LTH) wlds[0][i] = wg[i];
  __syncthreads();

  f32x4 acc[2][NTL];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int i = 0; i < NTL; ++i) acc[rt][i] = splat4(0.f);
  const int pos0 = 32 * wave + c;            // position of row tile 0 inside the hop; tile 1 = +16
#pragma unroll 1
  for (int kb = 0; kb < KBL; ++kb) {
    const int cur = kb & 1;
    // next slab: global -> registers now, -> LDS after the MFMAs (640 fragments / 256 threads: 2.5 per thread)
    f32x4 nw0 = splat4(0.f), nw1 = splat4(0.f), nw2 = splat4(0.f);
    if (kb + 1 < KBL) {
      const f32x4* src = wg + (size_t)(kb + 1) * NTL * 64;
      nw0 = src[threadIdx.x];
      nw1 = src[LTH + threadIdx.x];
      if (threadIdx.x < NTL * 64 - 2 * LTH) nw2 = src[2 * LTH + threadIdx.x];
    }
    f32x4 xf[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float* xp = xlds + pos0 + 16 * rt + 16 * kb + g4;
      xf[rt] = f32x4{xp[0], xp[1], xp[2], xp[3]};
    }
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
      const f32x4 w = wlds[cur][i * 64 + lane];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc[rt][i] = mma_kblock(w, xf[rt], acc[rt][i]);
    }
    if (kb + 1 < KBL) {
      wlds[cur ^ 1][threadIdx.x] = nw0;
      wlds[cur ^ 1][LTH + threadIdx.x] = nw1;
      if (threadIdx.x < NTL * 64 - 2 * LTH) wlds[cur ^ 1][2 * LTH + threadIdx.x] = nw2;
    }
    __syncthreads();
  }

  // ---- squared modulus: lane holds filters fA = 8 nt + 2 g (regs x, y = re, im) and fA + 1 (regs z, w)
  // ---- pooling: position n contributes g_f[tau] * |.|^2 to frame f, tau = n - (hop f - pl) in [0, 401)
  const int fb = (HOP_TILE * h + a.pl) / a.hop - 2;   // first of the four frames this tile can touch
  for (int rel = 0; rel < 4; ++rel) {
    const int f = fb + rel;
    float wsum[2 * NTL];
#pragma unroll
    for (int i = 0; i < 2 * NTL; ++i) wsum[i] = 0.f;
    if (f >= 0 && f < a.F) {                 // uniform
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int n = HOP_TILE * h + pos0 + 16 * rt;
        const int tau = n - (a.hop * f - a.pl);
        const bool ok = (n < L) && tau >= 0 && tau < KTAPS;
        const float t2 = (float)(tau - (KTAPS - 1) / 2) * (float)(tau - (KTAPS - 1) / 2);
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
          const f32x4 v = acc[rt][i];
          const float sa = v.x * v.x + v.y * v.y, sb = v.z * v.z + v.w * v.w;
          const float ca = a.gcoef[8 * i + 2 * g], cb = a.gcoef[8 * i + 2 * g + 1];      // -0.5 log2(e) / (sigma 200)^2
          wsum[2 * i] += ok ? sa * __builtin_amdgcn_exp2f(ca * t2) : 0.f;
          wsum[2 * i + 1] += ok ? sb * __builtin_amdgcn_exp2f(cb * t2) : 0.f;
        }
      }
    }
    // sum over the 16 positions of the tile (lanes c)
#pragma unroll
    for (int i = 0; i < 2 * NTL; ++i) {
      float v = wsum[i];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
      wsum[i] = v;
    }
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < NTL; ++i) {
        red[wave][rel][8 * i + 2 * g] = wsum[2 * i];
        red[wave][rel][8 * i + 2 * g + 1] = wsum[2 * i + 1];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 320; i += LTH) {
    const int rel = i / 80, ch = i % 80;
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < LW; ++w2) s += red[w2][rel][ch];
    a.part[(((size_t)b * a.NH + h) * 4 + rel) * 80 + ch] = s;
  }
}

// pooled[b][f][ch] = sum of the hop partials that map to frame f; PCEN; instance norm.  One thread per (b, ch).
__global__ __launch_bounds__(128) void leaf_pcen_norm_kernel(LeafPcenArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * 80) return;
  const int b = idx / 80, ch = idx % 80;
  const float alpha = fminf(a.alpha[ch], 1.0f), inv_root = 1.0f / fmaxf(a.root[ch], 1.0f), delta = a.delta[ch];
  const float sm = fminf(fmaxf(a.smooth[ch], 0.f), 1.f);
  const float dr = __powf(delta, inv_root);
  const float* __restrict__ part = a.part + (size_t)b * a.NH * 320;
  float* out = a.out + (size_t)b * a.F * 80 + ch;
  float state = 0.f, sum = 0.f;
  for (int f = 0; f < a.F; ++f) {
    // tiles whose four-frame range [fb, fb + 3], fb = (128 h + pl) / hop - 2, contains f (ascending order: deterministic)
    float p = 0.f;
    const int h_lo = max((a.hop * (f - 1) - a.pl) / HOP_TILE - 1, 0);
    const int h_hi = min((a.hop * (f + 3) - a.pl) / HOP_TILE + 1, a.NH - 1);
    for (int h = h_lo; h <= h_hi; ++h) {
      const int rel = f - ((HOP_TILE * h + a.pl) / a.hop - 2);
      if (rel >= 0 && rel < 4) p += part[((size_t)h * 4 + rel) * 80 + ch];
    }
    p = fmaxf(p, 1e-5f);
    state = (f == 0) ? p : sm * p + (1.0f - sm) * state;           // EMA, initial state = frame 0
    const float v = __powf(p / __powf(1e-12f + state, alpha) + delta, inv_root) - dr;
    out[(size_t)f * 80] = v;
    sum += v;
  }
  const float mean = sum / (float)a.F;
  float qq = 0.f;
  for (int f = 0; f < a.F; ++f) { const float d = out[(size_t)f * 80] - mean; qq += d * d; }
  const float rstd = 1.0f / sqrtf(qq / (float)a.F + 1e-6f);
  const float ga = a.gamma[ch], be = a.beta[ch];
  for (int f = 0; f < a.F; ++f) out[(size_t)f * 80] = (out[(size_t)f * 80] - mean) * rstd * ga + be;
}

}  // namespace

int launch_leaf_conv_pool(const LeafConvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(leaf_conv_pool_kernel, dim3(a.NH, a.B), dim3(LTH), 0, s, a);
  return 0;
}
int launch_leaf_pcen_norm(const LeafPcenArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(leaf_pcen_norm_kernel, dim3((a.B * 80 + 127) / 128), dim3(128), 0, s, a);
  return 0;
}
