// LEAF frontend (mel_layer_type 'leaf'): leaf_audio/frontend.py:170-194 as ConformerEncoder builds it
// (conformer_blocks.py:315-317) -- pre-emphasis Conv1D(k=2) -> Gabor complex conv (80 filters x (re, im), k = 401,
// stride 1, SAME) -> squared modulus -> per-channel Gaussian low-pass pooling (k = 401, stride = hop, SAME) ->
// max(., 1e-5) -> PCEN -> instance normalisation over time.
//
// Two kernels compute the Gabor convolution + pooling: leaf_conv_pool_split_kernel (default; fp32 operands as sums of
// bf16 terms on the bf16 MFMA, further down) and leaf_conv_pool_kernel (fp32 MFMA; MI355ASR_LEAF_TERMS=0).
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
// leaf_gather_kernel / leaf_pcen_norm_kernel: sum of the tile partials that fall on a frame + floor (one thread per
// element), then per utterance and channel group the EMA recurrence over frames (the only serial part), PCEN,
// instance-norm statistics and normalisation.
#include "common.h"
#include "launch.h"

namespace {

constexpr int LW = 4;                       // waves per workgroup
constexpr int LTH = LW * 64;
constexpr int KTAPS = 401, KBL = 26;        // taps, k-blocks (416 rows, zero padded)
constexpr int NTL = 10;                     // 160 channels
constexpr int HOP_TILE = 128;               // positions per workgroup (= 4 waves x 2 row tiles of 16)
constexpr int XWIN = HOP_TILE + 16 * KBL;   // staged samples per hop

// sum over the 16 lanes of a row (the 16 positions of a row tile) with DPP adds; every lane ends with the total
template <int CTRL>
DEV float dpp_add(float v) {
  const int s = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, s);
}
DEV float row16_sum(float v) {
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  return dpp_add<0x140>(v); // row_mirror
}

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
    for (int i = 0; i < 2 * NTL; ++i) wsum[i] = row16_sum(wsum[i]);
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


// ---------------------------------------------------------------------------------------------------------
// Split-bf16 variant of the Gabor convolution: fp32 operands are written as sums of NS bf16 terms
// (x = x0 + x1 (+ x2), each the round-to-nearest bf16 of the remainder; three terms hold all 24 mantissa bits), the
// products x_i w_j are exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16, and the terms with i + j < NS are
// summed smallest first.  NS = 3 (6 MFMAs per 32 taps): the dropped terms are below 2^-24 of the product -- the result
// is as accurate as an fp32 FMA chain (measured against the fp64 oracle: 8e-9 of the output range; an fp32 chain over
// 401 taps is ~1e-6).  NS = 2 (3 MFMAs): 6e-6.  One bf16 MFMA retires 32 taps in ~17 cycles where the fp32 MFMA needs
// 8 x 32, so 6 of them are 2.5x the fp32 rate.
// The signal window is split once while staging and stored as bf16 in 8 copies shifted by 0..7 samples, so that the 8 consecutive
// samples a lane needs (start = position + tap offset, any alignment) are one aligned ds_read_b128 from copy
// (position & 7); the copy stride is 32 bytes mod 256, which spreads the 16 positions of a row tile over all banks.
constexpr int SW = 8, STH = SW * 64, SPOS = kLeafSplitTile, KB32 = 13, NREL_S = kLeafSplitSlots;
static_assert(SPOS == 64 * SW, "one wave per 64 positions");
constexpr int XSTAGE = SPOS + 32 * KB32 + 8;  // staged samples
constexpr int XSTRIDE = 2080;               // bytes between shifted copies (>= 2 * XSTAGE + 16, = 32 mod 256)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEV unsigned bf16_rne_bits(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// Workgroup = 8 waves x 64 positions (4 row tiles each), one workgroup per CU (122 KB of LDS for NS = 3, two waves
// per SIMD); per 32 taps a 30 KB weight slab is double-buffered in LDS (global -> registers during the MFMAs of the
// previous slab, -> LDS before the barrier).  A variant with 4-wave workgroups, two per CU (half slabs, 74 KB) measured
// 5 % slower: with one wave per SIMD and workgroup the barrier makes every workgroup run at the pace of its slowest
// wave (SQ_WAIT_ANY 33 %, MFMA pipe busy 56 %), which costs more than the overlap of staging / epilogue gains.
template <int NS>
__global__ __launch_bounds__(STH, 2) void leaf_conv_pool_split_kernel(LeafConvArgs a) {
  constexpr int SLAB = NTL * NS * 64;        // 16-byte fragments per 32 taps
  constexpr int NQ = (SLAB + STH - 1) / STH;
  __shared__ __attribute__((aligned(16))) u32x4 wlds[2][SLAB];
  __shared__ __attribute__((aligned(16))) unsigned char xlds[NS * 8 * XSTRIDE];
  __shared__ float red[SW][4][80];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y;
  const int L = a.L, padl = (KTAPS - 1) / 2;
  const float* __restrict__ x = a.wav + (size_t)b * L;
  const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(a.wp);
  // first weight slab: global -> registers
  u32x4 nw[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int idx = threadIdx.x + STH * q;
    nw[q] = u32x4{0u, 0u, 0u, 0u};
    if (idx < SLAB) nw[q] = wg[idx];
  }
  // ---- stage xp[s] = p0 x[s] + p1 x[s+1] for s = SPOS h - 200 .. + XSTAGE, split, 8 shifted copies per term
  const int s_base = SPOS * h - padl;
  for (int i = threadIdx.x; i < XSTAGE; i += STH) {
    const int s = s_base + i;
    float r = 0.f;
    if (s >= 0 && s < L) r = a.p0 * x[s] + (s + 1 < L ? a.p1 * x[s + 1] : 0.f);
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      const unsigned hb = bf16_rne_bits(r);
      r -= __builtin_bit_cast(float, hb << 16);
#pragma unroll
      for (int cp = 0; cp < 8; ++cp)           // copy cp, element i - cp  holds sample i
        if (i - cp >= 0) *reinterpret_cast<unsigned short*>(xlds + (sp * 8 + cp) * XSTRIDE + 2 * (i - cp)) = (unsigned short)hb;
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int idx = threadIdx.x + STH * q;
    if (idx < SLAB) wlds[0][idx] = nw[q];
  }
  __syncthreads();

  f32x4 acc[4][NTL];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int i = 0; i < NTL; ++i) acc[rt][i] = splat4(0.f);
  const unsigned char* xb = xlds + (c & 7) * XSTRIDE + 2 * (64 * wave + (c & ~7) + 8 * g);
#pragma unroll 1
  for (int kb = 0; kb < KB32; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < KB32) {
      const u32x4* src = wg + (size_t)(kb + 1) * SLAB;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = threadIdx.x + STH * q;
        if (idx < SLAB) nw[q] = src[idx];
      }
    }
    bf16x8 xf[4][NS];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int sp = 0; sp < NS; ++sp)
        xf[rt][sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + sp * 8 * XSTRIDE + 32 * rt + 64 * kb));
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
      bf16x8 wf[NS];
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) wf[sp] = __builtin_bit_cast(bf16x8, wlds[cur][(i * NS + sp) * 64 + lane]);
      // terms x_p w_q with p + q = ord, smallest first
#pragma unroll
      for (int ord = NS - 1; ord >= 0; --ord)
#pragma unroll
        for (int p = 0; p <= ord; ++p)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ord - p], xf[rt][p], acc[rt][i], 0, 0, 0);
    }
    if (kb + 1 < KB32) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = threadIdx.x + STH * q;
        if (idx < SLAB) wlds[cur ^ 1][idx] = nw[q];
      }
    }
    __syncthreads();
  }

  // ---- squared modulus + Gaussian pooling: the wave's 64 positions touch frames fbw .. fbw + 3.
  // lane holds filters 8 nt + 2 g (x, y = re, im) and 8 nt + 2 g + 1 (z, w); |.|^2 once, then one mul + v_exp + fma per
  // (value, frame): positions outside [0, L) or outside the frame's window get t^2 = 3e38, i.e. weight exp2(-huge) = 0
  const int n_w = SPOS * h + 64 * wave;
  const int fbw = (n_w + a.pl) / a.hop - 2;
  float sq[4][2 * NTL], gc[2 * NTL];
#pragma unroll
  for (int i = 0; i < NTL; ++i) {
    gc[2 * i] = a.gcoef[8 * i + 2 * g];        // -0.5 log2(e) / (sigma 200)^2  (< 0)
    gc[2 * i + 1] = a.gcoef[8 * i + 2 * g + 1];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const f32x4 v = acc[rt][i];
      sq[rt][2 * i] = v.x * v.x + v.y * v.y;
      sq[rt][2 * i + 1] = v.z * v.z + v.w * v.w;
    }
  }
  for (int rel = 0; rel < 4; ++rel) {
    const int f = fbw + rel;
    float wsum[2 * NTL];
#pragma unroll
    for (int i = 0; i < 2 * NTL; ++i) wsum[i] = 0.f;
    if (f >= 0 && f < a.F) {                 // uniform
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int tau0 = n_w + 16 * rt - (a.hop * f - a.pl);        // tau of the tile's first position (uniform)
        if (tau0 + 15 < 0 || tau0 >= KTAPS) continue;
        const int n = n_w + 16 * rt + c, tau = tau0 + c;
        const bool ok = (n < L) && tau >= 0 && tau < KTAPS;
        const float t = (float)(tau - (KTAPS - 1) / 2);
        const float t2 = ok ? t * t : 3.0e38f;
#pragma unroll
        for (int i = 0; i < 2 * NTL; ++i) wsum[i] = __builtin_fmaf(sq[rt][i], __builtin_amdgcn_exp2f(gc[i] * t2), wsum[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2 * NTL; ++i) wsum[i] = row16_sum(wsum[i]);
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < NTL; ++i) {
        red[wave][rel][8 * i + 2 * g] = wsum[2 * i];
        red[wave][rel][8 * i + 2 * g + 1] = wsum[2 * i + 1];
      }
    }
  }
  __syncthreads();
  // the tile's NREL_S frame slots start at FB; wave w2 covers slots fbw(w2) - FB .. + 3 (fixed order: deterministic)
  const int FB = (SPOS * h + a.pl) / a.hop - 2;
  for (int i = threadIdx.x; i < NREL_S * 80; i += STH) {
    const int slot = i / 80, ch = i % 80;
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < SW; ++w2) {
      const int rel = slot - ((SPOS * h + 64 * w2 + a.pl) / a.hop - 2 - FB);
      if (rel >= 0 && rel < 4) s += red[w2][rel][ch];
    }
    a.part[(((size_t)b * a.NH + h) * NREL_S + slot) * 80 + ch] = s;
  }
}

// pooled[b][f][ch] = sum of the tile partials that map to frame f; floor; PCEN; instance norm.
// One workgroup per utterance and group of 16 channels, 20 frame rows x 16 channels of threads, frames in chunks of PCH
// through LDS:
//   (leaf_gather_kernel, one thread per element, has summed the tiles whose frame range [fb, fb + nrel - 1],
//    fb = (tile h + pl) / hop - 2, contains f, in ascending tile order -- deterministic -- and applied the floor;)
//   EMA (the only serial part: one FMA per frame, the threads of row 0, state carried across chunks in a register);
//   PCEN (parallel), written to `out`; per-channel sums for the instance norm are reduced in a fixed order.
// Then two more parallel sweeps over out[b]: variance about the mean (two-pass, as the reference), normalisation.
constexpr int PCH = 128, PCG = 16, PROWS = 20, PTH = PCG * PROWS;   // frames per chunk, channels / frame rows per workgroup

// out[b][f][ch] = max(sum of the partials of frame f, 1e-5): one thread per element
__global__ __launch_bounds__(256) void leaf_gather_kernel(LeafPcenArgs a) {
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= (size_t)a.B * a.F * 80) return;
  const int ch = (int)(idx % 80), f = (int)((idx / 80) % a.F), b = (int)(idx / ((size_t)80 * a.F));
  const int tile = a.tile, nrel = a.nrel;
  const float* __restrict__ part = a.part + (size_t)b * a.NH * nrel * 80;
  float p = 0.f;
  const int h_lo = max((a.hop * (f + 3 - nrel) - a.pl) / tile - 1, 0);
  const int h_hi = min((a.hop * (f + 3) - a.pl) / tile + 1, a.NH - 1);
  for (int h = h_lo; h <= h_hi; ++h) {
    const int rel = f - ((tile * h + a.pl) / a.hop - 2);
    if (rel >= 0 && rel < nrel) p += part[((size_t)h * nrel + rel) * 80 + ch];
  }
  a.out[idx] = fmaxf(p, 1e-5f);
}

__global__ __launch_bounds__(PTH) void leaf_pcen_norm_kernel(LeafPcenArgs a) {
  __shared__ float pool[PCH * PCG];
  __shared__ float ema[PCH * PCG];
  __shared__ float red[PROWS][PCG];
  __shared__ float stat[2][PCG];
  const int b = blockIdx.x, cl = threadIdx.x % PCG, ch = blockIdx.y * PCG + cl, r = threadIdx.x / PCG;
  const float alpha = fminf(a.alpha[ch], 1.0f), inv_root = 1.0f / fmaxf(a.root[ch], 1.0f), delta = a.delta[ch];
  const float sm = fminf(fmaxf(a.smooth[ch], 0.f), 1.f);
  const float dr = __powf(delta, inv_root);
  float* out = a.out + (size_t)b * a.F * 80;
  float state = 0.f, sum = 0.f;
  for (int f0 = 0; f0 < a.F; f0 += PCH) {
    const int n = min(PCH, a.F - f0);
#pragma unroll 4
    for (int j = r; j < n; j += PROWS) pool[j * PCG + cl] = out[(size_t)(f0 + j) * 80 + ch];     // gathered + floored
    __syncthreads();
    if (r == 0) {
#pragma unroll 8
      for (int j = 0; j < n; ++j) {
        const float p = pool[j * PCG + cl];
        state = (f0 + j == 0) ? p : sm * p + (1.0f - sm) * state;          // EMA, initial state = frame 0
        ema[j * PCG + cl] = state;
      }
    }
    __syncthreads();
    for (int j = r; j < n; j += PROWS) {
      const float v = __powf(pool[j * PCG + cl] / __powf(1e-12f + ema[j * PCG + cl], alpha) + delta, inv_root) - dr;
      out[(size_t)(f0 + j) * 80 + ch] = v;
      sum += v;
    }
    __syncthreads();
  }
  red[r][cl] = sum;
  __syncthreads();
  if (r == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PROWS; ++i) s += red[i][cl];
    stat[0][cl] = s / (float)a.F;
  }
  __syncthreads();
  const float mean = stat[0][cl];
  float qq = 0.f;
#pragma unroll 8
  for (int f = r; f < a.F; f += PROWS) { const float d = out[(size_t)f * 80 + ch] - mean; qq += d * d; }
  red[r][cl] = qq;
  __syncthreads();
  if (r == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PROWS; ++i) s += red[i][cl];
    stat[1][cl] = 1.0f / sqrtf(s / (float)a.F + 1e-6f);
  }
  __syncthreads();
  const float rstd = stat[1][cl], ga = a.gamma[ch], be = a.beta[ch];
#pragma unroll 8
  for (int f = r; f < a.F; f += PROWS) out[(size_t)f * 80 + ch] = (out[(size_t)f * 80 + ch] - mean) * rstd * ga + be;
}

}  // namespace

int launch_leaf_conv_pool(const LeafConvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(leaf_conv_pool_kernel, dim3(a.NH, a.B), dim3(LTH), 0, s, a);
  return 0;
}
int launch_leaf_conv_pool_split(int ns, const LeafConvArgs& a, hipStream_t s) {
  note_scheme(SCHEME_BF16X3);   // ns bf16 terms per operand (2: the Gabor filters' third term is below fp32 resolution)
  if (ns == 3) hipLaunchKernelGGL(leaf_conv_pool_split_kernel<3>, dim3(a.NH, a.B), dim3(STH), 0, s, a);
  else if (ns == 2) hipLaunchKernelGGL(leaf_conv_pool_split_kernel<2>, dim3(a.NH, a.B), dim3(STH), 0, s, a);
  else return -1;
  return 0;
}
int launch_leaf_pcen_norm(const LeafPcenArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * a.F * 80;
  hipLaunchKernelGGL(leaf_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(leaf_pcen_norm_kernel, dim3(a.B, 80 / PCG), dim3(PTH), 0, s, a);
  return 0;
}
