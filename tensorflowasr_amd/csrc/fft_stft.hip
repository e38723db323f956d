// STFT power spectrum through a 32 x 32 Cooley-Tukey factorisation of the 1024-point DFT, both stages on the
// fp32 matrix cores.  Same result as stft_kernel (the reference's dense DFT convolution,
// asr/models/layers/time_frequency.py:100-122) with 128 instead of 1040 MFMAs per frame; it is only selected when
// the model's DFT kernels are window[n] * exp(-2*pi*i*k*n/1024) (checked at mi355asr_finalize_weights: the
// kernels are Keras variables and may come from a checkpoint; anything else runs the dense kernel).
//
//   n = 32*n1 + n2,  k = k1 + 32*k2:
//   A[k1][n2] = sum_n1 xw[32 n1 + n2] W32^(n1 k1)            stage 1: tokens = n2 (two 16-tiles), K = n1 (32)
//   B[k1][n2] = A[k1][n2] * W1024^(n2 k1)                    twiddle, in registers
//   X[k1 + 32 k2] = sum_n2 B[k1][n2] W32^(n2 k2)             stage 2: tokens = k1, K = (re,im) of n2 (64), k2 < 16
// One wave owns whole frames; the 32 x 32 complex transpose between the stages goes through 9 KB of wave-private
// LDS (no block barrier).  Bin 512 (k1 = 0, k2 = 16) is the alternating sum of A[0][:].
// The stage weights (cos/sin of 32-point DFTs), the twiddles and the window live in registers for the whole kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "env.h"
#include "launch.h"

namespace {

constexpr int LDW = 36;   // LDS row stride in floats (16-byte aligned rows, 4 floats of padding)

__global__ __launch_bounds__(BLOCK_THREADS, 2) void fft_stft_kernel(FftStftArgs a) {
  __shared__ float lds[WAVES_PER_BLOCK][2][32 * LDW];
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int wave = threadIdx.x >> 6;
  float* Lre = lds[wave][0];
  float* Lim = lds[wave][1];
  const int wid = blockIdx.x * WAVES_PER_BLOCK + wave;
  const int nwaves = gridDim.x * WAVES_PER_BLOCK;
  const int total = a.B * a.F;

  // ---- frame-independent operands -> registers
  const f32x4* __restrict__ w1p = reinterpret_cast<const f32x4*>(a.w1p) + lane;   // [2 kb][4 nt]
  const f32x4* __restrict__ w2p = reinterpret_cast<const f32x4*>(a.w2p) + lane;   // [4 kb][2 nt]
  f32x4 w1[2][4], w2[4][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w1[kb][nt] = w1p[(kb * 4 + nt) * 64];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w2[kb][nt] = w2p[(kb * 2 + nt) * 64];
  f32x4 hw[2][2], twc[2][2], tws[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int n2 = 16 * rt + c;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // window at n = 32*n1 + n2, n1 = 16*kb + 4*g + j
      hw[rt][kb].x = a.window[32 * (16 * kb + g4 + 0) + n2];
      hw[rt][kb].y = a.window[32 * (16 * kb + g4 + 1) + n2];
      hw[rt][kb].z = a.window[32 * (16 * kb + g4 + 2) + n2];
      hw[rt][kb].w = a.window[32 * (16 * kb + g4 + 3) + n2];
      // twiddle W1024^(n2*k1), k1 = 16*kb + 4*g + j  (kb doubles as the re-tile index here)
      twc[rt][kb].x = a.tw_c[(16 * kb + g4 + 0) * 32 + n2]; tws[rt][kb].x = a.tw_s[(16 * kb + g4 + 0) * 32 + n2];
      twc[rt][kb].y = a.tw_c[(16 * kb + g4 + 1) * 32 + n2]; tws[rt][kb].y = a.tw_s[(16 * kb + g4 + 1) * 32 + n2];
      twc[rt][kb].z = a.tw_c[(16 * kb + g4 + 2) * 32 + n2]; tws[rt][kb].z = a.tw_s[(16 * kb + g4 + 2) * 32 + n2];
      twc[rt][kb].w = a.tw_c[(16 * kb + g4 + 3) * 32 + n2]; tws[rt][kb].w = a.tw_s[(16 * kb + g4 + 3) * 32 + n2];
    }
  }
  const int L = a.L;

#pragma unroll 1
  for (int fidx = wid; fidx < total; fidx += nwaves) {
    const int b = fidx / a.F, f = fidx - b * a.F;
    const float* __restrict__ wav = a.wav + (size_t)b * L;
    const int base = f * a.hop - a.pad_left;
    // ---- stage-1 operand: xw[32*n1 + n2], zero outside the signal (TF SAME / left-padded VALID framing)
    f32x4 xf[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int s0 = base + 32 * (16 * kb + g4) + 16 * rt + c;
        const bool o0 = (unsigned)(s0) < (unsigned)L, o1 = (unsigned)(s0 + 32) < (unsigned)L;
        const bool o2 = (unsigned)(s0 + 64) < (unsigned)L, o3 = (unsigned)(s0 + 96) < (unsigned)L;
        f32x4 v;
        v.x = wav[o0 ? s0 : 0]; v.y = wav[o1 ? s0 + 32 : 0]; v.z = wav[o2 ? s0 + 64 : 0]; v.w = wav[o3 ? s0 + 96 : 0];
        v.x = o0 ? v.x : 0.f; v.y = o1 ? v.y : 0.f; v.z = o2 ? v.z : 0.f; v.w = o3 ? v.w : 0.f;
        xf[rt][kb] = v * hw[rt][kb];
      }
    // ---- stage 1: A = F32 * Xw   (acc1[rt][nt]: nt 0,1 = Re k1 0..31 ; nt 2,3 = Im k1 0..31)
    f32x4 acc1[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc1[rt][nt] = splat4(0.f);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc1[rt][nt] = mfma4(w1[kb][nt][j], xf[rt][kb][j], acc1[rt][nt]);
    // ---- bin 512: sum_n2 (-1)^n2 A_re[0][n2]   (A_re[0][n2] sits in lanes g == 0, tile 0, reg 0)
    float nyq = (g == 0) ? ((c & 1) ? -1.f : 1.f) * (acc1[0][0].x + acc1[1][0].x) : 0.f;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) nyq += __shfl_xor(nyq, off);
    // ---- twiddle + transpose through wave-private LDS: L[k1][n2]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int n2 = 16 * rt + c;
#pragma unroll
      for (int nr = 0; nr < 2; ++nr) {
        const f32x4 are = acc1[rt][nr], aim = acc1[rt][2 + nr];
        const f32x4 bre = are * twc[rt][nr] + aim * tws[rt][nr];
        const f32x4 bim = aim * twc[rt][nr] - are * tws[rt][nr];
        const int k1 = 16 * nr + g4;
        Lre[(k1 + 0) * LDW + n2] = bre.x; Lre[(k1 + 1) * LDW + n2] = bre.y;
        Lre[(k1 + 2) * LDW + n2] = bre.z; Lre[(k1 + 3) * LDW + n2] = bre.w;
        Lim[(k1 + 0) * LDW + n2] = bim.x; Lim[(k1 + 1) * LDW + n2] = bim.y;
        Lim[(k1 + 2) * LDW + n2] = bim.z; Lim[(k1 + 3) * LDW + n2] = bim.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- stage 2: X[k1][k2] = sum_n2 B[k1][n2] W32^(n2 k2); tokens = k1 = 16*rt + c, K = [Re n2 | Im n2]
    f32x4 acc2[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      acc2[rt][0] = splat4(0.f);
      acc2[rt][1] = splat4(0.f);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 yf[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float* src = (kb < 2 ? Lre : Lim) + (16 * rt + c) * LDW + 16 * (kb & 1) + g4;
        yf[rt] = *reinterpret_cast<const f32x4*>(src);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc2[rt][nt] = mfma4(w2[kb][nt][j], yf[rt][j], acc2[rt][nt]);
    }
    // ---- power, log, store, per-frame max.  lane holds bins k1 + 32*k2, k1 = 16*rt + c, k2 = 4*g + j
    float mx = -INFINITY;
    float* orow = a.logp + ((size_t)b * a.F + f) * a.LP;
    const float kscale = a.db10 ? (10.0f * 0.69314718f / 2.30258509f) : (0.69314718f / 2.30258509f);   // log2 -> dB / log10
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const f32x4 re = acc2[rt][0], im = acc2[rt][1];
      const f32x4 p = re * re + im * im;
      const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float l = __log2f(fmaxf(pv[j], 1e-10f)) * kscale;
        orow[16 * rt + c + 32 * (g4 + j)] = l;
        mx = fmaxf(mx, l);
      }
    }
    if (lane == 0) {
      const float l = __log2f(fmaxf(nyq * nyq, 1e-10f)) * kscale;
      orow[512] = l;
      mx = fmaxf(mx, l);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) a.pmax[fidx] = mx;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same factorisation on the bf16 matrix pipe (round 3): both DFT-32 stages as v_mfma_f32_16x16x32_bf16 over operands
// split exactly into three bf16 terms (the six term pairs with i + j <= 2, smallest first: as accurate as the fp32 FMA chain,
// see subconv.hip / leaf.hip).  Stage 1 is ONE 32-wide step (K = n1), stage 2 two (K = re | im of n2): 96 MFMAs of 16 cycles
// per frame instead of 128 of 32 -- 1536 matrix-pipe cycles instead of 4096.  The k-slot order of a step (pack_split32:
// slot j of lane group g <-> k = 16 (j >> 2) + 4 g + (j & 3)) is exactly the pair of float4 fragments the fp32 kernel feeds
// to its two 16-wide k-blocks, so loads, window, twiddles, the LDS transpose and the epilogue are unchanged; the stage
// matrices are split on the host (cos / sin of multiples of 2 pi / 32 in fp32, three terms hold every bit).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct Split8 { u32x4_t t[3]; };
DEV Split8 split8(f32x4 lo, f32x4 hi) {      // exact: x = t0 + t1 + t2 (truncation; the remainders are exact)
  float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Split8 f;
#pragma unroll
  for (int term = 0; term < 3; ++term) {
    unsigned d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned a0 = __builtin_bit_cast(unsigned, v[2 * k]), a1 = __builtin_bit_cast(unsigned, v[2 * k + 1]);
      d[k] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
      if (term < 2) {
        v[2 * k] -= __builtin_bit_cast(float, a0 & 0xffff0000u);
        v[2 * k + 1] -= __builtin_bit_cast(float, a1 & 0xffff0000u);
      }
    }
    f.t[term] = u32x4_t{d[0], d[1], d[2], d[3]};
  }
  return f;
}
DEV f32x4 mma32(u32x4_t a, u32x4_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// Two-term scheme (TM = 2 below): the stage matrices (cos / sin: |w| <= 1) as hi + lo fp16 of w * 2^14,
// an operand column (one n2 of stage 1, one k1 of stage 2: this lane's token, spread over the four lanes c, c + 16, c + 32,
// c + 48) as hi + lo fp16 of the column times the power of two of its largest magnitude; three products per fragment pair.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr float kStageScale = 16384.f;
DEV unsigned pk_f16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t)); }
DEV Split8 split8h(f32x4 lo, f32x4 hi) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned d0[4], d1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d0[k] = pk_f16(v[2 * k], v[2 * k + 1]);
    const f16x2_t h = __builtin_bit_cast(f16x2_t, d0[k]);
    d1[k] = pk_f16(v[2 * k] - (float)h.x, v[2 * k + 1] - (float)h.y);
  }
  Split8 f;
  f.t[0] = u32x4_t{d0[0], d0[1], d0[2], d0[3]};
  f.t[1] = u32x4_t{d1[0], d1[1], d1[2], d1[3]};
  f.t[2] = u32x4_t{0u, 0u, 0u, 0u};
  return f;
}
DEV f32x4 mma32h(u32x4_t a, u32x4_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// 2^k with m * 2^k in [2^13, 2^14), k in [-14, 100], and its reciprocal.  The scale only ever multiplies fp32 values
// (operands before the split, accumulators after the MFMAs), so nothing ties it to fp16's range: columns of a quiet signal
// (peak 1e-8) are normalised like any other and keep both fp16 terms normal.  Below 2^-87 (and for m = 0) k stays at 100.
DEV float pow2_scale(float m) {
  const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u);
  return __builtin_bit_cast(float, (unsigned)(127 + min(100, max(-14, 140 - e))) << 23);
}
DEV float recip_pow2(float s) { return __builtin_bit_cast(float, 0x7f000000u - __builtin_bit_cast(unsigned, s)); }
DEV float max8(f32x4 a, f32x4 b) {
  return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
}
// reductions over the sixteen lanes of a row on the VALU (DPP row_ror): a __shfl_xor is a ds_bpermute, an LDS round trip in a
// dependent chain -- eighteen of them per frame were ~2000 of the ~8800 cycles a wave spends on a frame
template <int N> DEV float dpp_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
DEV float row_sum16(float v) { v += dpp_ror<8>(v); v += dpp_ror<4>(v); v += dpp_ror<2>(v); v += dpp_ror<1>(v); return v; }
DEV float row_max16(float v) {
  v = fmaxf(v, dpp_ror<8>(v)); v = fmaxf(v, dpp_ror<4>(v)); v = fmaxf(v, dpp_ror<2>(v)); return fmaxf(v, dpp_ror<1>(v));
}
DEV float col_max(float m) {                  // over the four lanes of a column
  return group_max(m);
}

#ifndef MI355ASR_STFT_DIAG
#define MI355ASR_STFT_DIAG 0
#endif
// timing experiments (tools/build_variant.py ... -DMI355ASR_STFT_DIAG=n; WRONG results): bit 0 = no sample loads, 1 = no spectrum
// stores (the per-frame maximum stays), 2 = no MFMAs of stage 2, 3 = no log
constexpr int FDG = MI355ASR_STFT_DIAG;
#ifndef MI355ASR_STFT_TABS_LDS
#define MI355ASR_STFT_TABS_LDS 1
#endif
// round 5: window and twiddles (48 registers) live in LDS instead, read per frame: 190 -> <= 168 registers, three waves per SIMD
// instead of two -- a frame is a serial chain (samples -> split -> 24 MFMAs -> twiddle -> LDS transpose -> split -> 24 MFMAs ->
// log -> store) and the other waves of the SIMD are what fills its gaps (two-term kernel; MI355ASR_STFT_TABS_LDS=0 at build time:
// in registers as before)
template <int TM>
__global__ __launch_bounds__(BLOCK_THREADS, (TM == 2 && MI355ASR_STFT_TABS_LDS) ? 3 : 2) void fft_stft_split_kernel(FftStftArgs a) {
  constexpr bool TL = TM == 2 && MI355ASR_STFT_TABS_LDS;
  __shared__ float lds[WAVES_PER_BLOCK][2][32 * LDW];
  __shared__ __attribute__((aligned(16))) f32x4 tabs[TL ? 12 : 1][TL ? 64 : 1];     // [rt][kb] x (window, cos, sin), one row per lane
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int wave = threadIdx.x >> 6;
  float* Lre = lds[wave][0];
  float* Lim = lds[wave][1];
  const int wid = blockIdx.x * WAVES_PER_BLOCK + wave;
  const int nwaves = gridDim.x * WAVES_PER_BLOCK;
  const int total = a.B * a.F;

  // ---- frame-independent operands -> registers: stage matrices as [tile][term] fragments
  const u32x4_t* __restrict__ w1p = reinterpret_cast<const u32x4_t*>(TM == 2 ? a.w1h : a.w1s) + lane;   // [4 nt][TM terms]
  const u32x4_t* __restrict__ w2p = reinterpret_cast<const u32x4_t*>(TM == 2 ? a.w2h : a.w2s) + lane;   // [2 steps][2 nt][TM terms]
  u32x4_t w1[4][TM], w2[2][2][TM];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int t = 0; t < TM; ++t) w1[nt][t] = w1p[(nt * TM + t) * 64];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int t = 0; t < TM; ++t) w2[st][nt][t] = w2p[((st * 2 + nt) * TM + t) * 64];
  f32x4 hw[TL ? 1 : 2][TL ? 1 : 2], twc[TL ? 1 : 2][TL ? 1 : 2], tws[TL ? 1 : 2][TL ? 1 : 2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int n2 = 16 * rt + c;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 h_, c_, s_;
      h_.x = a.window[32 * (16 * kb + g4 + 0) + n2];
      h_.y = a.window[32 * (16 * kb + g4 + 1) + n2];
      h_.z = a.window[32 * (16 * kb + g4 + 2) + n2];
      h_.w = a.window[32 * (16 * kb + g4 + 3) + n2];
      c_.x = a.tw_c[(16 * kb + g4 + 0) * 32 + n2]; s_.x = a.tw_s[(16 * kb + g4 + 0) * 32 + n2];
      c_.y = a.tw_c[(16 * kb + g4 + 1) * 32 + n2]; s_.y = a.tw_s[(16 * kb + g4 + 1) * 32 + n2];
      c_.z = a.tw_c[(16 * kb + g4 + 2) * 32 + n2]; s_.z = a.tw_s[(16 * kb + g4 + 2) * 32 + n2];
      c_.w = a.tw_c[(16 * kb + g4 + 3) * 32 + n2]; s_.w = a.tw_s[(16 * kb + g4 + 3) * 32 + n2];
      if constexpr (TL) {
        if (wave == 0) { tabs[(2 * rt + kb) * 3 + 0][lane] = h_; tabs[(2 * rt + kb) * 3 + 1][lane] = c_; tabs[(2 * rt + kb) * 3 + 2][lane] = s_; }
      } else {
        hw[rt][kb] = h_; twc[rt][kb] = c_; tws[rt][kb] = s_;
      }
    }
  }
  if constexpr (TL) __syncthreads();
  auto tab = [&](int rt, int kb, int which) -> f32x4 {
    if constexpr (TL) return tabs[(2 * rt + kb) * 3 + which][lane];
    else return which == 0 ? hw[rt][kb] : (which == 1 ? twc[rt][kb] : tws[rt][kb]);
  };
  const int L = a.L;
  // acc += W^T x over the six term pairs, smallest first, for NT tiles and two row tiles: MFMAs on one accumulator 2 NT apart
  auto mma6 = [&](auto& acc, const auto& w, const Split8 (&x)[2], auto NT_T) {
    constexpr int NT = decltype(NT_T)::value;
#pragma unroll
    for (int ord = TM - 1; ord >= 0; --ord)
#pragma unroll
      for (int p = 0; p <= ord; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc[rt][nt] = TM == 2 ? mma32h(w[nt][ord - p], x[rt].t[p], acc[rt][nt]) : mma32(w[nt][ord - p], x[rt].t[p], acc[rt][nt]);
  };

#pragma unroll 1
  for (int fidx = wid; fidx < total; fidx += nwaves) {
    const int b = fidx / a.F, f = fidx - b * a.F;
    const float* __restrict__ wav = a.wav + (size_t)b * L;
    const int base = f * a.hop - a.pad_left;
    // ---- stage-1 operand: xw[32*n1 + n2], zero outside the signal (TF SAME / left-padded VALID framing)
    Split8 xs[2];
    float un1[2];                              // two-term: 1 / (stage scale x column scale) of the stage-1 accumulators
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      f32x4 xf[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int s0 = base + 32 * (16 * kb + g4) + 16 * rt + c;
        const bool o0 = (unsigned)(s0) < (unsigned)L, o1 = (unsigned)(s0 + 32) < (unsigned)L;
        const bool o2 = (unsigned)(s0 + 64) < (unsigned)L, o3 = (unsigned)(s0 + 96) < (unsigned)L;
        f32x4 v;
        if constexpr (FDG & 1) v = f32x4{(float)(s0 & 7), (float)(s0 & 3), 1.f, (float)c};
        else { v.x = wav[o0 ? s0 : 0]; v.y = wav[o1 ? s0 + 32 : 0]; v.z = wav[o2 ? s0 + 64 : 0]; v.w = wav[o3 ? s0 + 96 : 0]; }
        v.x = o0 ? v.x : 0.f; v.y = o1 ? v.y : 0.f; v.z = o2 ? v.z : 0.f; v.w = o3 ? v.w : 0.f;
        xf[kb] = v * tab(rt, kb, 0);
      }
      if constexpr (TM == 2) {
        const float sx = pow2_scale(col_max(max8(xf[0], xf[1])));
        un1[rt] = recip_pow2(sx * kStageScale);
        xs[rt] = split8h(xf[0] * splat4(sx), xf[1] * splat4(sx));
      } else {
        un1[rt] = 1.f;
        xs[rt] = split8(xf[0], xf[1]);
      }
    }
    // ---- stage 1: A = F32 * Xw   (acc1[rt][nt]: nt 0,1 = Re k1 0..31 ; nt 2,3 = Im k1 0..31)
    f32x4 acc1[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc1[rt][nt] = splat4(0.f);
    mma6(acc1, w1, xs, std::integral_constant<int, 4>{});
    if constexpr (TM == 2) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc1[rt][nt] = acc1[rt][nt] * splat4(un1[rt]);
    }
    // ---- bin 512: sum_n2 (-1)^n2 A_re[0][n2]   (A_re[0][n2] sits in lanes g == 0, tile 0, reg 0)
    float nyq = (g == 0) ? ((c & 1) ? -1.f : 1.f) * (acc1[0][0].x + acc1[1][0].x) : 0.f;
    nyq = row_sum16(nyq);
    // ---- twiddle + transpose through wave-private LDS: L[k1][n2]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int n2 = 16 * rt + c;
#pragma unroll
      for (int nr = 0; nr < 2; ++nr) {
        const f32x4 are = acc1[rt][nr], aim = acc1[rt][2 + nr];
        const f32x4 tc = tab(rt, nr, 1), ts = tab(rt, nr, 2);
        const f32x4 bre = are * tc + aim * ts;
        const f32x4 bim = aim * tc - are * ts;
        const int k1 = 16 * nr + g4;
        Lre[(k1 + 0) * LDW + n2] = bre.x; Lre[(k1 + 1) * LDW + n2] = bre.y;
        Lre[(k1 + 2) * LDW + n2] = bre.z; Lre[(k1 + 3) * LDW + n2] = bre.w;
        Lim[(k1 + 0) * LDW + n2] = bim.x; Lim[(k1 + 1) * LDW + n2] = bim.y;
        Lim[(k1 + 2) * LDW + n2] = bim.z; Lim[(k1 + 3) * LDW + n2] = bim.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- stage 2: X[k1][k2] = sum_n2 B[k1][n2] W32^(n2 k2); tokens = k1 = 16*rt + c; step 0: K = Re n2, step 1: K = Im n2
    f32x4 acc2[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      acc2[rt][0] = splat4(0.f);
      acc2[rt][1] = splat4(0.f);
    }
    if constexpr (TM == 2) {
      // one scale per row k1 for its real and its imaginary half: they meet in the same accumulator
      f32x4 yr[2][2], yi[2][2];
      float sy[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float* sr = Lre + (16 * rt + c) * LDW + g4;
        const float* si = Lim + (16 * rt + c) * LDW + g4;
        yr[rt][0] = *reinterpret_cast<const f32x4*>(sr); yr[rt][1] = *reinterpret_cast<const f32x4*>(sr + 16);
        yi[rt][0] = *reinterpret_cast<const f32x4*>(si); yi[rt][1] = *reinterpret_cast<const f32x4*>(si + 16);
        sy[rt] = pow2_scale(col_max(fmaxf(max8(yr[rt][0], yr[rt][1]), max8(yi[rt][0], yi[rt][1]))));
      }
      Split8 ys[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) ys[rt] = split8h(yr[rt][0] * splat4(sy[rt]), yr[rt][1] * splat4(sy[rt]));
      mma6(acc2, w2[0], ys, std::integral_constant<int, 2>{});
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) ys[rt] = split8h(yi[rt][0] * splat4(sy[rt]), yi[rt][1] * splat4(sy[rt]));
      mma6(acc2, w2[1], ys, std::integral_constant<int, 2>{});
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const f32x4 un = splat4(recip_pow2(sy[rt] * kStageScale));
        acc2[rt][0] = acc2[rt][0] * un;
        acc2[rt][1] = acc2[rt][1] * un;
      }
    } else {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        Split8 ys[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const float* src = (st == 0 ? Lre : Lim) + (16 * rt + c) * LDW + g4;
          ys[rt] = split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 16));
        }
        mma6(acc2, w2[st], ys, std::integral_constant<int, 2>{});
      }
    }
    // ---- power, log, store, per-frame max.  lane holds bins k1 + 32*k2, k1 = 16*rt + c, k2 = 4*g + j
    float mx = -INFINITY;
    float* orow = a.logp + ((size_t)b * a.F + f) * a.LP;
    const float kscale = a.db10 ? (10.0f * 0.69314718f / 2.30258509f) : (0.69314718f / 2.30258509f);   // log2 -> dB / log10
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const f32x4 re = acc2[rt][0], im = acc2[rt][1];
      const f32x4 p = re * re + im * im;
      const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float l = (FDG & 8) ? pv[j] : __log2f(fmaxf(pv[j], 1e-10f)) * kscale;
        if constexpr (!(FDG & 2)) orow[16 * rt + c + 32 * (g4 + j)] = l;
        mx = fmaxf(mx, l);
      }
    }
    if (lane == 0) {
      const float l = __log2f(fmaxf(nyq * nyq, 1e-10f)) * kscale;
      orow[512] = l;
      mx = fmaxf(mx, l);
    }
    mx = col_max(row_max16(mx));
    if (lane == 0) a.pmax[fidx] = mx;
  }
}

}  // namespace

int launch_fft_stft(const FftStftArgs& a, hipStream_t s) {
  const int total = a.B * a.F;
  // enough waves for 4 per SIMD (their VALU / LDS phases hide under each other's MFMAs), each looping over frames
  const int waves = std::min(total, 4096);
  // MI355ASR_FFT_SPLIT=0: both DFT stages on the fp32 MFMA (round 1) instead of the split-bf16 pipe
  static const bool split = mi355_env("MI355ASR_FFT_SPLIT", 1) != 0;
  // MI355ASR_FFT_TERMS=3: three bf16 terms instead of two fp16 terms (the stage matrices and the per-column scales make the
  // two-term kernel independent of the signal's amplitude: no bound is assumed)
  static const bool three = mi355_env("MI355ASR_FFT_TERMS", -1) == 3;
  if (split && a.w1h && a.w2h && !three) {
    note_scheme(SCHEME_F16X2);
    // three workgroups per CU are resident (see MI355ASR_STFT_TABS_LDS): one round of 768 workgroups, every wave loops over its frames
    const int waves2 = MI355ASR_STFT_TABS_LDS ? std::min(total, 3072) : waves;
    hipLaunchKernelGGL(fft_stft_split_kernel<2>, dim3((waves2 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK_THREADS), 0, s, a);
    return 0;
  }
  if (split && a.w1s && a.w2s) {
    note_scheme(SCHEME_BF16X3);
    hipLaunchKernelGGL(fft_stft_split_kernel<3>, dim3((waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK_THREADS), 0, s, a);
    return 0;
  }
  note_scheme(SCHEME_F32);
  hipLaunchKernelGGL(fft_stft_kernel, dim3((waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
