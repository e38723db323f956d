// One dense layer on the bf16 matrix pipe with exactly split operands, for dmodel 256 / 512 and long batches
// (ConformerM / ConformerL, conformerM.yml / conformerL.yml):
//     Y[M, N] = epilogue( LN?(X)[M, K] . W[K, N] + b ),   K a multiple of 64, N a multiple of 128,
// the Gemm16Args contract of bf16.hip (same epilogues, same call sites in run_block) with fp32 results: every fp32
// operand is the exact sum of three bf16 terms and the six term pairs with i + j <= 2 go through
// v_mfma_f32_16x16x32_bf16, smallest first (subconv.hip / leaf.hip: as accurate as an fp32 FMA chain, 6 x 16 cycles
// per 32 k-slots instead of 8 x 32 for v_mfma_f32_16x16x4_f32).
//
// Shape of the work (the conv-subsampling slab kernel of subconv.hip with a token row as the operand):
//   * workgroup = 8 waves x RT row tiles of 16 tokens (256 or 128 tokens) x one chunk of 8 column tiles (128 columns;
//     for GLU: 4 value + 4 gate tiles); grid (ceil(M / tokens), N / 128);
//   * per 32-wide k-step a 24 KB weight slab [8 tiles][3 terms][64 lanes][8 bf16] goes global -> LDS directly
//     (global_load_lds_dwordx4) into the other half of a double buffer while the MFMAs of the step run; the eight waves
//     share it (48 or 96 MFMAs per wave and slab);
//   * the X operand: lane (token c, group g) loads x[token][32 s + 4 g .. + 3] and [32 s + 16 + 4 g .. + 3] two steps
//     ahead, applies the prologue LayerNorm (statistics from a two-pass prologue, gamma / beta in LDS) and splits into
//     three bf16x8 terms -- by the lower waves before the MFMAs of the step, by the upper waves (their SIMD partners)
//     for the next step after them, so one wave's VALU phase faces the other's MFMAs;
//   * the slab wait is a counted vmcnt: the X loads issued after the slab DMA may stay in flight across the barrier.
// X is re-read and re-split per column chunk (N / 128 times); at 48-96 MFMAs per ~60 VALU instructions of split that
// is hidden, and the re-reads come from L2.
#include <cstdlib>

#include "common.h"
#include "launch.h"
#include "wstream.h"

namespace {

constexpr int GW = 8, GT = GW * 64;               // waves / threads per workgroup
constexpr int GNB = 8;                            // column tiles per workgroup
constexpr int GSLAB = GNB * 3 * 64;               // 16-byte fragments per k-step (24 KB)
constexpr int GLN_MAX = 512;                      // widest prologue LayerNorm (dmodel)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEV void dma16(const u32x4* gsrc, u32x4* lds) {   // 16 bytes per lane, global -> LDS; lane i lands at lds + 16 i
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

struct Frag { u32x4 t[3]; };                      // 8 k-slots x 3 terms

// exact three-term bf16 split of eight fp32 values by truncation (remainders exact), two values per dword
DEV Frag split8(f32x4 lo, f32x4 hi) {
  float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Frag f;
#pragma unroll
  for (int term = 0; term < 3; ++term) {
    unsigned d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned a0 = __builtin_bit_cast(unsigned, v[2 * k]), a1 = __builtin_bit_cast(unsigned, v[2 * k + 1]);
      d[k] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);           // (a0 >> 16) | (a1 & 0xffff0000)
      if (term < 2) {
        v[2 * k] -= __builtin_bit_cast(float, a0 & 0xffff0000u);
        v[2 * k + 1] -= __builtin_bit_cast(float, a1 & 0xffff0000u);
      }
    }
    f.t[term] = u32x4{d[0], d[1], d[2], d[3]};
  }
  return f;
}

struct XRegs { f32x4 lo, hi; };                   // one lane's eight operand values of one k-step, before LN / split

template <int EPI, bool LN, int RT>
__global__ __launch_bounds__(GT, 2) void gemm_ring_kernel(Gemm16Args a, const u32x4* __restrict__ wring) {
  __shared__ __attribute__((aligned(16))) u32x4 wl[2][GSLAB];
  __shared__ __attribute__((aligned(16))) float p_g[LN ? GLN_MAX : 4], p_b[LN ? GLN_MAX : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int r0 = blockIdx.x * (GW * 16 * RT);
  const int chunk = blockIdx.y;
  const int steps = a.K / 32;
  const u32x4* __restrict__ wg = wring + (size_t)chunk * steps * GSLAB;
  constexpr int NQ = GSLAB / GT;                  // 3 DMA instructions per wave and slab
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // slab 0
#pragma unroll
  for (int q = 0; q < NQ; ++q) dma16(wg + GT * q + 64 * wv + lane, &wl[0][GT * q + 64 * wv]);
  if (LN) {
    for (int i = threadIdx.x; i < a.K; i += GT) { p_g[i] = a.ln_g[i]; p_b[i] = a.ln_b[i]; }
  }
  int tok[RT];
  bool live[RT];
  const float* xr[RT];
  float mean[RT], rstd[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    tok[rt] = r0 + (16 * RT) * wave + 16 * rt + c;
    live[rt] = tok[rt] < a.M;
    const int rowi = min(tok[rt], a.M - 1);
    xr[rt] = (a.rpb > 0 ? a.x + (size_t)(rowi / a.rpb) * a.bstride + (size_t)(rowi % a.rpb) * a.ldx
                        : a.x + (size_t)rowi * a.ldx) + g4;
    mean[rt] = 0.f;
    rstd[rt] = 1.f;
    if (LN) {                                     // two-pass statistics, biased variance, eps inside the sqrt (Keras)
      float s = 0.f;
      for (int st = 0; st < steps; ++st) {
        const f32x4 u = ldg4(xr[rt] + 32 * st), v = ldg4(xr[rt] + 32 * st + 16);
        s += ((u.x + u.y) + (u.z + u.w)) + ((v.x + v.y) + (v.z + v.w));
      }
      mean[rt] = group_sum(s) / (float)a.K;
      float q = 0.f;
      for (int st = 0; st < steps; ++st) {
        const f32x4 u = ldg4(xr[rt] + 32 * st) - splat4(mean[rt]), v = ldg4(xr[rt] + 32 * st + 16) - splat4(mean[rt]);
        q += ((u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w)) + ((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
      }
      rstd[rt] = 1.0f / sqrtf(group_sum(q) / (float)a.K + a.eps);
    }
  }
  auto xload = [&](int st, XRegs (&x)[RT]) {
    const int sc = min(st, steps - 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { x[rt].lo = ldg4(xr[rt] + 32 * sc); x[rt].hi = ldg4(xr[rt] + 32 * sc + 16); }
  };
  auto xsplit = [&](int st, const XRegs (&x)[RT], Frag (&f)[RT]) {
    const int sc = min(st, steps - 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 lo = x[rt].lo, hi = x[rt].hi;
      if (LN) {
        const f32x4 m = splat4(mean[rt]), r = splat4(rstd[rt]);
        lo = (lo - m) * r * *reinterpret_cast<const f32x4*>(p_g + 32 * sc + g4) + *reinterpret_cast<const f32x4*>(p_b + 32 * sc + g4);
        hi = (hi - m) * r * *reinterpret_cast<const f32x4*>(p_g + 32 * sc + 16 + g4) + *reinterpret_cast<const f32x4*>(p_b + 32 * sc + 16 + g4);
      }
      f[rt] = split8(lo, hi);
    }
  };

  f32x4 acc[RT][GNB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int n = 0; n < GNB; ++n) acc[rt][n] = splat4(0.f);

  XRegs xq[2][RT];
  Frag xa[RT];
  xload(0, xq[0]);
  xload(1, xq[1]);
  __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0): slab 0 (and the first operands)
  __syncthreads();

  auto mfma_step = [&](int cur) {
    bf16x8 wf[3];
#pragma unroll
    for (int n = 0; n < GNB; ++n) {
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[t] = __builtin_bit_cast(bf16x8, wl[cur][(n * 3 + t) * 64 + lane]);
#pragma unroll
      for (int ord = 2; ord >= 0; --ord)
#pragma unroll
        for (int p = 0; p <= ord; ++p)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ord - p], __builtin_bit_cast(bf16x8, xa[rt].t[p]),
                                                                 acc[rt][n], 0, 0, 0);
    }
  };
  auto slab_dma = [&](int st, int buf) {          // slab `st` into wl[buf] (read last in step st - 2, two barriers ago)
    const u32x4* src = wg + (size_t)min(st, steps - 1) * GSLAB;
#pragma unroll
    for (int q = 0; q < NQ; ++q) dma16(src + GT * q + 64 * wv + lane, &wl[buf][GT * q + 64 * wv]);
  };
  // one k-step; PAR = its parity (slab buffer, operand register stage), LATE = this wave splits after its MFMAs
  auto step = [&](int s, auto PAR_T, auto LATE_T) {
    constexpr int PAR = decltype(PAR_T)::value;
    constexpr bool LATE = decltype(LATE_T)::value;
    slab_dma(s + 1, PAR ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LATE) {
      xsplit(s, xq[PAR], xa);
      xload(s + 2, xq[PAR]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(PAR);
    } else {
      mfma_step(PAR);
      __builtin_amdgcn_sched_barrier(0);
      xsplit(s + 1, xq[PAR ^ 1], xa);
      xload(s + 3, xq[PAR ^ 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the 2 RT operand loads issued after the slab DMA may still be in flight: vmcnt(2 RT)
    if constexpr (RT == 2) __builtin_amdgcn_s_waitcnt(0x0f74); else __builtin_amdgcn_s_waitcnt(0x0f72);
    __syncthreads();
  };
  auto run = [&](auto LATE_T) {
    constexpr bool LATE = decltype(LATE_T)::value;
    if constexpr (LATE) {
      xsplit(0, xq[0], xa);
      xload(2, xq[0]);
    }
#pragma unroll 1
    for (int s = 0; s < steps; s += 2) {
      step(s, std::integral_constant<int, 0>{}, LATE_T);
      step(s + 1, std::integral_constant<int, 1>{}, LATE_T);
    }
  };
  if (wv >= GW / 2) run(std::integral_constant<bool, true>{});
  else run(std::integral_constant<bool, false>{});

  // ---- epilogue: lane holds Y[token c of row tile rt][feature 16 * tile + g4 + 0..3]
  const int half = a.NT / 2;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (!live[rt]) continue;
    float* yrow = a.y + (size_t)tok[rt] * a.ldy;
    if constexpr (EPI == E16_GLU) {
#pragma unroll
      for (int i = 0; i < GNB / 2; ++i) {
        const int f0 = 16 * (chunk * (GNB / 2) + i) + g4;
        const f32x4 va = acc[rt][i] + ldg4(a.bias + f0), vb = acc[rt][GNB / 2 + i] + ldg4(a.bias + 16 * half + f0);
        const f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
        stg4(yrow + f0, o);
      }
    } else {
#pragma unroll
      for (int i = 0; i < GNB; ++i) {
        const int tile = chunk * GNB + i, f0 = 16 * tile + g4;
        f32x4 v = acc[rt][i] + ldg4(a.bias + f0);
        if constexpr (EPI == E16_SWISH) v = swish4(v);
        if constexpr (EPI == E16_AFFSWISH) v = swish4(v * ldg4(a.aff_s + f0) + ldg4(a.aff_t + f0));
        if constexpr (EPI == E16_QKV) { if (tile < a.qtiles) v *= splat4(a.qscale); }
        if constexpr (EPI == E16_RES) v = ldg4(a.res + (size_t)tok[rt] * a.ldy + f0) + splat4(a.scale) * v;
        stg4(yrow + f0, v);
      }
    }
  }
}

// y[row] = LayerNorm(y[row]) in place, one wave per row (two-pass statistics, Keras semantics): the block's final
// LayerNorm after a residual epilogue whose row spans several column chunks
__global__ __launch_bounds__(256) void ring_layernorm_rows_kernel(float* y, const float* __restrict__ g, const float* __restrict__ b,
                                                                  int M, int N, int ld, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* p = y + (size_t)row * ld;
  float s = 0.f;
  for (int i = lane * 4; i < N; i += 256) { const f32x4 v = ldg4(p + i); s += (v.x + v.y) + (v.z + v.w); }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)N;
  float q = 0.f;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 d = ldg4(p + i) - splat4(mean);
    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) q += __shfl_xor(q, off);
  const float rstd = 1.0f / sqrtf(q / (float)N + eps);
  for (int i = lane * 4; i < N; i += 256)
    stg4(p + i, (ldg4(p + i) - splat4(mean)) * splat4(rstd) * ldg4(g + i) + ldg4(b + i));
}

template <int EPI, bool LN>
int go(const Gemm16Args& a, const void* ring, hipStream_t s) {
  const int chunks = (EPI == E16_GLU ? a.NT / 2 : a.NT) / (EPI == E16_GLU ? GNB / 2 : GNB);
  // two row tiles per wave when that still gives every CU a workgroup (MI355ASR_RING_RT=1 / 2 forces one shape: tests)
  static const int force_rt = [] { const char* v = getenv("MI355ASR_RING_RT"); return v ? atoi(v) : 0; }();
  if (force_rt == 2 || (force_rt != 1 && (size_t)((a.M + 255) / 256) * chunks >= 256))
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 2>), dim3((a.M + 255) / 256, chunks), dim3(GT), 0, s, a, (const u32x4*)ring);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 1>), dim3((a.M + 127) / 128, chunks), dim3(GT), 0, s, a, (const u32x4*)ring);
  return 0;
}

}  // namespace

// Shapes the ring kernel takes; the pack (api.hip: pack_ring) exists for the dense layers of dmodel 256 / 512 blocks.
bool gemm_ring_applicable(int epi, bool ln, const Gemm16Args& a) {
  const int ntc = epi == E16_GLU ? a.NT / 2 : a.NT;
  if (a.K % 64 != 0 || a.K < 64 || ntc % (epi == E16_GLU ? 4 : 8) != 0 || a.n_valid != (epi == E16_GLU ? 16 * ntc : 16 * a.NT)) return false;
  if (ln && a.K > GLN_MAX) return false;
  if ((a.ldx & 3) != 0 || (a.ldy & 3) != 0 || !a.y) return false;
  switch (epi) {
    case E16_BIAS: return !ln;
    case E16_SWISH: case E16_QKV: case E16_GLU: return ln;
    case E16_RES: case E16_AFFSWISH: return !ln;
    default: return false;
  }
}

int launch_gemm_ring(int epi, bool ln, const Gemm16Args& a, const void* ring, hipStream_t s) {
  if (!ring || !gemm_ring_applicable(epi, ln, a)) return -1;
  switch (epi) {
    case E16_BIAS: go<E16_BIAS, false>(a, ring, s); break;
    case E16_SWISH: go<E16_SWISH, true>(a, ring, s); break;
    case E16_QKV: go<E16_QKV, true>(a, ring, s); break;
    case E16_GLU: go<E16_GLU, true>(a, ring, s); break;
    case E16_AFFSWISH: go<E16_AFFSWISH, false>(a, ring, s); break;
    case E16_RES:
      go<E16_RES, false>(a, ring, s);
      // the optional LayerNorm over the output row needs all of it: a second pass over y (as bf16.hip does for wide rows)
      if (a.fln_g)
        hipLaunchKernelGGL(ring_layernorm_rows_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a.y, a.fln_g, a.fln_b, a.M, 16 * a.NT, a.ldy, a.eps);
      break;
    default: return -1;
  }
  return 0;
}
