// One dense layer on the bf16 matrix pipe with exactly split operands, for dmodel 256 / 512 and long batches
// (ConformerM / ConformerL, conformerM.yml / conformerL.yml):
//     Y[M, N] = epilogue( LN?(X)[M, K] . W[K, N] + b ),   K a multiple of 128, N a multiple of 128,
// the Gemm16Args contract of bf16.hip (same epilogues, same call sites in run_block) with fp32 results: every fp32
// operand is the exact sum of three bf16 terms and the six term pairs with i + j <= 2 go through
// v_mfma_f32_16x16x32_bf16, smallest first (subconv.hip / leaf.hip: as accurate as an fp32 FMA chain, 6 x 16 cycles
// per 32 k-slots instead of 8 x 32 for v_mfma_f32_16x16x4_f32).
//
// Shape of the work (the conv-subsampling slab kernel of subconv.hip with a token row as the operand):
//   * workgroup = 8 waves x RT row tiles of 16 tokens (256 or 128 tokens) x one chunk of 8 column tiles (128 columns;
//     for GLU: 4 value + 4 gate tiles); grid (ceil(M / tokens), N / 128);
//   * per 32-wide k-step a 24 KB weight slab [8 tiles][3 terms][64 lanes][8 bf16] goes global -> LDS directly
//     (global_load_lds_dwordx4) into a four-slot ring, three steps ahead of its use; the eight waves share it (48 or 96
//     MFMAs per wave and slab);
//   * the X operand: lane (token c, group g) loads x[token][32 s + 4 g .. + 3] and [32 s + 16 + 4 g .. + 3] four steps
//     ahead (two with two row tiles per wave: registers; a first version with one slab and two operand steps in flight ran at the memory latency: 2 us per step),
//     applies the prologue LayerNorm (two-pass statistics, gamma / beta in LDS) and splits into three bf16x8 terms --
//     by the lower waves before the MFMAs of the step, by the upper waves (their SIMD partners) for the next step after
//     them, so one wave's VALU phase faces the other's MFMAs;
//   * the wait before a step's barrier is a counted vmcnt: everything issued after the slab of the NEXT step -- two
//     more slabs and the operand loads of three iterations -- stays in flight across the barrier.
// X is re-read and re-split per column chunk (N / 128 times); at 48-96 MFMAs per ~60 VALU instructions of split that
// is hidden, and the re-reads come from L2.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "env.h"
#include "launch.h"
#include "wstream.h"

namespace {

constexpr int GW = 8, GT = GW * 64;               // waves / threads per workgroup
constexpr int GNB = 8;                            // column tiles per workgroup
// 16-byte fragments per k-step: GNB x TERMS x 64 (24 KB with three terms, 8 KB in bf16 mode)
constexpr int GLN_MAX = 512;                      // widest prologue LayerNorm (dmodel)
constexpr int GUNR = 4;                           // unroll of the step loop (slab slots and operand stages divide it)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEV void dma16(const u32x4* gsrc, u32x4* lds) {   // 16 bytes per lane, global -> LDS; lane i lands at lds + 16 i
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int OFF>
DEV u32x4 lds_read16(unsigned addr) {             // not visible to the compiler as an LDS access: see mfma_step
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// TERMS = 3: fp32 operands, exactly split.  TERMS = 1: bf16 mode (BASELINE config 3) -- weights rounded once on the host,
// activations rounded to nearest even at the operand, one MFMA per tile and step; same arithmetic as bf16.hip.
template <int TERMS>
struct Frag { u32x4 t[TERMS]; };                  // 8 k-slots x TERMS terms

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// exact three-term bf16 split of eight fp32 values by truncation (remainders exact), two values per dword; or the eight
// values rounded to nearest-even bf16
template <int TERMS>
DEV Frag<TERMS> split8(f32x4 lo, f32x4 hi) {
  float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Frag<TERMS> f;
  if constexpr (TERMS == 1) {
    unsigned d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 pr = {v[2 * k], v[2 * k + 1]};
      d[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2_t));     // v_cvt_pk_bf16_f32
    }
    f.t[0] = u32x4{d[0], d[1], d[2], d[3]};
  } else {
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
  }
  return f;
}

struct XRegs { f32x4 lo, hi; };                   // one lane's eight operand values of one k-step, before LN / split

template <int EPI, bool LN, int RT, int RING, int TERMS, bool EDMA>
__global__ __launch_bounds__(GT, 2) void gemm_ring_kernel(Gemm16Args a, const u32x4* __restrict__ wring, int cpw) {
  constexpr int GSLAB = GNB * TERMS * 64;
  __shared__ __attribute__((aligned(16))) u32x4 wl[RING][GSLAB];
  __shared__ __attribute__((aligned(16))) float p_gb[2 * (LN ? GLN_MAX : 4)];   // LayerNorm gamma, then beta
  float* const p_g = p_gb;
  float* const p_b = p_gb + (LN ? GLN_MAX : 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int r0 = blockIdx.x * (GW * 16 * RT);
  // this workgroup's column chunks chunk0 .. chunk0 + cpw - 1, one after the other on the same rows: their slabs are
  // one contiguous stream (the ring keeps running across the chunk boundary) and the prologue is paid once
  const int chunk0 = blockIdx.y * cpw;
  // (the class head split over ranges of chunks: the last range may hold fewer)
  const int ncc = (EPI == E16_HEAD && a.head_chunks > 0) ? min(cpw, a.head_chunks - chunk0) : cpw;
  const int steps = a.K / 32, total = ncc * steps;
  const u32x4* __restrict__ wg = wring + (size_t)chunk0 * steps * GSLAB;
  constexpr int NQ = (GSLAB + GT - 1) / GT;       // DMA instructions per wave and slab: 3, or 1 in bf16 mode
  // EDMA: only the lower four waves (the ones that split before their MFMAs) issue slab DMAs, twice as many each: an
  // LDS-DMA instruction holds the issuing wave for ~100 cycles, and the upper waves' MFMAs are what the SIMD should be
  // doing at the top of a step
  constexpr int NQE = GSLAB / 64 / (GW / 2);
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // the first RING - 1 slabs
#pragma unroll
  for (int st = 0; st < RING - 1; ++st) {
    const u32x4* src = wg + (size_t)min(st, total - 1) * GSLAB;
    if constexpr (EDMA) {
      if (wv < GW / 2) {
#pragma unroll
        for (int q = 0; q < NQE; ++q) dma16(src + 64 * ((GW / 2) * q + wv) + lane, &wl[st][64 * ((GW / 2) * q + wv)]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) dma16(src + GT * q + 64 * wv + lane, &wl[st][GT * q + 64 * wv]);
    }
  }
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
  }
  if (LN) {
    // two-pass statistics (biased variance, eps inside the sqrt: Keras).  dmodel 256: the whole row of every row tile
    // is requested at once and both passes run from registers (one memory round trip before the first step instead
    // of one per pass and row tile); wider rows: eight steps in flight at a time, the second pass reloads (L2).
    if (steps == 8) {
      f32x4 u[RT][8], v[RT][8];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 8; ++i) { u[rt][i] = ldg4(xr[rt] + 32 * i); v[rt][i] = ldg4(xr[rt] + 32 * i + 16); }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sm += ((u[rt][i].x + u[rt][i].y) + (u[rt][i].z + u[rt][i].w)) + ((v[rt][i].x + v[rt][i].y) + (v[rt][i].z + v[rt][i].w));
        mean[rt] = group_sum(sm) / (float)a.K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 du = u[rt][i] - splat4(mean[rt]), dv = v[rt][i] - splat4(mean[rt]);
          q += ((du.x * du.x + du.y * du.y) + (du.z * du.z + du.w * du.w)) + ((dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w));
        }
        rstd[rt] = 1.0f / sqrtf(group_sum(q) / (float)a.K + a.eps);
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float sm = 0.f;
        for (int st0 = 0; st0 < steps; st0 += 8) {
          f32x4 u[8], v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { u[i] = ldg4(xr[rt] + 32 * (st0 + i)); v[i] = ldg4(xr[rt] + 32 * (st0 + i) + 16); }
#pragma unroll
          for (int i = 0; i < 8; ++i) sm += ((u[i].x + u[i].y) + (u[i].z + u[i].w)) + ((v[i].x + v[i].y) + (v[i].z + v[i].w));
        }
        mean[rt] = group_sum(sm) / (float)a.K;
        float q = 0.f;
        for (int st0 = 0; st0 < steps; st0 += 8) {
          f32x4 u[8], v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { u[i] = ldg4(xr[rt] + 32 * (st0 + i)); v[i] = ldg4(xr[rt] + 32 * (st0 + i) + 16); }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x4 du = u[i] - splat4(mean[rt]), dv = v[i] - splat4(mean[rt]);
            q += ((du.x * du.x + du.y * du.y) + (du.z * du.z + du.w * du.w)) + ((dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w));
          }
        }
        rstd[rt] = 1.0f / sqrtf(group_sum(q) / (float)a.K + a.eps);
      }
    }
  }
  // The operand loads of the step loop are inline asm as well, with a counted vmcnt before their use: hipcc's own wait
  // there was vmcnt(0) (everything in flight, the slab just requested included).
  auto xload = [&](int st, XRegs (&x)[RT]) {   // st < 2 steps: the operand steps wrap into the next column chunk
    const int sc = st >= steps ? st - steps : st;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float* p = xr[rt] + 32 * sc;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[rt].lo) : "v"(p));
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(x[rt].hi) : "v"(p));
    }
  };
  // waits until the loads of stage `x` have landed: N = VMEM operations issued after them
  auto xwait = [&](XRegs (&x)[RT], auto N_T) {
    constexpr int N = decltype(N_T)::value;
    if constexpr (RT == 2)
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0].lo), "+v"(x[0].hi), "+v"(x[1].lo), "+v"(x[1].hi) : "n"(N));
    else
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x[0].lo), "+v"(x[0].hi) : "n"(N));
  };
  // gamma / beta of the prologue LayerNorm come from LDS through the same inline-asm reads as the weight fragments (a
  // compiler-visible LDS read would again cost a vmcnt(0) while slabs are in flight)
  const unsigned ln_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(p_g + g4);
  constexpr unsigned LN_B = sizeof(float) * (LN ? GLN_MAX : 4);      // p_b follows p_g
  auto xsplit = [&](int st, const XRegs (&x)[RT], Frag<TERMS> (&f)[RT]) {
    const int sc = st >= steps ? st - steps : st;
    u32x4 gl = {}, gh = {}, bl = {}, bh = {};
    if (LN) {
      const unsigned at = ln_base + 128u * (unsigned)sc;
      gl = lds_read16<0>(at); gh = lds_read16<64>(at); bl = lds_read16<LN_B>(at); bh = lds_read16<LN_B + 64>(at);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gl), "+v"(gh), "+v"(bl), "+v"(bh));
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 lo = x[rt].lo, hi = x[rt].hi;
      if (LN) {
        const f32x4 m = splat4(mean[rt]), r = splat4(rstd[rt]);
        lo = (lo - m) * r * __builtin_bit_cast(f32x4, gl) + __builtin_bit_cast(f32x4, bl);
        hi = (hi - m) * r * __builtin_bit_cast(f32x4, gh) + __builtin_bit_cast(f32x4, bh);
      }
      f[rt] = split8<TERMS>(lo, hi);
    }
  };

  f32x4 acc[RT][GNB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int n = 0; n < GNB; ++n) acc[rt][n] = splat4(0.f);

  // operand register stages: four steps ahead with one row tile, two with two (256 registers)
  constexpr int XST = (RT == 1 && RING == 4) ? GUNR : 2;
  XRegs xq[XST][RT];
  Frag<TERMS> xa[RT];
#pragma unroll
  for (int st = 0; st < XST; ++st) xload(st, xq[st]);
  __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0): the first slabs and operands
  __syncthreads();

  // The weight fragments are read with inline-asm ds_read_b128 and counted lgkmcnt waits: a compiler-visible LDS read of
  // the ring makes hipcc wait for vmcnt(0) first -- it cannot tell the slot being read from the slots the LDS-DMA in
  // flight is writing -- which put the whole DMA latency into every step (the first version of this kernel: 23 % MFMA
  // busy).  Two column tiles at a time (six MFMAs in a row on one accumulator would each wait for the one before), the
  // next pair requested before the MFMAs of the current one; LDS returns in order, so lgkmcnt(6) = "all but the six
  // newest reads".
  auto mfma_step = [&](int slot) {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&wl[slot][lane]);
    u32x4 wa[2][TERMS], wb[2][TERMS];
    auto fetch = [&](u32x4 (&w)[2][TERMS], auto PAIR_T) {
      constexpr int PAIR = decltype(PAIR_T)::value;
      static_for<0, 2>([&](auto H_T) {
        constexpr int h = decltype(H_T)::value;
        static_for<0, TERMS>([&](auto T_T) {
          constexpr int t = decltype(T_T)::value;
          w[h][t] = lds_read16<((2 * PAIR + h) * TERMS + t) * 1024>(base);
        });
      });
    };
    auto wait = [&](u32x4 (&w)[2][TERMS], auto N_T) {
      constexpr int N = decltype(N_T)::value;
      if constexpr (TERMS == 3)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[1][2]) : "n"(N));
      else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0][0]), "+v"(w[1][0]) : "n"(N));
    };
    auto mma = [&](const u32x4 (&w)[2][TERMS], auto PAIR_T) {
      constexpr int PAIR = decltype(PAIR_T)::value;
#pragma unroll
      for (int ord = TERMS - 1; ord >= 0; --ord)
#pragma unroll
        for (int p = 0; p <= ord; ++p)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[rt][2 * PAIR + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[h][ord - p]),
                  __builtin_bit_cast(bf16x8, xa[rt].t[p]), acc[rt][2 * PAIR + h], 0, 0, 0);
    };
    constexpr int NEW = 2 * TERMS;                // reads of one pair: "all but the newest pair" = lgkmcnt(NEW)
    fetch(wa, std::integral_constant<int, 0>{});
    fetch(wb, std::integral_constant<int, 1>{});
    wait(wa, std::integral_constant<int, NEW>{});
    mma(wa, std::integral_constant<int, 0>{});
    fetch(wa, std::integral_constant<int, 2>{});
    wait(wb, std::integral_constant<int, NEW>{});
    mma(wb, std::integral_constant<int, 1>{});
    fetch(wb, std::integral_constant<int, 3>{});
    wait(wa, std::integral_constant<int, NEW>{});
    mma(wa, std::integral_constant<int, 2>{});
    wait(wb, std::integral_constant<int, 0>{});
    mma(wb, std::integral_constant<int, 3>{});
  };
  auto slab_dma = [&](int st, int slot, auto LATE_T) {     // slab `st` into a slot last read in step st - RING
    const u32x4* src = wg + (size_t)min(st, total - 1) * GSLAB;
    if constexpr (EDMA) {
      if constexpr (!decltype(LATE_T)::value) {
#pragma unroll
        for (int q = 0; q < NQE; ++q) dma16(src + 64 * ((GW / 2) * q + wv) + lane, &wl[slot][64 * ((GW / 2) * q + wv)]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) dma16(src + GT * q + 64 * wv + lane, &wl[slot][GT * q + 64 * wv]);
    }
  };
  // one k-step; SL = s mod GUNR (slab slot SL % RING, operand stage SL % XST), LATE = this wave splits after its MFMAs
  auto step = [&](int s, int gs, auto SL_T, auto LATE_T) {     // s: step within the chunk, gs: within the slab stream
    constexpr int SL = decltype(SL_T)::value, NX = (SL + 1) % GUNR, PV = (SL + GUNR - 1) % GUNR;
    constexpr bool LATE = decltype(LATE_T)::value;
    slab_dma(gs + RING - 1, PV % RING, LATE_T);   // the slot of step gs - 1: every wave is past that step's barrier
    constexpr int NQR = EDMA ? (LATE ? 0 : NQE) : NQ;     // slab pieces this wave issues per step
    __builtin_amdgcn_sched_barrier(0);
    // operand loads of a stage are followed by XST - 1 whole iterations (3 slab pieces + 2 RT loads) and this one's slab
    constexpr int XAFTER = (XST - 1) * (NQR + 2 * RT) + NQR;
    if constexpr (!LATE) {
      xwait(xq[SL % XST], std::integral_constant<int, XAFTER>{});
      xsplit(s, xq[SL % XST], xa);
      xload(s + XST, xq[SL % XST]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(SL % RING);
    } else {
      mfma_step(SL % RING);
      __builtin_amdgcn_sched_barrier(0);
      xwait(xq[NX % XST], std::integral_constant<int, XAFTER>{});
      xsplit(s + 1, xq[NX % XST], xa);
      xload(s + 1 + XST, xq[NX % XST]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // slab s + 1 was issued RING - 2 iterations ago; what was issued after it -- the operand loads of that iteration and
    // everything of the iterations since (3 slab pieces + 2 RT operand loads each) -- may stay in flight
    constexpr int INFLIGHT = 2 * RT + (RING - 2) * (NQR + 2 * RT);
    __builtin_amdgcn_s_waitcnt(0x0f70 | (INFLIGHT & 15) | ((INFLIGHT >> 4) << 14));
    // a bare s_barrier: __syncthreads() carries a workgroup fence, for which hipcc waits for vmcnt(0) -- every slab and
    // operand load in flight.  What must be visible after this barrier is the slab of step s + 1 (waited for above by
    // each wave for its own pieces); the fragment reads of this step's slot are complete (lgkmcnt(0) before the last MFMAs)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- epilogue of one column chunk: lane holds Y[token c of row tile rt][feature 16 * tile + g4 + 0..3]
  const int half = a.NT / 2;
  float best_v[RT];
  int best_i[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { best_v[rt] = -INFINITY; best_i[rt] = 0; }
  auto epilogue = [&](int chunk) {
    float* yrow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) yrow[rt] = a.y ? a.y + (size_t)tok[rt] * a.ldy : nullptr;
    if constexpr (EPI == E16_HEAD) {
      // logits (optional) + running arg-max over the classes this lane sees; the ring is padded to whole chunks, the
      // bias to NT tiles: tiles past NT and classes past n_valid do not exist
#pragma unroll
      for (int i = 0; i < GNB; ++i) {
        const int tile = chunk * GNB + i, f0 = 16 * tile + g4;
        if (tile < a.NT) {                    // wave-uniform
          const f32x4 bv = ldg4(a.bias + f0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const f32x4 v = acc[rt][i] + bv;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (f0 + j < a.n_valid && vv[j] > best_v[rt]) { best_v[rt] = vv[j]; best_i[rt] = f0 + j; }   // first maximum wins
            if (yrow[rt] && live[rt]) {
              if (f0 + 3 < a.n_valid && (a.ldy & 3) == 0) stg4(yrow[rt] + f0, v);
              else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (f0 + j < a.n_valid) yrow[rt][f0 + j] = vv[j];
            }
          }
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (EPI == E16_GLU) {
#pragma unroll
      for (int i = 0; i < GNB / 2; ++i) {
        const int f0 = 16 * (chunk * (GNB / 2) + i) + g4;
        const f32x4 ba = ldg4(a.bias + f0), bb = ldg4(a.bias + 16 * half + f0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 va = acc[rt][i] + ba, vb = acc[rt][GNB / 2 + i] + bb;
          const f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
          if (live[rt]) stg4(yrow[rt] + f0, o);
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);     // two tiles' parameters in registers at a time, not all eight
      }
    } else {
#pragma unroll
      for (int i = 0; i < GNB; ++i) {
        const int tile = chunk * GNB + i, f0 = 16 * tile + g4;
        const f32x4 bv = ldg4(a.bias + f0);
        f32x4 as = splat4(1.f), at = splat4(0.f);
        if constexpr (EPI == E16_AFFSWISH) { as = ldg4(a.aff_s + f0); at = ldg4(a.aff_t + f0); }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          f32x4 v = acc[rt][i] + bv;
          if constexpr (EPI == E16_SWISH) v = swish4(v);
          if constexpr (EPI == E16_AFFSWISH) v = swish4(v * as + at);
          if constexpr (EPI == E16_QKV) { if (tile < a.qtiles) v *= splat4(a.qscale); }
          if constexpr (EPI == E16_RES) v = ldg4(a.res + (size_t)min(tok[rt], a.M - 1) * a.ldy + f0) + splat4(a.scale) * v;
          if (live[rt]) stg4(yrow[rt] + f0, v);
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int n = 0; n < GNB; ++n) acc[rt][n] = splat4(0.f);
  };
  auto run = [&](auto LATE_T) {
    constexpr bool LATE = decltype(LATE_T)::value;
    if constexpr (LATE) {
      xsplit(0, xq[0], xa);
      xload(XST, xq[0]);
    }
#pragma unroll 1
    for (int cc = 0; cc < ncc; ++cc) {
      const int g0 = cc * steps;
#pragma unroll 1
      for (int s = 0; s < steps; s += GUNR) {
        step(s, g0 + s, std::integral_constant<int, 0>{}, LATE_T);
        step(s + 1, g0 + s + 1, std::integral_constant<int, 1>{}, LATE_T);
        step(s + 2, g0 + s + 2, std::integral_constant<int, 2>{}, LATE_T);
        step(s + 3, g0 + s + 3, std::integral_constant<int, 3>{}, LATE_T);
      }
      // The epilogue's loads and stores are VMEM operations the counted waits of the step loop do not know about: they
      // sit between the operand loads already in flight and the next slab pieces, so every count stays an upper bound
      // of what may be outstanding only if they are complete before the next step starts.
      epilogue(chunk0 + cc);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (wv >= GW / 2) run(std::integral_constant<bool, true>{});
  else run(std::integral_constant<bool, false>{});
  // the last iterations requested operands nobody uses; hipcc does not know that those registers are still being
  // written: keep them until the loads have landed
#pragma unroll
  for (int st = 0; st < XST; ++st) xwait(xq[st], std::integral_constant<int, 0>{});
  if constexpr (EPI == E16_HEAD) {
    // the four lane groups of a token hold disjoint classes: max over the groups, lowest class on ties
    if (a.argmax_out) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float bv = best_v[rt];
        int bi = best_i[rt];
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          const float ov = __shfl_xor(bv, off);
          const int oi = __shfl_xor(bi, off);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (live[rt] && lane < 16) {
          if (gridDim.y > 1) {                      // this range's winner; launch_head_combine takes the best of the ranges
            a.part_v[(size_t)blockIdx.y * a.M + tok[rt]] = bv;
            a.part_i[(size_t)blockIdx.y * a.M + tok[rt]] = bi;
          } else {
            a.argmax_out[tok[rt]] = bi;
          }
        }
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

template <int EPI, bool LN, int TERMS, bool EDMA>
int go(const Gemm16Args& a, const void* ring, hipStream_t s) {
  // the head's ring holds ceil(V / 128) chunks (api.hip: put_ring_head), whatever tile count the P16 pack of the same matrix
  // was padded to: a.NT = 24 at V = 200 walked a third chunk that is not there (garbage columns nobody looked at, but
  // 393 KB read past the ring -- past the arena when the ring is its last entry)
  const int chunks = EPI == E16_HEAD ? ((a.n_valid + 15) / 16 + GNB - 1) / GNB : (EPI == E16_GLU ? a.NT / 2 : a.NT) / (EPI == E16_GLU ? GNB / 2 : GNB);
  // Shape of the launch: RT row tiles per wave (256 or 128 rows per workgroup) and cpw column chunks per workgroup, so
  // that the workgroups come as close as possible to a whole number of rounds over the 256 CUs (short K: every
  // workgroup pays a prologue -- first slabs, LayerNorm statistics -- worth several k-steps, and a second, half-empty
  // round costs as much as a full one); among equals the larger tile.  MI355ASR_RING_RT / _SLOTS / _CPW force one shape
  // (tests).
  static const int force_rt = (int)mi355_env("MI355ASR_RING_RT", 0);
  static const int force_slots = (int)mi355_env("MI355ASR_RING_SLOTS", 0);
  static const int force_cpw = (int)mi355_env("MI355ASR_RING_CPW", 0);
  int best_rt = 1, best_cpw = 1;
  double best = 1e30;
  for (int rt = 2; rt >= 1; --rt) {
    if (EPI == E16_HEAD && rt == 2) continue;               // the head runs one row tile per wave (registers)
    if (EPI != E16_HEAD && force_rt && rt != force_rt) continue;
    const int rows = (a.M + 128 * rt - 1) / (128 * rt);
    for (int cpw = chunks; cpw >= 1; --cpw) {
      if (EPI == E16_HEAD && cpw != chunks) continue;       // the arg-max needs every class of a row in one workgroup
      if (chunks % cpw != 0 || (force_cpw && cpw != std::min(force_cpw, chunks) && chunks % std::min(force_cpw, chunks) == 0)) continue;
      const long wgs = (long)rows * (chunks / cpw);
      const long rounds = (wgs + 255) / 256;
      // time ~ rounds x (prologue worth ~6 steps + cpw x steps x rt), in units of one row-tile step
      const double cost = (double)rounds * (6.0 + (double)cpw * (a.K / 32) * rt + 1.5 * cpw);
      if (cost < best * 0.999) { best = cost; best_rt = rt; best_cpw = cpw; }
    }
  }
  int cpw = best_cpw;
  dim3 grid((a.M + 128 * best_rt - 1) / (128 * best_rt), chunks / cpw);
  if constexpr (EPI == E16_HEAD) {
    // Round 6: the rows alone are 130 workgroups at 16 640 rows (125 at 16 000) -- with scratch for the per-range winners the
    // chunks of a row tile go to several workgroups.  Same cost model; a CU takes two workgroups of the bf16-mode kernel (32 KB
    // ring), one of the three-term kernel.  MI355ASR_RING_HEAD_RANGES=n forces n ranges (1: the single-workgroup head).
    static const int force_nr = (int)mi355_env("MI355ASR_RING_HEAD_RANGES", 0);
    if (a.argmax_out && a.part_v && a.part_i && a.part_max > 1 && chunks > 1) {
      const long cap = 256L * (TERMS == 1 ? 2 : 1);
      int best_nr = 1;
      double best_c = 1e30;
      for (int nr = 1; nr <= std::min(chunks, a.part_max); ++nr) {
        const int c = (chunks + nr - 1) / nr;
        if ((chunks + c - 1) / c != nr) continue;                // (this many ranges do not come out with whole chunks)
        if (force_nr && nr != std::min(force_nr, std::min(chunks, a.part_max))) continue;
        const long rounds = ((long)grid.x * nr + cap - 1) / cap;
        const double cost = (double)rounds * (6.0 + (double)c * (a.K / 32) + 1.5 * c) + (nr > 1 ? 3.0 : 0.0);
        if (cost < best_c * 0.999) { best_c = cost; best_nr = nr; }
      }
      if (best_nr > 1) {
        Gemm16Args b = a;
        b.head_chunks = chunks;
        cpw = (chunks + best_nr - 1) / best_nr;
        grid.y = best_nr;
        if (force_slots == 2 || (force_slots != 4 && (long)grid.x * grid.y > 320))
          hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 1, 2, TERMS, EDMA>), grid, dim3(GT), 0, s, b, (const u32x4*)ring, cpw);
        else
          hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 1, 4, TERMS, EDMA>), grid, dim3(GT), 0, s, b, (const u32x4*)ring, cpw);
        return launch_head_combine(a.part_v, a.part_i, best_nr, a.M, a.argmax_out, nullptr, s);
      }
    }
  }

  if (best_rt == 2 && force_slots != 2) {
    if constexpr (EPI != E16_HEAD)
      hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 2, 4, TERMS, EDMA>), grid, dim3(GT), 0, s, a, (const u32x4*)ring, cpw);
  } else if (best_rt == 2)
    return -1;
  else if (force_slots == 2 || (force_slots != 4 && (long)grid.x * grid.y > 320))
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 1, 2, TERMS, EDMA>), grid, dim3(GT), 0, s, a, (const u32x4*)ring, cpw);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, LN, 1, 4, TERMS, EDMA>), grid, dim3(GT), 0, s, a, (const u32x4*)ring, cpw);
  return 0;
}

}  // namespace

// Shapes the ring kernel takes; the pack (api.hip: pack_ring) exists for the dense layers of dmodel 256 / 512 blocks.
bool gemm_ring_applicable(int epi, bool ln, const Gemm16Args& a) {
  if (a.K % 128 != 0 || a.K < 128 || (a.ldx & 3) != 0) return false;
  if (epi == E16_HEAD)                          // logits optional; the ring is padded to whole chunks of 8 tiles
    return !ln && a.NT >= 1 && a.n_valid <= 16 * a.NT && (a.y || a.argmax_out) && a.rpb == 0;
  const int ntc = epi == E16_GLU ? a.NT / 2 : a.NT;
  if (ntc % (epi == E16_GLU ? 4 : 8) != 0 || a.n_valid != (epi == E16_GLU ? 16 * ntc : 16 * a.NT)) return false;
  if (ln && a.K > GLN_MAX) return false;
  if ((a.ldy & 3) != 0 || !a.y) return false;
  switch (epi) {
    case E16_BIAS: return !ln;
    case E16_SWISH: case E16_QKV: case E16_GLU: return ln;
    case E16_RES: case E16_AFFSWISH: return !ln;
    default: return false;
  }
}

template <int TERMS, bool EDMA>
static int launch_terms(int epi, bool ln, const Gemm16Args& a, const void* ring, hipStream_t s) {
  switch (epi) {
    case E16_BIAS: return go<E16_BIAS, false, TERMS, EDMA>(a, ring, s);
    case E16_SWISH: return go<E16_SWISH, true, TERMS, EDMA>(a, ring, s);
    case E16_QKV: return go<E16_QKV, true, TERMS, EDMA>(a, ring, s);
    case E16_GLU: return go<E16_GLU, true, TERMS, EDMA>(a, ring, s);
    case E16_AFFSWISH: return go<E16_AFFSWISH, false, TERMS, EDMA>(a, ring, s);
    case E16_HEAD: return go<E16_HEAD, false, TERMS, EDMA>(a, ring, s);
    case E16_RES: {
      const int rc = go<E16_RES, false, TERMS, EDMA>(a, ring, s);
      // the optional LayerNorm over the output row needs all of it: a second pass over y (as bf16.hip does for wide rows)
      if (rc == 0 && a.fln_g)
        hipLaunchKernelGGL(ring_layernorm_rows_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a.y, a.fln_g, a.fln_b, a.M, 16 * a.NT, a.ldy, a.eps);
      return rc;
    }
    default: return -1;
  }
}

// terms = 3: fp32 operands exactly split; 1: bf16 mode (ring of host-rounded bf16 weights, activations rounded at the operand)
int launch_gemm_ring(int epi, bool ln, const Gemm16Args& a, const void* ring, int terms, hipStream_t s) {
  if (!ring || !gemm_ring_applicable(epi, ln, a)) return -1;
  note_scheme(terms == 1 ? SCHEME_BF16 : SCHEME_BF16X3);
  // (EDMA = true: only the lower four waves issue slab DMAs; the every-wave version of round 2 lost and its instantiations are gone)
  return terms == 1 ? launch_terms<1, true>(epi, ln, a, ring, s) : launch_terms<3, true>(epi, ln, a, ring, s);
}
