// Pair-pipelined block kernels for dmodel 144 (round 3): the token-local runs of a ConformerBlock that fused.hip walks as
// hidden CHUNKS of nine tiles (W1 x 5 slabs -> activation + split -> W2 x 5 slabs) are walked here in hidden PAIRS of
// tiles, software-pipelined three deep:
//
//     unit p:   y      += W2[pair p]^T  hf(p)                 B(p)      54 MFMAs, 27 fragments
//               h(p+2)  = W1[:, pair p + 2]^T xf              A(p + 2)  60 MFMAs, 30 fragments
//               hf(p+1) = split(swish(h(p + 1)))              prep      48 slots of <= 2 VALU instructions behind the MFMAs
//
// What that buys over the round-2 kernels (profiles/r02_ring_experiments.md: 17 of tail_ff1's 85 us were activation /
// split VALU work that only the five W2 slabs of a chunk could carry -- 1.7 VALU per MFMA against ~1 that is free):
//   * the VALU work of a pair is spread over the 114 MFMAs of a unit: 0.8 instructions per MFMA, uniformly;
//   * 32 hidden features are exactly one 32-wide k-step of W2: no ninth tile paired with zeros (FFN: 2052 MFMAs, was 2160;
//     conv tail 1026, was 1080);
//   * bias, and for the conv module the folded BatchNorm, are part of the weight stream: W1 carries the BatchNorm scale in its
//     columns and (bias * scale + shift) in row 144 -- the K padding of the fifth k-step -- against a constant 1.0 operand, so
//     the loop reads no parameter from LDS and the accumulators start from zero;
//   * h is 2 x 2 tiles instead of 9 (+ 1 padding) tiles.
// The stream itself (fragment order, pool slots, counted waits, ring-slot hand-over) is generated and checked by
// tools/gen_pp.py -> pp_units.inc (device) / pp_layout.inc (host packing, api.hip: append_pp_chain).
// Workgroup = 8 waves as in fused.hip: waves 0-3 consume (16 tokens each), waves 4-7 issue the slab DMAs; ring = 5 slots of
// 30 fragments (30 KB).  Reference semantics: asr/models/conformer_blocks.py:126-134, :164-170, :209-219, :259-265.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "launch.h"
#include "wstream.h"

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;      // 9
constexpr int KS32X = 5;        // 32-wide steps over K = 144 (+ the bias row 144)
constexpr int LD_THREADS = 2 * BLOCK_THREADS;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct Split8 { u32x4_t t[3]; };

DEV Split8 split8(f32x4 lo, f32x4 hi) {      // exact: x = t0 + t1 + t2 (truncation, remainders are exact)
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
DEV void dma16(const u32x4_t* gsrc, u32x4_t* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// s_waitcnt vmcnt(PER * ahead) lgkmcnt(0) for a wave-uniform run-time `ahead` in [0, MAXA] (the count is an immediate)
template <int PER, int MAXA>
DEV void wait_dma_ahead(int ahead) {
  static_for<0, MAXA + 1>([&](auto K) {
    constexpr int k = decltype(K)::value, n = PER * k;
    if (ahead == k) __builtin_amdgcn_s_waitcnt(0x0070 | (n & 15) | ((n >> 4) << 14));
  });
}
template <int N>
struct StashRegs { float r[(N + BLOCK_THREADS - 1) / BLOCK_THREADS]; };
template <int N>
DEV StashRegs<N> stash_load(const float* __restrict__ src) {
  StashRegs<N> s;
#pragma unroll
  for (int k = 0; k < (N + BLOCK_THREADS - 1) / BLOCK_THREADS; ++k) {
    const int idx = threadIdx.x + BLOCK_THREADS * k;
    s.r[k] = idx < N ? src[idx] : 0.f;
  }
  return s;
}
template <int N>
DEV void stash_store(float* dst, const StashRegs<N>& s) {
#pragma unroll
  for (int k = 0; k < (N + BLOCK_THREADS - 1) / BLOCK_THREADS; ++k) {
    const int idx = threadIdx.x + BLOCK_THREADS * k;
    if (idx < N) dst[idx] = s.r[k];
  }
}
template <int OFF>
DEV u32x4_t lds_read16(unsigned addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

struct WaveCtx {
  int lane, g4, t, tok;
  size_t row;
  bool live;
};
DEV WaveCtx wave_ctx(int M) {
  WaveCtx c;
  c.lane = threadIdx.x & 63;
  c.g4 = (c.lane >> 4) * 4;
  c.t = c.lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  c.tok = wid * 16 + c.t;
  c.live = c.tok < M;
  c.row = (size_t)min(c.tok, M - 1) * D;     // waves past the end recompute the last token and store nothing
  return c;
}
// recomputed from an opaque copy of the thread index before the stores (keeps the 64-bit row offset out of the stream's
// live ranges: it was the value the register allocator spilled in the round-2 kernels)
DEV WaveCtx wave_ctx_fresh(int M) {
  unsigned tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  WaveCtx c;
  c.lane = tid & 63;
  c.g4 = (c.lane >> 4) * 4;
  c.t = c.lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (int)(tid >> 6);
  c.tok = wid * 16 + c.t;
  c.live = c.tok < M;
  c.row = (size_t)min(c.tok, M - 1) * D;
  return c;
}
DEV void ln_lds(f32x4 (&xs)[KB], const float* ga, const float* be, int g4, float eps) {
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * lds4(ga, kb, g4) + lds4(be, kb, g4);
}

// activation + exact three-term split of two finished hidden tiles, in slots of <= 2 instructions (prep_sched.inc)
struct PrepCtx {
  f32x4 &lo, &hi;
  const f32x4 &slo, &tlo, &shi, &thi;   // unused here (AFF = false): the BatchNorm is folded into the weight stream
  Split8& out;
  float ta, tb, m0, m1;
};
template <bool AFF, bool FULL> struct PrepSlots;
#include "prep_sched.inc"

// ---- the generated units ---------------------------------------------------------------------------------------------------
struct PpPool { u32x4_t f[9]; };
#define PP_RD(S, A, OFF) pl.f[S] = lds_read16<OFF>(A)
#define PP_WT0(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
#define PP_WT2(N, S0, S1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(pl.f[S0]), "+v"(pl.f[S1]))
#define PP_WT3(N, S0, S1, S2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(pl.f[S0]), "+v"(pl.f[S1]), "+v"(pl.f[S2]))
#define PP_MM(ACC, S, X) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pl.f[S]), __builtin_bit_cast(bf16x8_t, X), ACC, 0, 0, 0)
#define PP_PREP(K) prep_slot<K, false, true>(pc)
#define PP_FENCE __builtin_amdgcn_sched_barrier(0)
#include "pp_units.inc"

constexpr int PP_FR = 30;                 // fragments per ring slot (pp_layout.inc: kPpSlot)
constexpr int PP_SLB = PP_FR * 64;        // u32x4 per ring slot (30 KB)
constexpr int PP_RING = 5;

// waves 4..7: fragment f of a slab is fetched by loader wave f % 4 -- waves 0, 1 issue eight 1 KB pieces per slab, waves 2, 3
// seven -- into the ring slot the consumers read in the previous step; "slab s + 2 has landed" (this wave's pieces: counted
// vmcnt) before the barrier that ends step s, so that the consumers' fragment pipeline may run into slab s + 1 during step s.
template <int RING>
struct PpLoader {
  u32x4_t* ring;
  const u32x4_t *src, *src2;    // slabs [0, n1) from src, [n1, total) from src2
  int n1, total, wv, lane;      // wv = 0..3
  template <int PER>
  DEV void issue(int slab, int slot) const {
    const u32x4_t* g = (slab < n1 ? src + (size_t)slab * PP_SLB : src2 + (size_t)(slab - n1) * PP_SLB) + 64 * wv + lane;
    u32x4_t* l = ring + slot * PP_SLB + 64 * wv;
#pragma unroll
    for (int q = 0; q < PER; ++q) dma16(g + BLOCK_THREADS * q, l + BLOCK_THREADS * q);
  }
  template <int PER>
  DEV void run_() const {
    static_assert(RING >= 4, "slab s + 1 is read ahead while slab s + RING - 1 is written");
    const int pre = min(RING - 1, total);
    for (int i = 0; i < pre; ++i) issue<PER>(i, i);
    wait_dma_ahead<PER, RING - 3>(min(RING - 3, max(pre - 2, 0)));   // slabs 0 and 1 have landed
    __builtin_amdgcn_s_barrier();                                    // B0 (consumers: inputs + parameter stash)
    int rd = 0;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
      if (s + RING - 1 < total) issue<PER>(s + RING - 1, rd == 0 ? RING - 1 : rd - 1);
      wait_dma_ahead<PER, RING - 3>(max(min(RING - 3, total - 3 - s), 0));     // slab s + 2 has landed
      __builtin_amdgcn_s_barrier();
      rd = rd + 1 == RING ? 0 : rd + 1;
    }
  }
  DEV void run() const {
    if (wv < 2) run_<8>(); else run_<7>();
  }
};

template <int RING>
struct PpReader {               // waves 0..3
  u32x4_t* ring;
  int lane;
  int rd = 0;
  DEV void sync() const {       // this wave's LDS writes (parameter stash) are done; B0
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  }
  DEV unsigned slot_addr(int slot) const {
    return (unsigned)(size_t)(__attribute__((address_space(3))) void*)(ring + slot * PP_SLB + lane);
  }
  DEV unsigned cur_addr() const { return slot_addr(rd); }
  DEV unsigned next_addr() const { return slot_addr(rd + 1 == RING ? 0 : rd + 1); }
  DEV void advance() {          // every read of the slot has landed (the generated wait in front of each call covers them)
    __builtin_amdgcn_s_barrier();
    rd = rd + 1 == RING ? 0 : rd + 1;
    __builtin_amdgcn_sched_barrier(0);
  }
};

// the first nine fragments of the first slab (the state every unit starts from and leaves behind for the next slab)
template <class ST>
DEV void pp_prime(PpPool& pl, ST& st) {
  const unsigned a0 = st.cur_addr();
  PP_RD(0, a0, 0 * 1024); PP_RD(1, a0, 1 * 1024); PP_RD(2, a0, 2 * 1024);
  PP_RD(3, a0, 3 * 1024); PP_RD(4, a0, 4 * 1024); PP_RD(5, a0, 5 * 1024);
  PP_RD(6, a0, 6 * 1024); PP_RD(7, a0, 7 * 1024); PP_RD(8, a0, 8 * 1024);
}

// y += W2 swish(W1aug [x ; 1]) over P hidden pairs (2 P ring slots): units A, AP, P - 2 x F, BP, B.  h0 / h1 and f0 / f1
// swap roles from unit to unit (accumulate <-> being prepared, operand in use <-> operand being built).
template <int P, class ST>
DEV void pp_chain(f32x4 (&y)[KB], const Split8 (&xf)[KS32X], PpPool& pl, ST& st) {
  static_assert(P >= 3, "at least one full unit");
  f32x4 h0[2], h1[2];
  Split8 f0, f1;
  const f32x4 one = splat4(1.f), zero = splat4(0.f);
  h0[0] = zero; h0[1] = zero;
  pp_unit_A(h0, xf, pl, st);                                   // h(0)
  h1[0] = zero; h1[1] = zero;
  {
    PrepCtx pc{h0[0], h0[1], one, zero, one, zero, f0, 0.f, 0.f, 0.f, 0.f};
    pp_unit_AP(h1, xf, pc, pl, st);                            // h(1) ; hf(0)
  }
  constexpr int NF = P - 2;
#pragma unroll 1
  for (int i = 0; i < NF / 2; ++i) {
    h0[0] = zero; h0[1] = zero;
    {
      PrepCtx pc{h1[0], h1[1], one, zero, one, zero, f1, 0.f, 0.f, 0.f, 0.f};
      pp_unit_F(y, h0, xf, f0, pc, pl, st);                    // p even: B(p) with f0, A(p + 2) -> h0, h1 -> f1
    }
    h1[0] = zero; h1[1] = zero;
    {
      PrepCtx pc{h0[0], h0[1], one, zero, one, zero, f0, 0.f, 0.f, 0.f, 0.f};
      pp_unit_F(y, h1, xf, f1, pc, pl, st);                    // p odd: B(p) with f1, A(p + 2) -> h1, h0 -> f0
    }
  }
  if constexpr (NF & 1) {
    h0[0] = zero; h0[1] = zero;
    {
      PrepCtx pc{h1[0], h1[1], one, zero, one, zero, f1, 0.f, 0.f, 0.f, 0.f};
      pp_unit_F(y, h0, xf, f0, pc, pl, st);
    }
    {
      PrepCtx pc{h0[0], h0[1], one, zero, one, zero, f0, 0.f, 0.f, 0.f, 0.f};
      pp_unit_BP(y, f1, pc, pl, st);                           // B(P - 2) with f1 ; h(P - 1) -> f0
    }
    pp_unit_B(y, f0, pl, st);
  } else {
    {
      PrepCtx pc{h1[0], h1[1], one, zero, one, zero, f1, 0.f, 0.f, 0.f, 0.f};
      pp_unit_BP(y, f0, pc, pl, st);                           // B(P - 2) with f0 ; h(P - 1) -> f1
    }
    pp_unit_B(y, f1, pl, st);
  }
}

// the five split operands of a 16-token tile for the W1 steps; k-slot 144 (the first padding slot of step 4) carries 1.0:
// row 144 of the streamed W1 holds the bias
DEV void split_operand(Split8 (&xf)[KS32X], const f32x4 (&xs)[KB], int g4) {
#pragma unroll
  for (int t = 0; t < KS32X - 1; ++t) xf[t] = split8(xs[2 * t], xs[2 * t + 1]);
  f32x4 oh = splat4(0.f);
  oh.x = g4 == 0 ? 1.0f : 0.0f;
  xf[KS32X - 1] = split8(xs[KB - 1], oh);
}

struct PpTailLds { float pw2b[D], lng[D], lnb[D], b2[D], fg[D], fb[D]; };
struct PpFf1Lds { float ln1g[D], ln1b[D], b2[D], ln2g[D], ln2b[D]; };

DEV void pp_tail_stash(PpTailLds& p, const TailFf2Args& a) {
  const auto r0 = stash_load<D>(a.pw2_b), r1 = stash_load<D>(a.ff_ln_g), r2 = stash_load<D>(a.ff_ln_b),
             r3 = stash_load<D>(a.ff_b2), r4 = stash_load<D>(a.ln_g), r5 = stash_load<D>(a.ln_b);
  stash_store<D>(p.pw2b, r0); stash_store<D>(p.lng, r1); stash_store<D>(p.lnb, r2);
  stash_store<D>(p.b2, r3); stash_store<D>(p.fg, r4); stash_store<D>(p.fb, r5);
}
DEV void pp_ff1_stash(PpFf1Lds& p, const Ff1QkvArgs& a) {
  const auto r0 = stash_load<D>(a.ff_ln_g), r1 = stash_load<D>(a.ff_ln_b), r2 = stash_load<D>(a.ff_b2),
             r3 = stash_load<D>(a.att_ln_g), r4 = stash_load<D>(a.att_ln_b);
  stash_store<D>(p.ln1g, r0); stash_store<D>(p.ln1b, r1); stash_store<D>(p.b2, r2);
  stash_store<D>(p.ln2g, r3); stash_store<D>(p.ln2b, r4);
}

constexpr int PP_TAIL_SLABS = 2 * 9 + 2 * 18;     // conv tail (9 pairs) + ff_module_2 (18 pairs)
constexpr int PP_FF1_SLABS = 2 * 18 + 3 * KS32X;  // ff_module_1 (18 pairs) + q, k, v (five plain steps each)

// conv-module tail + ff_module_2 + block-final LayerNorm: xs = dw rows, y = x2 rows on entry; y = the block's output on exit
template <class ST>
DEV void pp_tail_consume(const TailFf2Args& a, const PpTailLds& p, int g4, ST& st, PpPool& pl, f32x4 (&xs)[KB], f32x4 (&y)[KB]) {
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] += lds4(p.pw2b, i, g4);
  Split8 xf[KS32X];
  split_operand(xf, xs, g4);
  pp_chain<9>(y, xf, pl, st);                                                        // x3 = x2 + conv module
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    xs[i] = y[i];
    y[i] = lds4(p.b2, i, g4) + splat4(inv_fc) * y[i];                                // x3 / fc + b2 (+ W2 h)
  }
  ln_lds(xs, p.lng, p.lnb, g4, a.eps);
  split_operand(xf, xs, g4);
  pp_chain<18>(y, xf, pl, st);
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = splat4(a.fc) * y[i];
  ln_lds(y, p.fg, p.fb, g4, a.eps);                                                  // block-final LayerNorm
}

// ff_module_1 + q / k / v projections of the 16 tokens in xs (x0 rows); stores x1 and qkv
template <class ST>
DEV void pp_ff1_consume(const Ff1QkvArgs& a, const PpFf1Lds& p, int g4, ST& st, PpPool& pl, f32x4 (&xs)[KB]) {
  f32x4 y[KB];
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = lds4(p.b2, kb, g4) + splat4(inv_fc) * xs[kb];
  ln_lds(xs, p.ln1g, p.ln1b, g4, a.eps);
  Split8 xf[KS32X];
  split_operand(xf, xs, g4);
  pp_chain<18>(y, xf, pl, st);
#pragma unroll
  for (int i = 0; i < KB; ++i) { y[i] = splat4(a.fc) * y[i]; xs[i] = y[i]; }        // x1 = x0 + fc * (ffn + b2)
  {
    const WaveCtx e = wave_ctx_fresh(a.M);
    if (e.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(a.x1 + e.row + 16 * i + e.g4, y[i]);
    }
  }
  ln_lds(xs, p.ln2g, p.ln2b, g4, a.eps);
  split_operand(xf, xs, g4);                                                         // the q / k / v bias rides in row 144 too
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      pp_unit_S(acc, xf[t], pl, st);
    });
    const float sc = q == 0 ? a.qscale : 1.0f;
    const WaveCtx e = wave_ctx_fresh(a.M);
    if (e.live) {
      float* qrow = a.qkv + (size_t)e.tok * (3 * D) + 16 * q * KB + e.g4;
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(qrow + 16 * i, acc[i] * splat4(sc));
    }
  }
}

// TAIL: conv tail + ff_module_2 + LayerNorm of one block (a);  FF1: ff_module_1 + qkv of a block (b) -- of the NEXT block
// when both are set (the block output stays in registers; a.y may be null then)
template <bool TAIL, bool FF1>
__global__ __launch_bounds__(LD_THREADS) void pp_block_kernel(TailFf2Args a, Ff1QkvArgs b) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[PP_RING * PP_SLB];
  __shared__ __attribute__((aligned(16))) PpTailLds pt;
  __shared__ __attribute__((aligned(16))) PpFf1Lds pf;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int N1 = TAIL ? PP_TAIL_SLABS : PP_FF1_SLABS, TOTAL = (TAIL ? PP_TAIL_SLABS : 0) + (FF1 ? PP_FF1_SLABS : 0);
  if (wv >= WAVES_PER_BLOCK) {
    const u32x4_t* s1 = reinterpret_cast<const u32x4_t*>(TAIL ? a.pp_slabs : b.pp_slabs);
    const u32x4_t* s2 = reinterpret_cast<const u32x4_t*>(b.pp_slabs);
    PpLoader<PP_RING>{ring, s1, s2, N1, TOTAL, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  const int M = TAIL ? a.M : b.M;
  const WaveCtx c = wave_ctx(M);
  PpReader<PP_RING> st{ring, c.lane};
  f32x4 xs[KB], y[KB];
  if constexpr (TAIL) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + c.row + 16 * kb + c.g4);     // residuals ride in the accumulators
    pp_tail_stash(pt, a);
  } else {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(b.x0 + c.row + 16 * kb + c.g4);
  }
  if constexpr (FF1) pp_ff1_stash(pf, b);
  st.sync();
  PpPool pl;
  pp_prime(pl, st);
  if constexpr (TAIL) {
    pp_tail_consume(a, pt, c.g4, st, pl, xs, y);
    if (a.y) {
      const WaveCtx e = wave_ctx_fresh(M);
      if (e.live) {
#pragma unroll
        for (int i = 0; i < KB; ++i) stg4(a.y + e.row + 16 * i + e.g4, y[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < KB; ++i) xs[i] = y[i];
  }
  if constexpr (FF1) pp_ff1_consume(b, pf, c.g4, st, pl, xs);
  // the pool still holds nine reads of the slot after the last one: drain them before the wave ends
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

}  // namespace

bool pp_enabled() {
  // MI355ASR_PP=0: the round-2 chunk-wise ring kernels (fused.hip) instead of the pair-pipelined ones
  static const bool on = [] { const char* v = getenv("MI355ASR_PP"); return v ? atoi(v) != 0 : true; }();
  return on;
}
int launch_pp_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, hipStream_t s) {
  if (!pp_enabled() || !a.pp_slabs || !b.pp_slabs || a.M != b.M || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
  hipLaunchKernelGGL((pp_block_kernel<true, true>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, b);
  return 0;
}
int launch_pp_tail_ff2(const TailFf2Args& a, hipStream_t s) {
  if (!pp_enabled() || !a.pp_slabs || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
  hipLaunchKernelGGL((pp_block_kernel<true, false>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, Ff1QkvArgs{});
  return 0;
}
int launch_pp_ff1_qkv(const Ff1QkvArgs& b, hipStream_t s) {
  if (!pp_enabled() || !b.pp_slabs || b.M <= 0) return -1;
  const int tiles = (b.M + 15) / 16;
  hipLaunchKernelGGL((pp_block_kernel<false, true>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, TailFf2Args{}, b);
  return 0;
}
