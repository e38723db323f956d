// Pair-pipelined block kernels for dmodel 144 (round 3): the token-local runs of a ConformerBlock that fused.hip walks as
// hidden CHUNKS of nine tiles (W1 x 5 slabs -> activation + split -> W2 x 5 slabs) are walked here in hidden PAIRS of
// tiles, software-pipelined three deep:
//
//     unit p:   y      += W2[pair p]^T  hf(p)                 B(p)      27 MFMAs, 18 fragments
//               h(p+2)  = W1[:, pair p + 2]^T xf              A(p + 2)  30 MFMAs, 20 fragments
//               hf(p+1) = split(swish(h(p + 1)))              prep      40 slots of <= 2 VALU instructions behind the MFMAs
//
// (counts of the two-term operand scheme below; the first version of this file ran three bf16 terms: 54 / 60 MFMAs, 27 / 30
// fragments, 48 slots.)  What the pairing buys over the round-2 kernels (profiles/r02_ring_experiments.md: 17 of tail_ff1's
// 85 us were activation / split VALU work that only the five W2 slabs of a chunk could carry):
//   * the VALU work of a pair is spread over the MFMAs of a unit, uniformly;
//   * 32 hidden features are exactly one 32-wide k-step of W2: no ninth tile paired with zeros;
//   * bias, and for the conv module the folded BatchNorm, are part of the weight stream: W1 carries the BatchNorm scale in its
//     columns and (bias * scale + shift) in row 144 -- the K padding of the fifth k-step -- against the operand's unit in that
//     k-slot, so the loop reads no parameter from LDS and the accumulators start from zero;
//   * h is 2 x 2 tiles instead of 9 (+ 1 padding) tiles.
// The stream itself (fragment order, pool slots, counted waits, ring-slot hand-over) is generated and checked by
// tools/gen_pp.py -> pp_units.inc (device) / pp_layout.inc (host packing, api.hip: append_pp_chain).
// Workgroup = 8 waves as in fused.hip: waves 0-3 consume (16 tokens each), waves 4-7 issue the slab DMAs; ring = 7 slots of
// 20 fragments (20 KB).  Reference semantics: asr/models/conformer_blocks.py:126-134, :164-170, :209-219, :259-265.
//
// Operand scheme (round 3, second half): TWO fp16 terms, THREE products.  A weight matrix is streamed as hi + lo fp16 terms of
// W * sw (sw = the power of two that puts max |W| in [2^14, 2^15); api.hip: append_pp_chain), an operand row x as hi + lo of
// x * sx with sx the power of two that puts the ROW's largest magnitude in [2^13, 2^14) (pp_row_max: the rows are this lane's
// token, so every scale is a per-lane register), hi = fp16(v), lo = fp16(v - hi), both round-to-nearest: |v - hi - lo| <=
// 2^-22 |v| while lo is a normal fp16 (|v| >= 2^-2), 2^-25 absolute below.  a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi: three
// v_mfma_f32_16x16x32_f16 per fragment pair; what is dropped (a_lo b_lo, the representation errors) is of the size of the
// terms the six-product bf16 scheme dropped -- both sit at the same distance from the fp64 oracle (the error-corrected fp16
// GEMM of Ootomo & Yokota; their second accumulator for the correction terms is what the power-of-two scales replace).
// Accumulators carry sw * sx (hidden tiles) or sw2 * sh (outputs; sh = the power of two for the bound L1 * max|x| + max|b| of
// the token's hidden row); the activation schedule takes the unit out inside its first and third instruction
// (prep2_sched.inc), residuals are multiplied into the accumulator's unit when they are loaded, outputs are multiplied out
// of it once per chain: powers of two, exact.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "env.h"
#include "launch.h"
#include "wstream.h"
#include "split_f16.h"

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;      // 9
constexpr int KS32X = 5;        // 32-wide steps over K = 144 (+ the bias row 144)
constexpr int LD_THREADS = 2 * BLOCK_THREADS;

#ifndef PP_DMA_AUX
#define PP_DMA_AUX 0        // cache-policy bits of the slab DMAs (experiments: tools/build_variant.py ... -DPP_DMA_AUX=n)
#endif
#ifndef PP_DMA_SLEEP
#define PP_DMA_SLEEP 0      // s_sleep units (64 cycles) between the pieces a loader wave issues (experiments)
#endif
DEV void dma16(const u32x4_t* gsrc, u32x4_t* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, PP_DMA_AUX);
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
// T > 0: utterance-aligned tiling (depthwise conv folded in): workgroup (blockIdx.x, blockIdx.y) = 64-frame chunk blockIdx.x of
// utterance blockIdx.y, so that the conv window of a workgroup never crosses an utterance; T = 0: tokens tiled flat
DEV void tok_of(WaveCtx& c, unsigned tid, int M, int T) {
  c.lane = tid & 63;
  c.g4 = (c.lane >> 4) * 4;
  c.t = c.lane & 15;
  const int wv = (int)(tid >> 6);
  if (T > 0) {
    const int f = (blockIdx.x * WAVES_PER_BLOCK + wv) * 16 + c.t;      // frame inside the utterance
    c.live = f < T;
    c.tok = blockIdx.y * T + min(f, T - 1);                             // frames past the end recompute the last one
    c.row = (size_t)c.tok * D;
    return;
  }
  const int wid = blockIdx.x * WAVES_PER_BLOCK + wv;
  c.tok = wid * 16 + c.t;
  c.live = c.tok < M;
  c.row = (size_t)min(c.tok, M - 1) * D;     // waves past the end recompute the last token and store nothing
  if (!c.live) c.tok = M - 1;
}
DEV WaveCtx wave_ctx(int M, int T = 0) {
  WaveCtx c;
  tok_of(c, threadIdx.x, M, T);
  return c;
}
// recomputed from an opaque copy of the thread index before the stores (keeps the 64-bit row offset out of the stream's
// live ranges: it was the value the register allocator spilled in the round-2 kernels)
DEV WaveCtx wave_ctx_fresh(int M, int T = 0) {
  unsigned tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  WaveCtx c;
  tok_of(c, tid, M, T);
  return c;
}
DEV void ln_lds(f32x4 (&xs)[KB], const float* ga, const float* be, int g4, float eps) {
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * lds4(ga, kb, g4) + lds4(be, kb, g4);
}

DEV void pp_fake_prep(PpPrep& pc) {      // DG bit 0: consumes the hidden tiles and defines the operand without any work
  asm volatile("; no prep" : "=v"(pc.out.t[0]), "=v"(pc.out.t[1]) : "v"(pc.lo), "v"(pc.hi));
}
// ---- the generated units ---------------------------------------------------------------------------------------------------
// DG (diagnostics, timing only -- wrong results; instantiated only with -DMI355ASR_DIAG_KERNELS and selected by
// MI355ASR_PP_DIAG): bit 0 = no activation / split work, 1 = no fragment reads, 2 = no MFMAs, 3 = no barriers, 4 = no DMA
// (the diagnostic stand-ins are opaque definitions: with plain constants the compiler merges MFMAs that become identical)
#define PP_RD(S, A, OFF) \
  do { if constexpr (!(DG & 2)) pl.f[S] = lds_read16<OFF>(A); else asm volatile("; no read" : "=v"(pl.f[S]) : "v"(A)); } while (0)
#define PP_WT0(N) do { if constexpr (!(DG & 2)) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); } while (0)
#define PP_WT2(N, S0, S1) \
  do { if constexpr (!(DG & 2)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(pl.f[S0]), "+v"(pl.f[S1])); } while (0)
#define PP_WT3(N, S0, S1, S2) \
  do { if constexpr (!(DG & 2)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(pl.f[S0]), "+v"(pl.f[S1]), "+v"(pl.f[S2])); } while (0)
#define PP_MM(ACC, S, X) \
  do { if constexpr (!(DG & 4)) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, pl.f[S]), __builtin_bit_cast(f16x8_t, X), ACC, 0, 0, 0); } while (0)
// bit 6 (timing only): without the three products of order 2^-16 -- the MFMA count of a two-term operand scheme
#define PP_MM2(ACC, S, X) do { if constexpr (!(DG & 64)) PP_MM(ACC, S, X); } while (0)
#define PP_PREP(K) \
  do { if constexpr (!(DG & 1)) prep2_slot<K>(pc); else if constexpr (K == PREP2_SLOTS - 1) pp_fake_prep(pc); } while (0)
#define PP_FENCE __builtin_amdgcn_sched_barrier(0)
#include "pp_units.inc"
// Every unit leaves the first PP_NPOOL fragments of the next slab in flight into the pool registers (asm reads the compiler knows
// nothing about).  Unit follows unit without a gap inside a chain; wherever OTHER code follows -- epilogues, LayerNorm, the
// operand split of the next chain -- the reads are landed first, with the pool tied to the wait: code the compiler is free to
// schedule must never see a pool register that is still being written (it may copy it).
DEV void pp_pool_land(PpPool& pl) {
  static_assert(PP_NPOOL == 10, "one operand per pool register");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(pl.f[0]), "+v"(pl.f[1]), "+v"(pl.f[2]), "+v"(pl.f[3]), "+v"(pl.f[4]), "+v"(pl.f[5]), "+v"(pl.f[6]), "+v"(pl.f[7]),
                 "+v"(pl.f[8]), "+v"(pl.f[9]));
}

constexpr int PP_FR = 20;                 // fragments per ring slot (pp_layout.inc: kPpSlot)
constexpr int PP_SLB = PP_FR * 64;        // u32x4 per ring slot (20 KB)
constexpr int PP_RING = 7;
constexpr int PP_PER = PP_FR / 4;         // 1 KB pieces of a slab per loader wave

// waves 4..7: fragment f of a slab is fetched by loader wave f % 4 -- five 1 KB pieces per slab and wave -- into the ring slot the consumers read in the previous step; "slab s + 2 has landed" (this wave's pieces: counted
// vmcnt) before the barrier that ends step s, so that the consumers' fragment pipeline may run into slab s + 1 during step s.
template <int RING, int DG>
struct PpLoader {
  u32x4_t* ring;
  const u32x4_t *src, *src2;    // slabs [n0, n0 + n1) from src, [n0 + n1, total) from src2
  int n1, total, wv, lane;      // wv = 0..3; n1 and total count from the first slab of the stream (src0's included)
  const u32x4_t* src0 = nullptr;   // slabs [0, n0): the out-projection + GLU stream in front (OGF kernels)
  int n0 = 0;
  DEV const u32x4_t* slab_ptr(int slab) const {
    if (slab < n0) return src0 + (size_t)slab * PP_SLB;
    return slab < n1 ? src + (size_t)(slab - n0) * PP_SLB : src2 + (size_t)(slab - n1) * PP_SLB;
  }
  template <int PER>
  DEV void issue(int slab, int slot) const {
    const u32x4_t* g = slab_ptr(slab) + 64 * wv + lane;
    u32x4_t* l = ring + slot * PP_SLB + 64 * wv;
    if constexpr (DG & 16) return;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      dma16(g + BLOCK_THREADS * q, l + BLOCK_THREADS * q);
      if constexpr (PP_DMA_SLEEP > 0) __builtin_amdgcn_s_sleep(PP_DMA_SLEEP);
    }
  }
  // prologue slabs (depthwise-conv fold): all thirty pieces from waves 6 and 7, so that waves 4 and 5 -- which stage and
  // compute with the consumers -- have no DMA in flight (hipcc waits for every pending LDS-DMA before an LDS read it sees)
  DEV void issue_pro(int slab, int slot) const {
    if (wv < 2) return;
    const u32x4_t* g = slab_ptr(slab) + 64 * (wv - 2) + lane;
    u32x4_t* l = ring + slot * PP_SLB + 64 * (wv - 2);
#pragma unroll
    for (int q = 0; q < PP_FR / 2; ++q) dma16(g + 128 * q, l + 128 * q);
  }
  // pro(): work between the first slabs and the rest of the prefill (the depthwise-conv prologue: PRE0 = 2, the LDS of
  // ring slots 2.. is scratch until pro() returns); every barrier inside pro() has its twin on the consumer side
  template <int PER, int PRE0, class PRO>
  DEV void run_(PRO&& pro) const {
    static_assert(RING >= 4, "slab s + 1 is read ahead while slab s + RING - 1 is written");
    const int pre = min(RING - 1, total);
    for (int i = 0; i < min(PRE0, pre); ++i) {
      if constexpr (PRE0 == RING - 1) issue<PER>(i, i); else issue_pro(i, i);
    }
    pro();
    for (int i = min(PRE0, pre); i < pre; ++i) issue<PER>(i, i);
    wait_dma_ahead<PER, RING - 3>(min(RING - 3, max(pre - 2, 0)));   // slabs 0 and 1 have landed
    __builtin_amdgcn_s_barrier();                                    // B0 (consumers: inputs + parameter stash)
    main_loop<PER>(0, 0);
  }
  // steps s0 .. total - 1 of the steady state: slabs up to s0 + RING - 2 have been issued, slot rd holds slab s0
  template <int PER>
  DEV void main_loop(int s0, int rd) const {
#pragma unroll 1
    for (int s = s0; s < total; ++s) {
      if (s + RING - 1 < total) issue<PER>(s + RING - 1, rd == 0 ? RING - 1 : rd - 1);
      wait_dma_ahead<PER, RING - 3>(max(min(RING - 3, total - 3 - s), 0));     // slab s + 2 has landed
      if constexpr (!(DG & 8)) __builtin_amdgcn_s_barrier();
      rd = rd + 1 == RING ? 0 : rd + 1;
    }
  }
  template <int PRE0, class PRO>
  DEV void run(PRO&& pro) const {
    run_<PP_PER, PRE0>(pro);
  }
  DEV void run() const { run<RING - 1>([] {}); }

  // OGF kernels, first phase (loader waves 2 and 3 = waves 6 and 7 of the workgroup; waves 4 and 5 compute the halo tiles):
  // the n0 out-projection + GLU slabs and the first two slabs of the chain behind them -- HOLD = n0 + 2 slabs in all -- flow
  // through the ring; nothing beyond is requested, so that when step n0 - 1 ends every DMA has landed and the ring slots
  // that hold neither slab n0 nor n0 + 1 are free for the depthwise window (pp_block_kernel)
  DEV void run_og_phase() const {
    const int hold = n0 + 2;
    for (int i = 0; i < RING - 1; ++i) issue_pro(i, i);                        // n0 >= RING - 1
    wait_dma_ahead<PP_FR / 2, RING - 3>(RING - 3);                             // slabs 0 and 1 have landed
    __builtin_amdgcn_s_barrier();                                              // B0
    int wr = RING - 1;
#pragma unroll 1
    for (int s = 0; s < n0; ++s) {
      if (s + RING - 1 < hold) issue_pro(s + RING - 1, wr);
      wr = wr + 1 == RING ? 0 : wr + 1;
      wait_dma_ahead<PP_FR / 2, RING - 3>(max(min(RING - 3, hold - 3 - s), 0)); // slab s + 2 has landed
      __builtin_amdgcn_s_barrier();
    }
  }
  DEV void barriers_og_phase() const {       // loader waves 0 and 1 only take part in the barriers when they do not compute
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int s = 0; s < n0; ++s) __builtin_amdgcn_s_barrier();
  }
  // second phase (all four loader waves, after the prologue): catch up with the steady state of step n0, then run it
  DEV void run_after_og() const {
    const int hold = n0 + 2;
    for (int i = hold; i < min(n0 + RING - 1, total); ++i) issue<PP_PER>(i, i % RING);
    main_loop<PP_PER>(n0, n0 % RING);
  }
};

template <int RING, int DG>
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
    if constexpr (!(DG & 8)) __builtin_amdgcn_s_barrier();
    rd = rd + 1 == RING ? 0 : rd + 1;
    __builtin_amdgcn_sched_barrier(0);
  }
};

// y += W2 swish(W1aug [x ; 1]) over P hidden pairs (2 P ring slots): units A, AP, P - 2 x F, BP, B.  h0 / h1 and f0 / f1
// swap roles from unit to unit (accumulate <-> being prepared, operand in use <-> operand being built).
template <int P, int DG, class ST>
DEV void pp_chain(f32x4 (&y)[KB], const Split8 (&xf)[KS32X], PpPool& pl, ST& st, float k1, float ik2) {
  static_assert(P >= 3, "at least one full unit");
  f32x4 h0[2], h1[2];
  Split8 f0, f1;
  const f32x4 zero = splat4(0.f);
  h0[0] = zero; h0[1] = zero;
  pp_unit_A<DG>(h0, xf, pl, st);                                   // h(0)
  h1[0] = zero; h1[1] = zero;
  {
    PpPrep pc{h0[0], h0[1], f0, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
    pp_unit_AP<DG>(h1, xf, pc, pl, st);                            // h(1) ; hf(0)
  }
  constexpr int NF = P - 2;
#pragma unroll 1
  for (int i = 0; i < NF / 2; ++i) {
    h0[0] = zero; h0[1] = zero;
    {
      PpPrep pc{h1[0], h1[1], f1, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
      pp_unit_F<DG>(y, h0, xf, f0, pc, pl, st);                    // p even: B(p) with f0, A(p + 2) -> h0, h1 -> f1
    }
    h1[0] = zero; h1[1] = zero;
    {
      PpPrep pc{h0[0], h0[1], f0, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
      pp_unit_F<DG>(y, h1, xf, f1, pc, pl, st);                    // p odd: B(p) with f1, A(p + 2) -> h1, h0 -> f0
    }
  }
  if constexpr (NF & 1) {
    h0[0] = zero; h0[1] = zero;
    {
      PpPrep pc{h1[0], h1[1], f1, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
      pp_unit_F<DG>(y, h0, xf, f0, pc, pl, st);
    }
    {
      PpPrep pc{h0[0], h0[1], f0, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
      pp_unit_BP<DG>(y, f1, pc, pl, st);                           // B(P - 2) with f1 ; h(P - 1) -> f0
    }
    pp_unit_B<DG>(y, f0, pl, st);
  } else {
    {
      PpPrep pc{h1[0], h1[1], f1, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
      pp_unit_BP<DG>(y, f0, pc, pl, st);                           // B(P - 2) with f0 ; h(P - 1) -> f1
    }
    pp_unit_B<DG>(y, f1, pl, st);
  }
}

// largest magnitude of this lane's token row (the row is spread over the four lanes c, c + 16, c + 32, c + 48)
DEV float pp_row_max(const f32x4 (&xs)[KB]) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < KB; ++i)
    m = fmaxf(fmaxf(m, fmaxf(fabsf(xs[i].x), fabsf(xs[i].y))), fmaxf(fabsf(xs[i].z), fabsf(xs[i].w)));
  return group_max(m);
}
// scales of one chain y += W2 act(W1aug [x ; 1]) for this lane's token (ChainSc: the chain's host-side constants)
struct PpTok { float sx, k1, ik2, s2, inv2; };
DEV PpTok pp_chain_scales(const PpChainSc& c, float xmax) {
  PpTok t;
  t.sx = pp_pow2_scale(xmax);
  const float sh = pp_pow2_scale(fmaf(c.l1, xmax, c.bmax));      // |swish(h)| <= |h| <= l1 max|x| + max|b|
  const float u1 = c.sw1 * t.sx;                                  // unit of the hidden accumulators
  t.k1 = -1.4426950408889634f * pp_recip_pow2(u1);
  t.ik2 = u1 * pp_recip_pow2(sh);
  t.s2 = c.sw2 * sh;                                              // unit of the output accumulators
  t.inv2 = pp_recip_pow2(t.s2);
  return t;
}
// the five split operands of a 16-token tile for the W1 steps, in units of sx; k-slot 144 (the first padding slot of step 4)
// carries 1.0 (= sx in those units): row 144 of the streamed W1 holds the bias
DEV void split_operand(Split8 (&xf)[KS32X], const f32x4 (&xs)[KB], int g4, float sx) {
  const f32x4 s4 = splat4(sx);
#pragma unroll
  for (int t = 0; t < KS32X - 1; ++t) xf[t] = split8(xs[2 * t] * s4, xs[2 * t + 1] * s4);
  f32x4 oh = splat4(0.f);
  oh.x = g4 == 0 ? sx : 0.0f;
  xf[KS32X - 1] = split8(xs[KB - 1] * s4, oh);
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
template <int DG, class ST>
DEV void pp_tail_consume(const TailFf2Args& a, const PpTailLds& p, int g4, ST& st, PpPool& pl, f32x4 (&xs)[KB], f32x4 (&y)[KB]) {
  Split8 xf[KS32X];
  {
    const PpTok t = pp_chain_scales(a.pp_sc[0], pp_row_max(xs));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = (y[i] + lds4(p.pw2b, i, g4)) * splat4(t.s2);
    split_operand(xf, xs, g4, t.sx);
    pp_chain<9, DG>(y, xf, pl, st, t.k1, t.ik2);                                         // x3 = x2 + conv module
    pp_pool_land(pl);
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = y[i] * splat4(t.inv2);
  }
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = y[i];
  ln_lds(xs, p.lng, p.lnb, g4, a.eps);
  {
    const PpTok t = pp_chain_scales(a.pp_sc[1], pp_row_max(xs));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = (lds4(p.b2, i, g4) + splat4(inv_fc) * y[i]) * splat4(t.s2);   // x3 / fc + b2 (+ W2 h)
    split_operand(xf, xs, g4, t.sx);
    pp_chain<18, DG>(y, xf, pl, st, t.k1, t.ik2);
    pp_pool_land(pl);
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = splat4(a.fc * t.inv2) * y[i];
  }
  ln_lds(y, p.fg, p.fb, g4, a.eps);                                                  // block-final LayerNorm
}

// ff_module_1 + q / k / v projections of the 16 tokens in xs (x0 rows); stores x1 and qkv
template <int DG, class ST>
DEV void pp_ff1_consume(const Ff1QkvArgs& a, const PpFf1Lds& p, int g4, ST& st, PpPool& pl, f32x4 (&xs)[KB], int T) {
  f32x4 y[KB];
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = lds4(p.b2, kb, g4) + splat4(inv_fc) * xs[kb];
  ln_lds(xs, p.ln1g, p.ln1b, g4, a.eps);
  Split8 xf[KS32X];
  {
    const PpTok t = pp_chain_scales(a.pp_sc, pp_row_max(xs));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = y[i] * splat4(t.s2);
    split_operand(xf, xs, g4, t.sx);
    pp_chain<18, DG>(y, xf, pl, st, t.k1, t.ik2);
    pp_pool_land(pl);
#pragma unroll
    for (int i = 0; i < KB; ++i) { y[i] = splat4(a.fc * t.inv2) * y[i]; xs[i] = y[i]; }   // x1 = x0 + fc * (ffn + b2)
  }
  {
    const WaveCtx e = wave_ctx_fresh(a.M, T);
    if (e.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(a.x1 + e.row + 16 * i + e.g4, y[i]);
    }
  }
  if (a.xq_pe) {                                                                     // RBlock: the query is projected from x1 + PE
    const WaveCtx e = wave_ctx_fresh(a.M, T);
    const float* __restrict__ pe = a.xq_pe + (size_t)(e.tok % a.xq_U) * D + e.g4;
#pragma unroll
    for (int i = 0; i < KB; ++i) xs[i] += ldg4(pe + 16 * i);
  }
  ln_lds(xs, p.ln2g, p.ln2b, g4, a.eps);
  const float sx = pp_pow2_scale(pp_row_max(xs));
  split_operand(xf, xs, g4, sx);                                                     // the q / k / v bias rides in row 144 too
  const float invq = pp_recip_pow2(a.pp_sw_qkv * sx);
  const int nproj = a.xq_pe ? 1 : 3;                                                 // (the loaders stop behind the q group as well)
#pragma unroll 1
  for (int q = 0; q < nproj; ++q) {
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      pp_unit_S<DG>(acc, xf[t], pl, st);
    });
    pp_pool_land(pl);
    const float sc = (q == 0 ? a.qscale : 1.0f) * invq;
    const WaveCtx e = wave_ctx_fresh(a.M, T);
    if (e.live) {
      if (a.qkv_T > 0) {
        // head-major (AttnArgs::head_major): plane q of M D floats, row ((b H + h) T + t) of 36: features 16 i + g4 .. + 3 lie in
        // one head (36 and 16 i + g4 are multiples of four)
        const int bq = e.tok / a.qkv_T, tq = e.tok - bq * a.qkv_T;
        float* plane = a.qkv + (size_t)q * a.M * D + ((size_t)bq * a.qkv_H * a.qkv_T + tq) * 36;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int f0 = 16 * i + e.g4, hq = f0 / 36;
          stg4(plane + (size_t)hq * a.qkv_T * 36 + (f0 - 36 * hq), acc[i] * splat4(sc));
        }
      } else {
        float* qrow = a.qkv + (size_t)e.tok * (3 * D) + 16 * q * KB + e.g4;
#pragma unroll
        for (int i = 0; i < KB; ++i) stg4(qrow + 16 * i, acc[i] * splat4(sc));
      }
    }
  }
}

// ---- depthwise conv in the prologue (DWF) -------------------------------------------------------------------------------------
// conformer_blocks.py:205 (SeparableConv1D's depthwise half, 'same': 15 / 16 zeros) and chunk_conformer_blocks.py:262 ('causal':
// 31 zeros in front).  The round-2 path ran it as its own launch (dwconv_tile_kernel: 10 us, of which the work is ~2) and sent
// dw through HBM.  Here the workgroup's 64 frames + 31 halo rows of u and the 32 x 144 taps are staged in the LDS of ring slots
// 2..4 (the loaders fetch only slabs 0 and 1 until the prologue is over), 288 threads compute 8 frames x 4 channels each from
// register windows (the tile kernel's scheme), the results go back through the same LDS in token-major order and the consumer
// waves pick up their operand rows.  Tiling per utterance (tok_of) keeps the window inside one utterance: rows outside it are
// the conv's zero padding.
constexpr int DW_K = 32, DW_ROWS = 64 + DW_K - 1, DW_C4 = D / 4;
constexpr int DW_WORK = 6 * 64;                 // waves 0..5 stage and compute; waves 6, 7 have the slab DMAs in flight
DEV void pp_bar_lds() {                         // this wave's LDS operations are done; workgroup barrier; no fence (DMAs stay in flight)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// all eight waves call this; on return (consumer waves) xs = the depthwise conv output rows of the wave's 16 frames.
// WIN_READY (OGF kernels): the window rows were written by the waves that computed them (pp_og_tile); only the taps are staged
template <bool WIN_READY = false>
DEV void pp_dw_prologue(float* scratch, const TailFf2Args& a, f32x4 (&xs)[KB]) {
  float* win = scratch;                         // [DW_ROWS][D], later the conv output [64][D]
  float* wt = scratch + DW_ROWS * D;            // [DW_K][D]
  const int tid = threadIdx.x;
  const int T = a.dw_T, f0 = blockIdx.x * 64 - a.dw_pad;        // utterance frame of window row 0
  constexpr int NL = (DW_ROWS * DW_C4 + DW_WORK - 1) / DW_WORK, NW = (DW_K * DW_C4 + DW_WORK - 1) / DW_WORK;
  if (tid < DW_WORK) {
    f32x4 wstage[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) wstage[k] = ldg4(a.dw_wd + 4 * min(tid + k * DW_WORK, DW_K * DW_C4 - 1));
    if constexpr (!WIN_READY) {
      const float* __restrict__ ub = a.dw_u + (size_t)blockIdx.y * T * D;
      f32x4 stage[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int i = tid + k * DW_WORK, r = i / DW_C4, c4 = i - r * DW_C4, f = f0 + r;
        stage[k] = (i < DW_ROWS * DW_C4 && f >= 0 && f < T) ? ldg4(ub + (size_t)f * D + 4 * c4) : splat4(0.f);
      }
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int i = tid + k * DW_WORK;
        if (i < DW_ROWS * DW_C4) *reinterpret_cast<f32x4*>(&win[4 * i]) = stage[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int i = tid + k * DW_WORK;
      if (i < DW_K * DW_C4) *reinterpret_cast<f32x4*>(&wt[4 * i]) = wstage[k];
    }
  }
  pp_bar_lds();                                                       // P1: window + taps staged
  constexpr int TS = 8, GJ = 8;
  f32x4 acc[TS];
  const int c4 = tid % DW_C4, tg = tid / DW_C4;
  const bool worker = tid < (64 / TS) * DW_C4;                         // 288 threads: 8 frame groups x 36 channel groups
  if (worker) {
    const float* __restrict__ trow = &win[(tg * TS) * D + 4 * c4];
    const float* __restrict__ wrow = &wt[4 * c4];
#pragma unroll
    for (int i = 0; i < TS; ++i) acc[i] = splat4(0.f);
#pragma unroll 1
    for (int j0 = 0; j0 < DW_K; j0 += GJ) {
      f32x4 w8[TS + GJ - 1], tp[GJ];
#pragma unroll
      for (int r = 0; r < TS + GJ - 1; ++r) w8[r] = *reinterpret_cast<const f32x4*>(trow + (j0 + r) * D);
#pragma unroll
      for (int j = 0; j < GJ; ++j) tp[j] = *reinterpret_cast<const f32x4*>(wrow + (j0 + j) * D);
#pragma unroll
      for (int j = 0; j < GJ; ++j)
#pragma unroll
        for (int i = 0; i < TS; ++i) acc[i] += w8[i + j] * tp[j];
    }
  }
  pp_bar_lds();                                                       // P2: every window read is done
  if (worker) {
#pragma unroll
    for (int i = 0; i < TS; ++i) *reinterpret_cast<f32x4*>(&win[(tg * TS + i) * D + 4 * c4]) = acc[i];
  }
  pp_bar_lds();                                                       // P3: conv output in LDS, token-major
  if (tid < BLOCK_THREADS) {
    const int lane = tid & 63, row = (tid >> 6) * 16 + (lane & 15), g4 = (lane >> 4) * 4;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = *reinterpret_cast<const f32x4*>(&win[row * D + 16 * kb + g4]);
  }
  pp_bar_lds();                                                       // P4: the scratch is free: the loaders fill the ring
}

// ---- out-projection + residual + LayerNorm + pw_conv_1 + GLU in the prologue (OGF, round 4) ------------------------------------
// conformer_blocks.py:164-170 (x2 = x1 + attention output), :209-213 (u = GLU(pw_conv_1(LN(x2)))).  pp_out_glu_kernel ran this
// as a launch of its own -- 16 us for 2 us of matrix work, x2 and u through HBM.  Here the six waves 0..5 of the workgroup
// each take ONE 16-frame tile of the 96 frames the depthwise window needs -- waves 0..3 the workgroup's own 64 frames (their
// x2 rows stay in registers as the residual of the conv tail), waves 4 and 5 the halo: the 16 frames in front and the 16
// behind for Keras 'same' padding (15 / 16), the 32 in front for 'causal' (31 / 0) -- and walk the fifteen plain ring slots of
// the out_glu stream (five of the out projection, five of pw_conv_1's value tiles, five of its gate tiles), which precede
// the conv tail's slots in the same ring (PpLoader::run_og_phase).  The halo is recomputed by both neighbours: 50 % more of a
// layer that is 6 % of the block's matrix work, against a launch, its fixed ~11 us, and 37 MB of x2 / u traffic per block.
// The u rows go straight into the depthwise window in LDS; rows outside the utterance are the conv's zero padding.
struct PpOgLds { float lng[D], lnb[D]; };
DEV int pp_og_tile_frame0(int wv, int pad) {      // first utterance frame of wave wv's tile (wave-uniform; may be negative)
  const int own0 = blockIdx.x * 64;
  if (wv < WAVES_PER_BLOCK) return own0 + 16 * wv;
  const int nfront = (pad + 15) / 16;             // 'same' (15): one tile in front, one behind; 'causal' (31): two in front
  const int h = wv - WAVES_PER_BLOCK;             // 0 or 1
  return h < nfront ? own0 - 16 * (nfront - h) : own0 + 64 + 16 * (h - nfront);
}
template <int DG, class ST>
DEV void pp_og_tile(const OutGluArgs& g, const PpOgLds& p, ST& st, PpPool& pl, f32x4 (&xs)[KB], f32x4 (&x2)[KB], int lane,
                    int frame0, int T, int f0, float* win) {
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  Split8 xf[KS32X];
  {
    const float sx = pp_pow2_scale(pp_row_max(xs));
    split_operand(xf, xs, g4, sx);
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      pp_unit_S<DG>(acc, xf[t], pl, st);
    });
    pp_pool_land(pl);
    const f32x4 inv = splat4(pp_recip_pow2(g.pp_sw_out * sx));
#pragma unroll
    for (int i = 0; i < KB; ++i) { x2[i] += acc[i] * inv; xs[i] = x2[i]; }                // x2 = x1 + attention (+ bias: row 144)
  }
  ln_lds(xs, p.lng, p.lnb, g4, g.eps);
  const float sx = pp_pow2_scale(pp_row_max(xs));
  split_operand(xf, xs, g4, sx);
  f32x4 val[KB], gate[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) { val[i] = splat4(0.f); gate[i] = splat4(0.f); }
  static_for<0, KS32X>([&](auto T_) {
    constexpr int t = decltype(T_)::value;
    pp_unit_S<DG>(val, xf[t], pl, st);
  });
  static_for<0, KS32X>([&](auto T_) {
    constexpr int t = decltype(T_)::value;
    pp_unit_S<DG>(gate, xf[t], pl, st);
  });
  pp_pool_land(pl);                                       // the reads into the slot behind the stream: landed, then forgotten
  const float inv = pp_recip_pow2(g.pp_sw_pw1 * sx);
  const int f = frame0 + c, r = f - f0;                   // this lane's frame and its window row
  const bool inside = f >= 0 && f < T;
  if (r >= 0 && r < DW_ROWS) {
    float* wrow = win + r * D + g4;
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const f32x4 va = val[i] * splat4(inv), vb = gate[i] * splat4(inv);
      f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
      *reinterpret_cast<f32x4*>(wrow + 16 * i) = inside ? o : splat4(0.f);
    }
  }
}

// CTC class head on the 16 tokens in xs (pp_head_kernel's loop, behind a block's tail): a.head_* (launch.h)
template <int DG, class ST>
DEV void pp_head_consume(const TailFf2Args& a, int g4, int lane, ST& st, PpPool& pl, const f32x4 (&xs)[KB], int T) {
  Split8 xf[KS32X];
  const float sx = pp_pow2_scale(pp_row_max(xs));
  split_operand(xf, xs, g4, sx);
  const float inv = pp_recip_pow2(a.head_sw * sx);
  float best_v = -INFINITY;
  int best_i = 0;
  const bool want_max = a.head_argmax != nullptr || a.head_maxval != nullptr;
#pragma unroll 1
  for (int g = 0; g < a.head_groups; ++g) {
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto Tt) {
      constexpr int t = decltype(Tt)::value;
      pp_unit_S<DG>(acc, xf[t], pl, st);
    });
    pp_pool_land(pl);
    const WaveCtx e = wave_ctx_fresh(a.M, T);
    float* yrow = a.head_y ? a.head_y + (size_t)e.tok * a.head_ldy : nullptr;
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int tile = KB * g + i, f0 = 16 * tile + e.g4;
      if (16 * tile < a.head_nvalid) {           // wave-uniform: the last group is padded with zero columns
        const f32x4 v = acc[i] * splat4(inv);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if (want_max) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (f0 + j < a.head_nvalid && vv[j] > best_v) { best_v = vv[j]; best_i = f0 + j; }
        }
        if (yrow && e.live) {
          if (f0 + 3 < a.head_nvalid && (a.head_ldy & 3) == 0) stg4(yrow + f0, v);
          else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (f0 + j < a.head_nvalid) yrow[f0 + j] = vv[j];
        }
      }
    }
  }
  if (!want_max) return;
  // the four lane groups of a token hold disjoint classes: max over the groups, lowest class on ties
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float ov = __shfl_xor(best_v, off);
    const int oi = __shfl_xor(best_i, off);
    if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
  }
  const WaveCtx e = wave_ctx_fresh(a.M, T);
  if (a.head_argmax && e.live && lane < 16) a.head_argmax[e.tok] = best_i;
  if (a.head_maxval && e.live && lane < 16) a.head_maxval[e.tok] = best_v;
}

// TAIL: conv tail + ff_module_2 + LayerNorm of one block (a);  FF1: ff_module_1 + qkv of a block (b) -- of the NEXT block
// when both are set (the block output stays in registers; a.y may be null then).  DWF: the depthwise conv runs in the
// prologue (a.dw_u / dw_wd / dw_T / dw_pad) and the grid is (ceil(T / 64), utterances).  OGF (needs DWF): the window of the
// depthwise conv is computed in the prologue too, from the attention output and x1 (g; a.dw_u and a.x2 are not read).
// PRE (FF1 without TAIL): the plain layer in front of the block -- the subsampling Dense, the CTC decoder's projection -- runs
// first, x0 = pre_x W + b from b.pre_x [M, 144 * pre_chunks] and the stream b.pre_pp (pp_sublinear_kernel's loop); b.x0 is not read.
// HEAD (TAIL + OGF without FF1): the CTC class head runs behind the block (a.head_*: pp_head_kernel's loop on the block output).
template <bool TAIL, bool FF1, int DG = 0, bool DWF = false, bool OGF = false, bool PRE = false, bool HEAD = false>
__global__ __launch_bounds__(LD_THREADS) void pp_block_kernel(TailFf2Args a, Ff1QkvArgs b, OutGluArgs g) {
  static_assert(!HEAD || (TAIL && OGF && !FF1), "the class head follows the last block's tail");
  static_assert(TAIL || !DWF, "the depthwise conv feeds the conv tail");
  static_assert(DWF || !OGF, "the out-projection + GLU prologue feeds the depthwise window");
  static_assert(!PRE || (FF1 && !TAIL), "the layer in front feeds ff_module_1");
  __shared__ __attribute__((aligned(16))) u32x4_t ring[PP_RING * PP_SLB];
  __shared__ __attribute__((aligned(16))) PpTailLds pt;
  __shared__ __attribute__((aligned(16))) PpFf1Lds pf;
  __shared__ __attribute__((aligned(16))) PpOgLds pg;
  static_assert((PP_RING - 2) * PP_SLB * 16 >= (DW_ROWS + DW_K) * D * 4, "the prologue scratch is ring slots 2..");
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int NOG = OGF ? 3 * KS32X : 0;
  constexpr int N1 = NOG + (TAIL ? PP_TAIL_SLABS : PP_FF1_SLABS), TOTAL = NOG + (TAIL ? PP_TAIL_SLABS : 0) + (FF1 ? PP_FF1_SLABS : 0);
  // OGF: when the out_glu stream is through, slabs NOG and NOG + 1 sit in ring slots NOG % RING and (NOG + 1) % RING and
  // every other slot is free: the window + taps take the contiguous run of slots behind them
  constexpr int OG_SCR = (NOG + 2) % PP_RING;
  static_assert(!OGF || (OG_SCR + (PP_RING - 3) <= PP_RING && (PP_RING - 3) * PP_SLB * 16 >= (DW_ROWS + DW_K) * D * 4),
                "the slots behind slab NOG + 1 up to the end of the ring hold the depthwise window and the taps");
  float* scratch = reinterpret_cast<float*>(ring + (OGF ? OG_SCR : 2) * PP_SLB);
  f32x4 xs[KB], y[KB];
  const u32x4_t* s0 = reinterpret_cast<const u32x4_t*>(PRE ? b.pre_pp : g.pp_slabs);
  const u32x4_t* s1 = reinterpret_cast<const u32x4_t*>(TAIL ? a.pp_slabs : b.pp_slabs);
  const u32x4_t* s2 = reinterpret_cast<const u32x4_t*>(HEAD ? a.head_pp : b.pp_slabs);
  const int total_rt = TOTAL + (HEAD ? KS32X * a.head_groups : 0) - ((FF1 && b.xq_pe) ? 2 * KS32X : 0);      // RBlock: no k / v groups
  if (wv >= WAVES_PER_BLOCK + (OGF ? 2 : 0)) {
    if constexpr (PRE) {
      const int n0 = KS32X * b.pre_chunks;
      PpLoader<PP_RING, DG>{ring, s1, s2, n0 + N1, n0 + TOTAL, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63), s0, n0}.run();
      return;
    }
    const PpLoader<PP_RING, DG> ld{ring, s1, s2, N1, total_rt, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63), s0, NOG};
    if constexpr (OGF) {
      ld.run_og_phase();
      pp_dw_prologue<true>(scratch, a, xs);
      ld.run_after_og();
    } else if constexpr (DWF) ld.template run<2>([&] { pp_dw_prologue(scratch, a, xs); });
    else ld.run();
    return;
  }
  const int M = TAIL ? a.M : b.M, TT = DWF ? a.dw_T : 0;
  const WaveCtx c = wave_ctx(M, TT);
  PpReader<PP_RING, DG> st{ring, c.lane};
  if constexpr (OGF) {
    // waves 0..5: one tile of the window each.  The rows of ctx / x1 (frames outside the utterance: the nearest real frame;
    // their u is replaced by zeros)
    const int frame0 = pp_og_tile_frame0(wv, a.dw_pad), T = a.dw_T, f0 = blockIdx.x * 64 - a.dw_pad;
    const size_t row = ((size_t)blockIdx.y * T + (size_t)min(max(frame0 + (c.lane & 15), 0), T - 1)) * D;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(g.ctx + row + 16 * kb + c.g4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(g.x1 + row + 16 * kb + c.g4);
    {
      const auto r0 = stash_load<D>(g.cv_ln_g), r1 = stash_load<D>(g.cv_ln_b);
      stash_store<D>(pg.lng, r0); stash_store<D>(pg.lnb, r1);
    }
    pp_tail_stash(pt, a);
    if constexpr (FF1) pp_ff1_stash(pf, b);
    st.sync();                                           // B0
    PpPool pl;
    pp_prime<DG>(pl, st);
    pp_pool_land(pl);
    pp_og_tile<DG>(g, pg, st, pl, xs, y, c.lane, frame0, T, f0, scratch);      // y = x2 rows of this wave's tile
    pp_dw_prologue<true>(scratch, a, xs);                // waves 0..3: xs = depthwise output rows of the wave's own frames
    if (wv >= WAVES_PER_BLOCK) {                         // the halo waves join the loaders
      const PpLoader<PP_RING, DG> ld{ring, s1, s2, N1, total_rt, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63), s0, NOG};
      ld.run_after_og();
      return;
    }
    pp_prime<DG>(pl, st);
    pp_pool_land(pl);
    pp_tail_consume<DG>(a, pt, c.g4, st, pl, xs, y);
    if (a.y) {
      const WaveCtx e = wave_ctx_fresh(M, TT);
      if (e.live) {
#pragma unroll
        for (int i = 0; i < KB; ++i) stg4(a.y + e.row + 16 * i + e.g4, y[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < KB; ++i) xs[i] = y[i];
    if constexpr (FF1) pp_ff1_consume<DG>(b, pf, c.g4, st, pl, xs, TT);
    if constexpr (HEAD) pp_head_consume<DG>(a, c.g4, c.lane, st, pl, xs, TT);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return;
  }
  if constexpr (TAIL) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + c.row + 16 * kb + c.g4);     // residuals ride in the accumulators
    if constexpr (!DWF) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
    }
    pp_tail_stash(pt, a);
  } else if constexpr (!PRE) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(b.x0 + c.row + 16 * kb + c.g4);
  }
  const float* __restrict__ prow = PRE ? b.pre_x + (size_t)c.tok * (D * b.pre_chunks) + c.g4 : nullptr;
  if constexpr (PRE) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) { xs[kb] = ldg4(prow + 16 * kb); y[kb] = splat4(0.f); }
  }
  if constexpr (FF1) pp_ff1_stash(pf, b);
  if constexpr (DWF) pp_dw_prologue(scratch, a, xs);     // after the parameter loads: their latency hides under the prologue
  st.sync();
  PpPool pl;
  pp_prime<DG>(pl, st);
  pp_pool_land(pl);
  if constexpr (PRE) {
    // every 144-wide chunk of the row under its own power-of-two scale (pp_sublinear_kernel); the next chunk's rows are
    // requested before this chunk's units
#pragma unroll 1
    for (int f = 0; f < b.pre_chunks; ++f) {
      Split8 xf[KS32X];
      const float sx = pp_pow2_scale(pp_row_max(xs));
      split_operand(xf, xs, c.g4, sx);
      if (f + 1 < b.pre_chunks) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(prow + (size_t)(f + 1) * D + 16 * kb);
      }
      f32x4 acc[KB];
#pragma unroll
      for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
      static_for<0, KS32X>([&](auto Tt) {
        constexpr int t = decltype(Tt)::value;
        pp_unit_S<DG>(acc, xf[t], pl, st);
      });
      pp_pool_land(pl);
      const f32x4 inv = splat4(pp_recip_pow2(b.pre_sw * sx));
#pragma unroll
      for (int i = 0; i < KB; ++i) y[i] += acc[i] * inv;
    }
#pragma unroll
    for (int i = 0; i < KB; ++i) xs[i] = y[i];
  }
  if constexpr (TAIL) {
    pp_tail_consume<DG>(a, pt, c.g4, st, pl, xs, y);
    if (a.y) {
      const WaveCtx e = wave_ctx_fresh(M, TT);
      if (e.live) {
#pragma unroll
        for (int i = 0; i < KB; ++i) stg4(a.y + e.row + 16 * i + e.g4, y[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < KB; ++i) xs[i] = y[i];
  }
  if constexpr (FF1) pp_ff1_consume<DG>(b, pf, c.g4, st, pl, xs, TT);
  // the pool still holds nine reads of the slot after the last one: drain them before the wave ends
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}


// attention out-projection + residual, conv-module LayerNorm, pw_conv_1 + GLU (conformer_blocks.py:164-170, :209-213) on the
// same stream machinery: 15 plain ring slots (five of the out projection, five of pw_conv_1's value tiles, five of its gate
// tiles; the three biases in row 144), 405 MFMAs per wave where the three-term out_glu_ld_kernel (fused.hip) ran 810
__global__ __launch_bounds__(LD_THREADS) void pp_out_glu_kernel(OutGluArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[PP_RING * PP_SLB];
  __shared__ __attribute__((aligned(16))) PpOgLds p;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int TOTAL = 3 * KS32X;
  if (wv >= WAVES_PER_BLOCK) {
    const u32x4_t* s1 = reinterpret_cast<const u32x4_t*>(a.pp_slabs);
    PpLoader<PP_RING, 0>{ring, s1, s1, TOTAL, TOTAL, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  constexpr int DG = 0;
  const WaveCtx c = wave_ctx(a.M);
  PpReader<PP_RING, 0> st{ring, c.lane};
  f32x4 xs[KB], x2[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.ctx + c.row + 16 * kb + c.g4);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) x2[kb] = ldg4(a.x1 + c.row + 16 * kb + c.g4);
  {
    const auto r0 = stash_load<D>(a.cv_ln_g), r1 = stash_load<D>(a.cv_ln_b);
    stash_store<D>(p.lng, r0); stash_store<D>(p.lnb, r1);
  }
  st.sync();
  PpPool pl;
  pp_prime<DG>(pl, st);
  pp_pool_land(pl);
  Split8 xf[KS32X];
  {
    const float sx = pp_pow2_scale(pp_row_max(xs));
    split_operand(xf, xs, c.g4, sx);
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      pp_unit_S<DG>(acc, xf[t], pl, st);
    });
    pp_pool_land(pl);
    const f32x4 inv = splat4(pp_recip_pow2(a.pp_sw_out * sx));
#pragma unroll
    for (int i = 0; i < KB; ++i) { x2[i] += acc[i] * inv; xs[i] = x2[i]; }                // x2 = x1 + attention (+ bias: row 144)
  }
  {
    const WaveCtx e = wave_ctx_fresh(a.M);
    if (e.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(a.x2 + e.row + 16 * i + e.g4, x2[i]);
    }
  }
  ln_lds(xs, p.lng, p.lnb, c.g4, a.eps);
  const float sx = pp_pow2_scale(pp_row_max(xs));
  split_operand(xf, xs, c.g4, sx);
  f32x4 val[KB], gate[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) { val[i] = splat4(0.f); gate[i] = splat4(0.f); }
  static_for<0, KS32X>([&](auto T) {
    constexpr int t = decltype(T)::value;
    pp_unit_S<DG>(val, xf[t], pl, st);
  });
  static_for<0, KS32X>([&](auto T) {
    constexpr int t = decltype(T)::value;
    pp_unit_S<DG>(gate, xf[t], pl, st);
  });
  pp_pool_land(pl);                                       // the pool's reads past the last slot
  const float inv = pp_recip_pow2(a.pp_sw_pw1 * sx);
  const WaveCtx e = wave_ctx_fresh(a.M);
  if (e.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const f32x4 va = val[i] * splat4(inv), vb = gate[i] * splat4(inv);
      f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
      stg4(a.u + e.row + 16 * i + e.g4, o);
    }
  }
}


// CTC class head of dmodel 144: logits = x W + b over `groups` column groups of nine tiles (five plain ring slots each, the
// bias in row 144), per-frame arg-max (first maximum wins) and / or the logits themselves -- head_ld_kernel's contract
// (fused.hip) on the two-term stream: 135 MFMAs per group and wave instead of 270, nothing to fetch for the bias
// Round 6: blockIdx.y splits the column groups (gper per workgroup) when the rows alone leave most of the chip idle -- the Translator's
// 144 -> 9160 head over 5 952 rows is 93 row workgroups; the per-range (maximum, class) pairs then go to part_v / part_i [ranges][M]
// and head_combine_kernel picks the winner (lowest class among equal maxima, as within a range).
__global__ __launch_bounds__(LD_THREADS) void pp_head_kernel(GemmArgs a, const u32x4_t* __restrict__ pp_all, float pp_sw, int groups_all, int gper,
                                                               float* part_v, int32_t* part_i) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[PP_RING * PP_SLB];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g_first = blockIdx.y * gper, groups = min(groups_all, g_first + gper) - g_first;
  const u32x4_t* __restrict__ pp = pp_all + (size_t)g_first * KS32X * PP_SLB;
  if (wv >= WAVES_PER_BLOCK) {
    PpLoader<PP_RING, 0>{ring, pp, pp, KS32X * groups, KS32X * groups, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  constexpr int DG = 0;
  const WaveCtx c = wave_ctx(a.M);
  PpReader<PP_RING, 0> st{ring, c.lane};
  Split8 xf[KS32X];
  float inv;
  {
    f32x4 xs[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x + c.row + 16 * kb + c.g4);
    const float sx = pp_pow2_scale(pp_row_max(xs));
    split_operand(xf, xs, c.g4, sx);
    inv = pp_recip_pow2(pp_sw * sx);
  }
  st.sync();
  PpPool pl;
  pp_prime<DG>(pl, st);
  pp_pool_land(pl);
  float best_v = -INFINITY;
  int best_i = 0;
  const bool want_max = a.argmax_out != nullptr || a.maxval_out != nullptr;
#pragma unroll 1
  for (int g = 0; g < groups; ++g) {
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      pp_unit_S<DG>(acc, xf[t], pl, st);
    });
    pp_pool_land(pl);
    const WaveCtx e = wave_ctx_fresh(a.M);
    float* yrow = a.y ? a.y + (size_t)e.tok * a.ldy : nullptr;
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int tile = KB * (g_first + g) + i, f0 = 16 * tile + e.g4;
      if (16 * tile < a.n_valid) {               // wave-uniform: the last group is padded with zero columns
        const f32x4 v = acc[i] * splat4(inv);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if (want_max) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (f0 + j < a.n_valid && vv[j] > best_v) { best_v = vv[j]; best_i = f0 + j; }
        }
        if (yrow && e.live) {
          if (f0 + 3 < a.n_valid && (a.ldy & 3) == 0) stg4(yrow + f0, v);
          else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (f0 + j < a.n_valid) yrow[f0 + j] = vv[j];
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the pool's reads past the last slot
  if (!want_max) return;
  // the four lane groups of a token hold disjoint classes: max over the groups, lowest class on ties
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float ov = __shfl_xor(best_v, off);
    const int oi = __shfl_xor(best_i, off);
    if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
  }
  const WaveCtx e = wave_ctx_fresh(a.M);
  if (part_v) {                                  // one of several class ranges: the pair goes to the combine kernel
    if (e.live && c.lane < 16) { part_v[(size_t)blockIdx.y * a.M + e.tok] = best_v; part_i[(size_t)blockIdx.y * a.M + e.tok] = best_i; }
    return;
  }
  if (a.argmax_out && e.live && c.lane < 16) a.argmax_out[e.tok] = best_i;
  if (a.maxval_out && e.live && c.lane < 16) a.maxval_out[e.tok] = best_v;
}
__global__ __launch_bounds__(256) void head_combine_kernel(const float* __restrict__ part_v, const int32_t* __restrict__ part_i, int ranges, int M,
                                                            int32_t* __restrict__ argmax_out, float* __restrict__ maxval_out) {
  const int tok = blockIdx.x * 256 + threadIdx.x;
  if (tok >= M) return;
  float bv = part_v[tok];
  int bi = part_i[tok];
  for (int r = 1; r < ranges; ++r) {
    const float v = part_v[(size_t)r * M + tok];
    const int i = part_i[(size_t)r * M + tok];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  if (argmax_out) argmax_out[tok] = bi;
  if (maxval_out) maxval_out[tok] = bv;
}


// Subsampling Dense (conformer_blocks.py:93-96): y[t, :] = relu(conv2)[t, f2, ch] W[f2 * 144 + ch, :] + b, K = F2 * 144 = 2880,
// as F2 chunks of one plain group each (five ring slots per chunk; the bias in row 144 of chunk 0, zeros in the others).  The
// round-2 kernel (sublinear_split_ld_kernel) runs three bf16 terms = 4 860 MFMAs per wave and streams 6 bytes per weight; the
// operand here has no static bound (ReLU outputs), so every 144-wide chunk of a token's row gets ITS OWN power-of-two scale
// from its own maximum -- per-chunk accumulators, multiplied out of their unit into y as each chunk ends (exact) --: 2 430
// MFMAs, 4 bytes per weight.  The rows of chunk f + 1 are requested before the units of chunk f.
__global__ __launch_bounds__(LD_THREADS) void pp_sublinear_kernel(StreamGemmArgs a, const u32x4_t* __restrict__ pp, float pp_sw, int chunks) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[PP_RING * PP_SLB];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= WAVES_PER_BLOCK) {
    PpLoader<PP_RING, 0>{ring, pp, pp, KS32X * chunks, KS32X * chunks, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  constexpr int DG = 0;
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4;
  const int tok = (blockIdx.x * WAVES_PER_BLOCK + wv) * 16 + (lane & 15);
  const bool live = tok < a.M;
  const float* __restrict__ xrow = a.x + (size_t)min(tok, a.M - 1) * a.K + g4;
  PpReader<PP_RING, 0> st{ring, lane};
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) { xs[kb] = ldg4(xrow + 16 * kb); y[kb] = splat4(0.f); }
  st.sync();
  PpPool pl;
  pp_prime<DG>(pl, st);
  pp_pool_land(pl);
#pragma unroll 1
  for (int f = 0; f < chunks; ++f) {
    Split8 xf[KS32X];
    const float sx = pp_pow2_scale(pp_row_max(xs));
    split_operand(xf, xs, g4, sx);
    if (f + 1 < chunks) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(xrow + (size_t)(f + 1) * D + 16 * kb);
    }
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = splat4(0.f);
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      pp_unit_S<DG>(acc, xf[t], pl, st);
    });
    pp_pool_land(pl);
    const f32x4 inv = splat4(pp_recip_pow2(pp_sw * sx));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] += acc[i] * inv;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the pool's reads past the last slot
  if (live) {
    float* yrow = a.y + (size_t)tok * a.ldy + g4;
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(yrow + 16 * i, y[i]);
  }
}

}  // namespace

// the depthwise conv rides in the tail kernel's prologue when the caller asked for it (dw_u) and the per-utterance tiling wastes
// little: ceil(T / 64) chunks of 64 frames per utterance (T = 250: 2.4 % idle rows; T = 100 would idle 22 %)
bool pp_dw_fold_ok(int T, int ksz) {
  // MI355ASR_PP_DW=0: depthwise conv as its own launch (dwconv_tile_kernel)
  static const bool on = mi355_env("MI355ASR_PP_DW", 1) != 0;
  return on && pp_enabled() && ksz == DW_K && T >= 64 && 64 * ((T + 63) / 64) * 10 <= 11 * T;
}
bool pp_enabled();
static bool pp_dw_fold(const TailFf2Args& a) { return a.dw_u && a.dw_wd && a.dw_T > 0 && a.M % a.dw_T == 0; }
bool pp_enabled() {
  // MI355ASR_PP=0: the round-2 chunk-wise ring kernels (fused.hip) instead of the pair-pipelined ones
  static const bool on = mi355_env("MI355ASR_PP", 1) != 0;
  return on;
}
int launch_pp_head(const GemmArgs& a, const float* pp, float pp_sw, int groups, hipStream_t s) {
  // MI355ASR_PP_HEAD=0: the three-term head_ld_kernel (fused.hip)
  static const bool on = mi355_env("MI355ASR_PP_HEAD", 1) != 0;
  static const bool ring_on = mi355_env("MI355ASR_HEAD_RING", 1) != 0;
  if (!on || !ring_on || !pp_enabled() || !pp || groups < 1 || a.n_valid > 144 * groups || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL(pp_head_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, reinterpret_cast<const u32x4_t*>(pp), pp_sw, groups, groups,
                     (float*)nullptr, (int32_t*)nullptr);
  return 0;
}
// the same head with the column groups split over `ranges` workgroups per row tile; scratch = 2 * ranges * M words (maxima, then classes)
int pp_head_ranges(int M, int groups) {
  static const int ncu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
  const int wgs = (M + 63) / 64;
  if (wgs < 1) return 1;
  // rounds over the chip x (groups a workgroup walks + its prologue, ~one group's worth): 5 952 rows x 64 groups = 93 workgroups:
  // 8 ranges (3 rounds of 8 groups) beat 2 (one round of 32); 12 000 rows = 188 workgroups: 4 ranges (3 rounds of 16) beat 1 (64).
  // At least four groups per range, at most 8 ranges (the scratch of the callers); MI355ASR_PP_HEAD_RANGES=n forces n.
  static const int force = (int)mi355_env("MI355ASR_PP_HEAD_RANGES", 0);
  int best = 1;
  double best_c = 1e30;
  for (int nr = 1; nr <= 8; ++nr) {
    const int gper = (groups + nr - 1) / nr;
    if (nr > 1 && gper < 4) break;
    if ((groups + gper - 1) / gper != nr) continue;
    if (force && nr != force) continue;
    const double c = (double)(((long)wgs * nr + ncu - 1) / ncu) * (gper + 1.0) + (nr > 1 ? 0.5 : 0.0);
    if (c < best_c * 0.97) { best_c = c; best = nr; }
  }
  return best;
}
int launch_pp_head_split(const GemmArgs& a, const float* pp, float pp_sw, int groups, int ranges, float* scratch, hipStream_t s) {
  const bool wants = a.argmax_out != nullptr || a.maxval_out != nullptr;
  if (ranges <= 1 || (wants && !scratch)) return launch_pp_head(a, pp, pp_sw, groups, s);   // (logits only: the ranges need no combine)
  static const bool on = mi355_env("MI355ASR_PP_HEAD", 1) != 0;
  static const bool ring_on = mi355_env("MI355ASR_HEAD_RING", 1) != 0;
  if (!on || !ring_on || !pp_enabled() || !pp || groups < 1 || a.n_valid > 144 * groups || a.M <= 0) return -1;
  const int gper = (groups + ranges - 1) / ranges, nr = (groups + gper - 1) / gper;
  const int tiles = (a.M + 15) / 16;
  note_scheme(SCHEME_F16X2);
  const bool want_max = wants;
  float* pv = want_max ? scratch : nullptr;
  int32_t* pi = want_max ? reinterpret_cast<int32_t*>(scratch + (size_t)nr * a.M) : nullptr;
  hipLaunchKernelGGL(pp_head_kernel, dim3((tiles + 3) / 4, nr), dim3(LD_THREADS), 0, s, a, reinterpret_cast<const u32x4_t*>(pp), pp_sw, groups, gper, pv, pi);
  if (want_max) hipLaunchKernelGGL(head_combine_kernel, dim3((a.M + 255) / 256), dim3(256), 0, s, pv, pi, nr, a.M, a.argmax_out, a.maxval_out);
  return 0;
}
int launch_head_combine(const float* part_v, const int32_t* part_i, int ranges, int M, int32_t* argmax_out, float* maxval_out, hipStream_t s) {
  hipLaunchKernelGGL(head_combine_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part_v, part_i, ranges, M, argmax_out, maxval_out);
  return 0;
}
bool pp_sublinear_ok(const StreamGemmArgs& a, const float* pp) {
  // MI355ASR_PP_SUBLINEAR=0: the three-term sublinear_split_ld_kernel (fused.hip)
  static const bool on = mi355_env("MI355ASR_PP_SUBLINEAR", 1) != 0;
  return on && pp_enabled() && pp && a.NT == KB && a.K % D == 0 && a.K >= D && a.M > 0 && (a.ldy & 3) == 0;
}
int launch_pp_sublinear(const StreamGemmArgs& a, const float* pp, float pp_sw, hipStream_t s) {
  if (!pp_sublinear_ok(a, pp)) return -1;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL(pp_sublinear_kernel, dim3((a.M + 63) / 64), dim3(LD_THREADS), 0, s, a, reinterpret_cast<const u32x4_t*>(pp), pp_sw, a.K / D);
  return 0;
}
int launch_pp_out_glu(const OutGluArgs& a, hipStream_t s) {
  // MI355ASR_PP_OUTGLU=0: the three-term out_glu_ld_kernel (fused.hip)
  static const bool on = mi355_env("MI355ASR_PP_OUTGLU", 1) != 0;
  if (!on || !pp_enabled() || !a.pp_slabs || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL(pp_out_glu_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a);
  return 0;
}
int launch_pp_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, hipStream_t s) {
  if (!pp_enabled() || !a.pp_slabs || !b.pp_slabs || a.M != b.M || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
  note_scheme(SCHEME_F16X2);
#ifdef MI355ASR_DIAG_KERNELS
  static const int dg = (int)mi355_env("MI355ASR_PP_DIAG", 0);
  if (dg) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "MI355ASR_PP_DIAG=%d: timing-only kernel variant, results are WRONG\n", dg); warned = true; }
    const dim3 g((tiles + 3) / 4), t(LD_THREADS);
#define PP_DIAG_CASE(N) case N: hipLaunchKernelGGL((pp_block_kernel<true, true, N>), g, t, 0, s, a, b, OutGluArgs{}); return 0;
    switch (dg) {
      PP_DIAG_CASE(1) PP_DIAG_CASE(2) PP_DIAG_CASE(3) PP_DIAG_CASE(4) PP_DIAG_CASE(8) PP_DIAG_CASE(16) PP_DIAG_CASE(24)
      PP_DIAG_CASE(26) PP_DIAG_CASE(27) PP_DIAG_CASE(25) PP_DIAG_CASE(7) PP_DIAG_CASE(10) PP_DIAG_CASE(18) PP_DIAG_CASE(64) PP_DIAG_CASE(65)
      default: break;
    }
#undef PP_DIAG_CASE
  }
#endif
  if (pp_dw_fold(a)) {
    hipLaunchKernelGGL((pp_block_kernel<true, true, 0, true>), dim3((a.dw_T + 63) / 64, a.M / a.dw_T), dim3(LD_THREADS), 0, s, a, b, OutGluArgs{});
    return 0;
  }
  hipLaunchKernelGGL((pp_block_kernel<true, true>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, b, OutGluArgs{});
  return 0;
}
int launch_pp_tail_ff2(const TailFf2Args& a, hipStream_t s) {
  if (!pp_enabled() || !a.pp_slabs || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
  note_scheme(SCHEME_F16X2);
  if (pp_dw_fold(a)) {
    hipLaunchKernelGGL((pp_block_kernel<true, false, 0, true>), dim3((a.dw_T + 63) / 64, a.M / a.dw_T), dim3(LD_THREADS), 0, s, a, Ff1QkvArgs{}, OutGluArgs{});
    return 0;
  }
  hipLaunchKernelGGL((pp_block_kernel<true, false>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, Ff1QkvArgs{}, OutGluArgs{});
  return 0;
}
// out-projection + GLU in the prologue of the tail kernels (OGF): the block runs as attention -> this, two launches.
// g: what pp_out_glu_kernel would have been given (ctx, x1, the LayerNorm parameters, the out_glu stream and its scales;
// x2 and u are not written).  Needs the depthwise fold (a.dw_wd / dw_T / dw_pad set; a.dw_u is not read).
bool pp_og_fold_ok(const TailFf2Args& a, const OutGluArgs& g) {
  // MI355ASR_PP_OGF=0: out-projection + GLU as its own launch (pp_out_glu_kernel)
  static const bool on = mi355_env("MI355ASR_PP_OGF", 1) != 0;
  static const bool og_on = mi355_env("MI355ASR_PP_OUTGLU", 1) != 0;
  return on && og_on && pp_enabled() && a.pp_slabs && g.pp_slabs && g.ctx && g.x1 && a.dw_wd && a.dw_T > 0 && a.M > 0 && a.M % a.dw_T == 0 &&
         a.M == g.M && (a.dw_pad == 15 || a.dw_pad == 31) && pp_dw_fold_ok(a.dw_T, DW_K);
}
int launch_pp_og_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, const OutGluArgs& g, hipStream_t s) {
  if (!pp_og_fold_ok(a, g) || !b.pp_slabs || a.M != b.M) return -1;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL((pp_block_kernel<true, true, 0, true, true>), dim3((a.dw_T + 63) / 64, a.M / a.dw_T), dim3(LD_THREADS), 0, s, a, b, g);
  return 0;
}
bool pp_head_fold_ok(int M, int n_valid, int groups) {
  // MI355ASR_PP_HEADF=0: the class head as its own launch (pp_head_kernel); the switches of that kernel apply here too
  static const bool on = mi355_env("MI355ASR_PP_HEADF", 1) != 0;
  static const bool head_on = mi355_env("MI355ASR_PP_HEAD", 1) != 0;
  static const bool ring_on = mi355_env("MI355ASR_HEAD_RING", 1) != 0;
  return on && head_on && ring_on && pp_enabled() && groups >= 1 && n_valid <= 144 * groups && M > 0;
}
int launch_pp_og_tail_ff2(const TailFf2Args& a, const OutGluArgs& g, hipStream_t s) {
  if (!pp_og_fold_ok(a, g)) return -1;
  note_scheme(SCHEME_F16X2);
  if (a.head_pp) {
    if (!pp_head_fold_ok(a.M, a.head_nvalid, a.head_groups)) return -1;
    hipLaunchKernelGGL((pp_block_kernel<true, false, 0, true, true, false, true>), dim3((a.dw_T + 63) / 64, a.M / a.dw_T), dim3(LD_THREADS), 0, s, a, Ff1QkvArgs{}, g);
    return 0;
  }
  hipLaunchKernelGGL((pp_block_kernel<true, false, 0, true, true>), dim3((a.dw_T + 63) / 64, a.M / a.dw_T), dim3(LD_THREADS), 0, s, a, Ff1QkvArgs{}, g);
  return 0;
}
bool pp_pre_fold_ok() {
  // MI355ASR_PP_PRE=0: the subsampling Dense and the CTC decoder's projection as their own launches
  static const bool on = mi355_env("MI355ASR_PP_PRE", 1) != 0;
  return on && pp_enabled();
}
int launch_pp_ff1_qkv(const Ff1QkvArgs& b, hipStream_t s) {
  if (!pp_enabled() || !b.pp_slabs || b.M <= 0) return -1;
  if (b.xq_pe && (b.pre_pp || b.qkv_T > 0 || b.xq_U < 1)) return -1;
  const int tiles = (b.M + 15) / 16;
  note_scheme(SCHEME_F16X2);
  if (b.pre_pp) {
    if (!pp_pre_fold_ok() || !b.pre_x || b.pre_chunks < 1) return -1;
    hipLaunchKernelGGL((pp_block_kernel<false, true, 0, false, false, true>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, TailFf2Args{}, b, OutGluArgs{});
    return 0;
  }
  if (launch_ns1_ff1_qkv(b, s) == 0) return 0;           // round 6, small batches: one 16-token tile per workgroup (fused_ns.hip)
  hipLaunchKernelGGL((pp_block_kernel<false, true>), dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, TailFf2Args{}, b, OutGluArgs{});
  return 0;
}
