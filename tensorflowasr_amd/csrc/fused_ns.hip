// N-split block kernels for dmodel 144 (round 6): the token-local chains of a ConformerBlock with the HIDDEN dimension split
// over the four waves of a workgroup instead of the tokens.
//
// The pair-pipelined kernels (fused_pp.hip) give each consumer wave 16 tokens and the whole hidden dimension: every wave
// reads every weight fragment from LDS (0.67 ds_read_b128 per MFMA), four loader waves stream 2.4 MB of weights per
// workgroup through a seven-slot LDS ring with a workgroup barrier per 20 fragments, and the measurements of rounds 3-5
// (profiles/r03_pp_experiments.md, DESIGN.md "round 4") price the result at ~29 cycles per 16-cycle MFMA: fragment reads,
// slab DMA (power: 1.97 against 2.17 GHz), barriers, and one wave per SIMD that carries all of it.
//
// Here a workgroup is FOUR waves (one per SIMD, up to 512 registers each) that own 64 tokens TOGETHER:
//   * wave w computes hidden pairs w, w + 4, ... of  y += W2 act(W1aug [x ; 1])  for all four 16-token tiles -- a weight
//     fragment is needed by exactly ONE wave of the workgroup, so it goes from L2 straight into that wave's registers
//     (global_load_dwordx4, ten fragments in flight): no LDS ring, no loader waves, no per-slot barrier, a quarter of the
//     fragment traffic per MFMA (one fragment feeds 4 x 1.5 MFMAs);
//   * the operand rows x (after LayerNorm, split into fp16 hi + lo terms by the wave that owns the tile) are exchanged once
//     per chain through 40 KB of LDS; a wave reads 0.33 operand fragments per MFMA in the W1 half of a unit, none in the W2 half
//     (the hidden operand is built in registers, as before);
//   * every wave ends a chain with partial outputs of all 64 tokens: three of its four tiles go to their owners through
//     LDS (108 KB, one barrier), the owner adds them to its own -- one exchange per chain;
//   * 18 pairs = 4 x 4 + 2 and 9 = 4 x 2 + 1 do not divide by four: the left-over pairs are split over the waves by TOKEN
//     tile (a half unit on two tiles, a quarter unit on one), so every wave issues the same number of MFMAs.
// A wave keeps its accumulators by SLOT: slot s of wave w is token tile w ^ {1, 2, 3, 0}[s] -- slot 3 is the wave's own tile,
// slots {0, 1} of waves (0, 1) and of waves (2, 3) cover all four tiles (half units), slot 0 of the four waves likewise
// (quarter units): register indices stay compile-time constants, only LDS addresses depend on the wave.
// Units are software-pipelined as in fused_pp.hip -- B(k - 1) | A(k + 1) | activation + split of pair k in <= 2-instruction
// slots behind the MFMAs (prep2_sched.inc) -- but as C++ templates the compiler allocates registers and counts the waits for.
// Arithmetic, scales and fragment format are those of fused_pp.hip (two fp16 terms, three products, per-token power-of-two
// scales; api.hip: pack_half32); only the order in which a token's hidden features are summed differs.
// Reference semantics: asr/models/conformer_blocks.py:126-134, :164-170, :209-219, :259-265.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "launch.h"
#include "wstream.h"
#include "split_f16.h"

#ifndef NS_NP
#define NS_NP 20
#endif

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;      // 9
constexpr int KS = 5;           // 32-wide steps over K = 144 (+ the bias row 144)
constexpr int NP = NS_NP;       // weight fragments in flight per wave (the pool)
constexpr int NTT = 4;          // token tiles per workgroup

#define NS_FENCE __builtin_amdgcn_sched_barrier(0)
// timing-only variants (wrong results; tools/build_variant.py NAME fused_ns.hip -DNS_DIAG=n): bit 0 = the chain runs twice (is the
// first pass bound by instruction fetch?), 1 = no fragment loads after the first ten, 2 = no activation / split work, 3 = no
// operand reads from LDS after the first
#ifndef NS_DIAG
#define NS_DIAG 0
#endif
// bit 4: cycle stamps (s_memtime) at the phase boundaries of workgroup 7, printed per wave at the end of the kernel
#define NS_STAMP(i) do { if constexpr ((NS_DIAG & 16) != 0) { if (blockIdx.x == 7) ns_stamp[i] = __builtin_readcyclecounter(); } } while (0)
#define NS_STAMP_DECL unsigned long long ns_stamp[24] = {}

DEV f32x4 ns_mfma(u32x4_t w, u32x4_t x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), acc, 0, 0, 0);
}
DEV float ns_row_max(const f32x4 (&xs)[KB]) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < KB; ++i)
    m = fmaxf(fmaxf(m, fmaxf(fabsf(xs[i].x), fabsf(xs[i].y))), fmaxf(fabsf(xs[i].z), fabsf(xs[i].w)));
  return group_max(m);
}
struct NsTok { float sx, k1, ik2, s2, inv2; };
DEV NsTok ns_chain_scales(const PpChainSc& c, float xmax) {      // fused_pp.hip: pp_chain_scales
  NsTok t;
  t.sx = pp_pow2_scale(xmax);
  const float sh = pp_pow2_scale(fmaf(c.l1, xmax, c.bmax));
  const float u1 = c.sw1 * t.sx;
  t.k1 = -1.4426950408889634f * pp_recip_pow2(u1);
  t.ik2 = u1 * pp_recip_pow2(sh);
  t.s2 = c.sw2 * sh;
  t.inv2 = pp_recip_pow2(t.s2);
  return t;
}
DEV void ns_ln(f32x4 (&xs)[KB], const float* ga, const float* be, int g4, float eps) {
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * lds4(ga, kb, g4) + lds4(be, kb, g4);
}

// ---- the exchange areas in LDS -------------------------------------------------------------------------------------------------
struct NsXf { u32x4_t f[KS][2][64]; };                     // one token tile's operand: five 32-wide steps x (hi, lo) x 64 lanes (10 KB)
struct NsRed { f32x4 v[3][KB][64]; };                      // partial outputs for one owner from the three other waves (27 KB)
struct NsTokLds { float k1[16], ik2[16], aux[16]; };       // per token of a tile, written by its owner

struct NsCtx {
  int w, lane, g4, t;
  const NsXf* xf;           // [tiles]
  const NsTokLds* tk;       // [tiles]
  DEV int tt(int slot) const { return w ^ ((slot + 1) & 3); }      // slots 0..3 -> w ^ {1, 2, 3, 0}
};

// the split operand of this wave's own tile into the exchange area (k-slot 144 carries 1.0 in the operand's unit: the bias row)
DEV void ns_put_operand(NsXf& dst, const f32x4 (&xs)[KB], int lane, int g4, float sx) {
  const f32x4 s4 = splat4(sx);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    Split8 f;
    if (s < KS - 1) f = split8(xs[2 * s] * s4, xs[2 * s + 1] * s4);
    else {
      f32x4 oh = splat4(0.f);
      oh.x = g4 == 0 ? sx : 0.0f;
      f = split8(xs[KB - 1] * s4, oh);
    }
    dst.f[s][0][lane] = f.t[0];
    dst.f[s][1][lane] = f.t[1];
  }
}

// ---- one pipelined phase of a chain ----------------------------------------------------------------------------------------------
//   B part:  y[s][tile]  += W2[pair b][tile] fB[s]          s < NTB      (27 NTB MFMAs, 18 fragments + 2 idle positions)
//   A part:  hA[s][0..1] += W1[:, pair a] x[tile of s]       s < NTA      (30 NTA MFMAs, 20 fragments, 8 NTA... operand reads)
//   prep:    fP[s] = split(swish(hP[s][0..1]))               s < NTP      (40 NTP slots of <= 2 VALU instructions)
// Fragment stream of a phase: positions [B: (tile 0 lo, tile 0 hi, tile 1 lo, ...), two idle][A: per step (b0 lo, b1 lo, b0 hi,
// b1 hi)]; the fragment at position q lives in pool[q % NP]; it is released after its last MFMA and the load of position q + NP --
// of this phase, or of the next one (NXT: 0 none, 1 its B part comes first, 2 its A part) -- is issued there.  Every phase has 20
// or 40 positions, so the pool is in the same state at every phase boundary.
// Products of a fragment pair, smallest first: (w lo, x hi), (w hi, x lo), (w hi, x hi).
struct NsPtr { const u32x4_t *b, *a; };      // + lane; b: W2 fragments of the pair [9][2][64]; a: W1 fragments [KS][NT1][2][64] at tile 2 pair
template <int NT1>
DEV const u32x4_t* ns_frag_b(const NsPtr& p, int q) {        // q < 18: tile q / 2, lo first
  return p.b + ((q >> 1) * 2 + ((q & 1) ^ 1)) * 64;
}
template <int NT1>
DEV const u32x4_t* ns_frag_a(const NsPtr& p, int q) {        // q < 20: step q / 4, (b0 lo, b1 lo, b0 hi, b1 hi)
  const int s = q >> 2, j = q & 3;
  return p.a + ((s * NT1 + (j & 1)) * 2 + ((j >> 1) ^ 1)) * 64;
}

template <int NTB, int NTA, int NTP, int NXT, int NT1>
DEV void ns_phase(f32x4 (&y)[NTT][KB], f32x4 (&hA)[NTT][2], f32x4 (&hP)[NTT][2], const Split8 (&fB)[NTT], Split8 (&fP)[NTT],
                  u32x4_t (&pool)[NP], const NsPtr& cur, const NsPtr& nxt, const NsCtx& c, const float (&k1)[NTT], const float (&ik2)[NTT]) {
  constexpr int NB = NTB ? 20 : 0, NA = NTA ? 20 : 0, NPOS = NB + NA;
  constexpr int MB = 27 * NTB, MA = 30 * NTA, M = MB + MA, SL = PREP2_SLOTS * NTP;
  static_assert(NPOS % NP == 0 && NPOS > 0, "the pool returns to its canonical state at a phase boundary");
  // position q of THIS phase's stream (q may run into the next phase's): real fragment? its address
  auto load_pos = [&](auto Q) {
    constexpr int q = decltype(Q)::value;
    if constexpr ((NS_DIAG & 2) != 0) return;
    if constexpr (q < NPOS) {
      if constexpr (q < NB) {
        if constexpr (q < 18) pool[q % NP] = ns_frag_b<NT1>(cur, q)[0];
      } else {
        pool[q % NP] = ns_frag_a<NT1>(cur, q - NB)[0];
      }
    } else if constexpr (NXT == 1) {
      if constexpr (q - NPOS < 18) pool[q % NP] = ns_frag_b<NT1>(nxt, q - NPOS)[0];
    } else if constexpr (NXT == 2) {
      pool[q % NP] = ns_frag_a<NT1>(nxt, q - NPOS)[0];
    }
  };
  auto release = [&](auto Q) { load_pos(std::integral_constant<int, decltype(Q)::value + NP>()); };
  Prep2Ctx pc[NTT] = {{hP[0][0], hP[0][1], fP[0], k1[0], ik2[0], 0.f, 0.f, 0.f, 0.f, 0u},
                      {hP[1][0], hP[1][1], fP[1], k1[1], ik2[1], 0.f, 0.f, 0.f, 0.f, 0u},
                      {hP[2][0], hP[2][1], fP[2], k1[2], ik2[2], 0.f, 0.f, 0.f, 0.f, 0u},
                      {hP[3][0], hP[3][1], fP[3], k1[3], ik2[3], 0.f, 0.f, 0.f, 0.f, 0u}};
  // the prep slots that follow MFMA number m of the phase (uniformly spread; all of them by the last MFMA)
  auto prep_after = [&](auto Mi) {
    constexpr int m = decltype(Mi)::value;
    if constexpr (SL > 0 && !(NS_DIAG & 4)) {
      constexpr int lo = (m * SL) / M, hi = ((m + 1) * SL) / M;
      static_for<lo, hi>([&](auto S) {
        constexpr int sl = decltype(S)::value;
        Prep2Ctx& p = pc[sl / PREP2_SLOTS];
        prep2_slot<sl % PREP2_SLOTS>(p);
        // the slot's results are pinned HERE: they feed nothing before the next phase, and without a use the compiler emits the whole
        // activation + split of a pair in front of its first consumer -- the next phase's first MFMAs (seen in the ISA: 32 v_exp in
        // the first 200 instructions of a phase, none in the phase that was meant to carry them)
        asm volatile("" : "+v"(p.ta), "+v"(p.tb), "+v"(p.m0), "+v"(p.m1), "+v"(p.hp), "+v"(p.lo), "+v"(p.hi), "+v"(p.out.t[0]), "+v"(p.out.t[1]));
      });
    }
  };
  if constexpr ((NS_DIAG & 4) != 0 && NTP > 0) {          // timing only: the operand is "defined" without any work
#pragma unroll
    for (int s = 0; s < NTP; ++s) asm volatile("; no prep" : "=v"(fP[s].t[0]), "=v"(fP[s].t[1]) : "v"(hP[s][0]), "v"(hP[s][1]));
  }
  // operand fragments of the A part: x hi double-buffered by step, x lo refilled as it is used
  u32x4_t xh[2][NTT], xl[NTT];
  if constexpr (NTA > 0) {
#pragma unroll
    for (int s = 0; s < NTA; ++s) {
      xh[0][s] = c.xf[c.tt(s)].f[0][0][c.lane];
      xl[s] = c.xf[c.tt(s)].f[0][1][c.lane];
    }
  }
  NS_FENCE;
  // ---- B part
  if constexpr (NTB > 0) {
    static_for<0, KB>([&](auto TI) {
      constexpr int tile = decltype(TI)::value;
      static_for<0, 3>([&](auto G) {
        constexpr int g = decltype(G)::value;
        static_for<0, NTB>([&](auto S) {
          constexpr int s = decltype(S)::value;
          constexpr int m = (tile * 3 + g) * NTB + s;
          constexpr int q = 2 * tile + (g == 0 ? 0 : 1);
          y[s][tile] = ns_mfma(pool[q % NP], g == 1 ? fB[s].t[1] : fB[s].t[0], y[s][tile]);
          prep_after(std::integral_constant<int, m>());
          if constexpr (s == NTB - 1 && g == 0) release(std::integral_constant<int, 2 * tile>());
          if constexpr (s == NTB - 1 && g == 2) {
            release(std::integral_constant<int, 2 * tile + 1>());
            if constexpr (tile == KB - 1) { release(std::integral_constant<int, 18>()); release(std::integral_constant<int, 19>()); }
          }
          NS_FENCE;
        });
      });
    });
  }
  // ---- A part
  if constexpr (NTA > 0) {
    static_for<0, KS>([&](auto SI) {
      constexpr int st = decltype(SI)::value;
      static_for<0, 3>([&](auto G) {
        constexpr int g = decltype(G)::value;
        static_for<0, NTA>([&](auto S) {
          constexpr int s = decltype(S)::value;
          static_for<0, 2>([&](auto Bi) {
            constexpr int b = decltype(Bi)::value;
            constexpr int m = MB + ((st * 3 + g) * NTA + s) * 2 + b;
            constexpr int q = NB + 4 * st + (g == 0 ? 0 : 2) + b;
            hA[s][b] = ns_mfma(pool[q % NP], g == 1 ? xl[s] : xh[st & 1][s], hA[s][b]);
            prep_after(std::integral_constant<int, m>());
            if constexpr (b == 1 && st + 1 < KS && !(NS_DIAG & 8)) {
              // x lo of the next step as soon as this step's is used; x hi of the next step behind the first group
              if constexpr (g == 1) xl[s] = c.xf[c.tt(s)].f[st + 1][1][c.lane];
              if constexpr (g == 0) xh[(st + 1) & 1][s] = c.xf[c.tt(s)].f[st + 1][0][c.lane];
            }
            if constexpr (s == NTA - 1 && b == 1 && g == 0) { release(std::integral_constant<int, NB + 4 * st>()); release(std::integral_constant<int, NB + 4 * st + 1>()); }
            if constexpr (s == NTA - 1 && b == 1 && g == 2) { release(std::integral_constant<int, NB + 4 * st + 2>()); release(std::integral_constant<int, NB + 4 * st + 3>()); }
            NS_FENCE;
          });
        });
      });
    });
  }
  if constexpr (M == 0) static_assert(SL == 0, "prep rides behind MFMAs");
}

// first NP fragments of a chain's first phase (the A part of wave w's first pair); issued as early as the caller can -- a
// fragment takes ~1 600 cycles from L2 when every workgroup of the chip asks for the same lines (measured: profiles/r06_ns_*)
template <int P>
DEV void ns_prime_chain(u32x4_t (&pool)[NP], const u32x4_t* w1, int w, int lane) {
  const NsPtr p{nullptr, w1 + (size_t)(2 * w) * (2 * 64) + lane};
  static_for<0, NP>([&](auto Q) { constexpr int q = decltype(Q)::value; pool[q] = ns_frag_a<2 * P>(p, q)[0]; });
}

// One chain  y += W2 act(W1aug [x ; 1])  over P hidden pairs (H = 32 P), N-split over the four waves.  P = 4 F + R: wave w takes
// the full pairs w, w + 4, ...; R = 2 (ff modules: 18): pair 4 F + (w >> 1) as a half unit on slots {0, 1}; R = 1 (conv tail: 9):
// pair 4 F as a quarter unit on slot 0.  On entry: the operands of the four tiles and their (k1, ik2) are in LDS (barrier
// passed), y holds this wave's initial partial (the owner's slot 3: residual + bias in the output unit; the others: zero).
// w1 / w2: pack_half32 fragments [KS][2 P][2][64] / [P][9][2][64] of u32x4.
template <int P>
DEV void ns_chain(f32x4 (&y)[NTT][KB], u32x4_t (&pool)[NP], const u32x4_t* w1, const u32x4_t* w2, const NsCtx& c, unsigned long long (&ns_stamp)[24]) {
  constexpr int F = P / 4, R = P % 4, NT1 = 2 * P, NR = R == 2 ? 2 : (R == 1 ? 1 : 0);
  static_assert(F >= 2 && (R == 1 || R == 2), "ff modules (18 pairs) and the conv tail (9)");
  float k1[NTT], ik2[NTT];
#pragma unroll
  for (int s = 0; s < NTT; ++s) { k1[s] = c.tk[c.tt(s)].k1[c.t]; ik2[s] = c.tk[c.tt(s)].ik2[c.t]; }
  auto ptr = [&](int pair) { return NsPtr{w2 + (size_t)pair * (KB * 2 * 64) + c.lane, w1 + (size_t)(2 * pair) * (2 * 64) + c.lane}; };
  const int prem = 4 * F + (R == 2 ? (c.w >> 1) : 0);        // the left-over pair this wave shares
  f32x4 h0[NTT][2], h1[NTT][2];
  Split8 f0[NTT], f1[NTT];
  const f32x4 zero = splat4(0.f);
  auto clear = [&](f32x4 (&h)[NTT][2]) {
#pragma unroll
    for (int s = 0; s < NTT; ++s) { h[s][0] = zero; h[s][1] = zero; }
  };
  NsPtr p0 = ptr(c.w), p1 = ptr(c.w + 4);
  clear(h0);
  NS_STAMP(3);
  ns_phase<0, 4, 0, 2, NT1>(y, h0, h1, f0, f1, pool, p0, p1, c, k1, ik2);                       // A(0)
  NS_STAMP(4);
  clear(h1);
  {
    NsPtr nx = p0;                                                                                // next phase starts with B(0)
    ns_phase<0, 4, 4, 1, NT1>(y, h1, h0, f0, f0, pool, p1, nx, c, k1, ik2);                      // A(1) | prep(0) -> f0
  }
  NS_STAMP(5);
  // full units: B(k - 2) with f(k) | A(k) | prep(k - 1); k = 2 .. F - 1; the pairs alternate between (h0, f0) and (h1, f1)
  static_assert(F == 2 || F == 4, "two (conv tail) or four (ff modules) full pairs per wave");
  if constexpr (F == 4) {
    {
      NsPtr cu{ptr(c.w).b, ptr(c.w + 8).a}, nx{ptr(c.w + 4).b, nullptr};
      clear(h0);
      ns_phase<4, 4, 4, 1, NT1>(y, h0, h1, f0, f1, pool, cu, nx, c, k1, ik2);                    // B(0; f0) | A(2) -> h0 | prep(1) -> f1
    }
    NS_STAMP(6);
    {
      NsPtr cu{ptr(c.w + 4).b, ptr(c.w + 12).a}, nx{ptr(c.w + 8).b, nullptr};
      clear(h1);
      ns_phase<4, 4, 4, 1, NT1>(y, h1, h0, f1, f0, pool, cu, nx, c, k1, ik2);                    // B(1; f1) | A(3) -> h1 | prep(2) -> f0
    }
    NS_STAMP(7);
  }
  // the left-over pair rides behind the last two full pairs: (hL, fL) = the pair that is free
  {
    constexpr int kb = F - 2;                                                                     // B of full pair kb, then kb + 1
    NsPtr cu{ptr(c.w + 4 * kb).b, ptr(prem).a}, nx{ptr(c.w + 4 * (kb + 1)).b, nullptr};
    clear(h0);
    ns_phase<4, NR, 4, 1, NT1>(y, h0, h1, f0, f1, pool, cu, nx, c, k1, ik2);                     // B(kb; f0) | A(rem) -> h0 | prep(kb + 1) -> f1
  }
  NS_STAMP(8);
  {
    NsPtr cu{ptr(c.w + 4 * (F - 1)).b, nullptr}, nx{ptr(prem).b, nullptr};
    ns_phase<4, 0, NR, 1, NT1>(y, h1, h0, f1, f0, pool, cu, nx, c, k1, ik2);                     // B(kb + 1; f1) | prep(rem) -> f0
  }
  NS_STAMP(9);
  {
    NsPtr cu{ptr(prem).b, nullptr}, nx{nullptr, nullptr};
    ns_phase<NR, 0, 0, 0, NT1>(y, h0, h1, f0, f1, pool, cu, nx, c, k1, ik2);                     // B(rem; f0)
  }
  NS_STAMP(10);
}

// partial outputs of the three tiles this wave does not own -> their owners; on return (after the barrier inside) the owner's
// slot 3 holds the complete rows of its 16 tokens
DEV void ns_reduce(f32x4 (&y)[NTT][KB], NsRed* red, const NsCtx& c) {
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < KB; ++i) red[c.tt(s)].v[s][i][c.lane] = y[s][i];
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < KB; ++i) y[3][i] += red[c.w].v[s][i][c.lane];
}

// A plain layer  acc[j][s] = W[:, tile_j] x[tile of s]  for NCT column tiles (fragment pointers [KS][NT][2][64] + lane at the tile)
// and the four token tiles: 2 NCT fragments and 12 NCT MFMAs per step.  The fragments stream through the same rotating pool as the
// chains': position q = (step, lo of tile 0 .., hi of tile 0 ..) lives in pool[(OFF + q) % NP], is released after its last MFMA, and
// the load of position q + NP -- of this group, or of the next one (NNXT column tiles at wn) -- is issued there.  The first NP
// positions are in flight on entry (ns_prime_plain, or the group in front).
template <int NCT>
DEV const u32x4_t* ns_frag_p(const u32x4_t* const (&wt)[NCT], int NT, int q) {      // q < 10 NCT
  const int st = q / (2 * NCT), r = q % (2 * NCT), j = r % NCT, term = r < NCT ? 1 : 0;
  return wt[j] + (size_t)(st * NT) * (2 * 64) + term * 64;
}
template <int NCT, int OFF>
DEV void ns_prime_plain(u32x4_t (&pool)[NP], const u32x4_t* const (&wt)[NCT], int NT) {
  static_for<0, NP>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    if constexpr (q < 10 * NCT) pool[(OFF + q) % NP] = ns_frag_p<NCT>(wt, NT, q)[0];
  });
}
template <int NCT, int OFF, int NNXT>
DEV void ns_plain(f32x4 (&acc)[NCT][NTT], u32x4_t (&pool)[NP], const u32x4_t* const (&wt)[NCT], const u32x4_t* const (&wn)[NNXT > 0 ? NNXT : 1],
                  int NT, const NsCtx& c) {
  constexpr int NPOS = 10 * NCT;
  static_assert(NP <= NPOS, "a group is at least as long as the pool");
  auto release = [&](auto Q) {
    constexpr int q = decltype(Q)::value + NP;
    if constexpr ((NS_DIAG & 2) != 0) return;
    if constexpr (q < NPOS) pool[(OFF + q) % NP] = ns_frag_p<NCT>(wt, NT, q)[0];
    else if constexpr (NNXT > 0) {
      if constexpr (q - NPOS < 10 * NNXT) pool[(OFF + q) % NP] = ns_frag_p<(NNXT > 0 ? NNXT : 1)>(wn, NT, q - NPOS)[0];
    }
  };
  u32x4_t xh[2][NTT], xl[NTT];
#pragma unroll
  for (int s = 0; s < NTT; ++s) { xh[0][s] = c.xf[c.tt(s)].f[0][0][c.lane]; xl[s] = c.xf[c.tt(s)].f[0][1][c.lane]; }
  NS_FENCE;
  static_for<0, KS>([&](auto SI) {
    constexpr int st = decltype(SI)::value;
    static_for<0, 3>([&](auto G) {
      constexpr int g = decltype(G)::value;
      static_for<0, NTT>([&](auto S) {
        constexpr int s = decltype(S)::value;
        static_for<0, NCT>([&](auto J) {
          constexpr int j = decltype(J)::value;
          constexpr int q = st * 2 * NCT + (g == 0 ? 0 : NCT) + j;
          acc[j][s] = ns_mfma(pool[(OFF + q) % NP], g == 1 ? xl[s] : xh[st & 1][s], acc[j][s]);
          if constexpr (j == NCT - 1 && st + 1 < KS && !(NS_DIAG & 8)) {
            if constexpr (g == 1) xl[s] = c.xf[c.tt(s)].f[st + 1][1][c.lane];
            if constexpr (g == 0) xh[(st + 1) & 1][s] = c.xf[c.tt(s)].f[st + 1][0][c.lane];
          }
          if constexpr (s == NTT - 1 && (g == 0 || g == 2)) release(std::integral_constant<int, q>());
          NS_FENCE;
        });
      });
    });
  });
}

struct NsFf1Lds {
  NsXf xf[NTT];                    // 40 KB
  NsRed red[NTT];                  // 108 KB
  NsTokLds tk[NTT];
  float ln1g[D], ln1b[D], b2[D], ln2g[D], ln2b[D];
};

// ff_module_1 + q / k / v projections (pp_block_kernel<false, true> of fused_pp.hip, N-split): x0 -> x1, qkv
__global__ __launch_bounds__(BLOCK_THREADS, 1) void ns_ff1_qkv_kernel(Ff1QkvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ns_smem[];
  NsFf1Lds& L = *reinterpret_cast<NsFf1Lds*>(ns_smem);
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, t = lane & 15;
  const NsCtx c{w, lane, g4, t, L.xf, L.tk};
  NS_STAMP_DECL;
  NS_STAMP(0);
  const int tok0 = blockIdx.x * 64;
  const int tok = tok0 + 16 * w + t;
  const bool live = tok < a.M;
  const size_t row = (size_t)min(tok, a.M - 1) * D;
  const u32x4_t* w1 = reinterpret_cast<const u32x4_t*>(a.ns_w1);
  const u32x4_t* w2 = reinterpret_cast<const u32x4_t*>(a.ns_w2);
  u32x4_t pool[NP];
  ns_prime_chain<18>(pool, w1, w, lane);              // in flight under the whole prologue
  f32x4 xs[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x0 + row + 16 * kb + g4);
  for (int i = threadIdx.x; i < D; i += BLOCK_THREADS) {
    L.ln1g[i] = a.ff_ln_g[i]; L.ln1b[i] = a.ff_ln_b[i]; L.b2[i] = a.ff_b2[i]; L.ln2g[i] = a.att_ln_g[i]; L.ln2b[i] = a.att_ln_b[i];
  }
  __syncthreads();
  f32x4 y[NTT][KB];
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < KB; ++i) y[s][i] = splat4(0.f);
#pragma unroll
  for (int i = 0; i < KB; ++i) y[3][i] = lds4(L.b2, i, g4) + splat4(inv_fc) * xs[i];
  ns_ln(xs, L.ln1g, L.ln1b, g4, a.eps);
  const NsTok tk = ns_chain_scales(a.pp_sc, ns_row_max(xs));
#pragma unroll
  for (int i = 0; i < KB; ++i) y[3][i] = y[3][i] * splat4(tk.s2);
  ns_put_operand(L.xf[w], xs, lane, g4, tk.sx);
  if (g4 == 0) { L.tk[w].k1[t] = tk.k1; L.tk[w].ik2[t] = tk.ik2; }
  NS_STAMP(1);
  __syncthreads();
  NS_STAMP(2);
  ns_chain<18>(y, pool, w1, w2, c, ns_stamp);
  // q, k, v: 27 column tiles, wave w takes tiles w, w + 4, ... (7, 7, 7, 6; wave 3 repeats tile 26 and does not store it) in groups
  // of 3, 2, 2; the first fragments are requested before the exchange of the partial outputs
  const u32x4_t* wq = reinterpret_cast<const u32x4_t*>(a.ns_qkv) + lane;
  auto tile_ptr = [&](int tile) { return wq + (size_t)min(tile, 26) * (2 * 64); };
  const u32x4_t* const wt0[3] = {tile_ptr(w), tile_ptr(w + 4), tile_ptr(w + 8)};
  const u32x4_t* const wt1[2] = {tile_ptr(w + 12), tile_ptr(w + 16)};
  const u32x4_t* const wt2[2] = {tile_ptr(w + 20), tile_ptr(w + 24)};
  ns_prime_plain<3, 0>(pool, wt0, 27);
  ns_reduce(y, L.red, c);
  NS_STAMP(11);
#pragma unroll
  for (int i = 0; i < KB; ++i) { y[3][i] = splat4(a.fc * tk.inv2) * y[3][i]; xs[i] = y[3][i]; }        // x1 = x0 + fc (ffn + b2)
  if (live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x1 + row + 16 * i + g4, y[3][i]);
  }
  ns_ln(xs, L.ln2g, L.ln2b, g4, a.eps);
  const float sx = pp_pow2_scale(ns_row_max(xs));
  ns_put_operand(L.xf[w], xs, lane, g4, sx);          // every wave is past the barrier of ns_reduce: the old operands are dead
  if (g4 == 0) L.tk[w].aux[t] = pp_recip_pow2(a.pp_sw_qkv * sx);
  NS_STAMP(12);
  __syncthreads();
  NS_STAMP(13);
  float invq[NTT];
#pragma unroll
  for (int s = 0; s < NTT; ++s) invq[s] = L.tk[c.tt(s)].aux[t];
  auto store = [&](int tile, const f32x4 (&acc)[NTT]) {
    if (tile >= 27) return;
    const int q = tile / KB, i = tile - KB * q;
    const float scq = q == 0 ? a.qscale : 1.0f;
#pragma unroll
    for (int s = 0; s < NTT; ++s) {
      const int tk_ = tok0 + 16 * c.tt(s) + t;
      if (tk_ >= a.M) continue;
      const f32x4 v = acc[s] * splat4(scq * invq[s]);
      if (a.qkv_T > 0) {
        const int bq = tk_ / a.qkv_T, tq = tk_ - bq * a.qkv_T;
        float* plane = a.qkv + (size_t)q * a.M * D + ((size_t)bq * a.qkv_H * a.qkv_T + tq) * 36;
        const int f0 = 16 * i + g4, hq = f0 / 36;
        stg4(plane + (size_t)hq * a.qkv_T * 36 + (f0 - 36 * hq), v);
      } else {
        stg4(a.qkv + (size_t)tk_ * (3 * D) + 16 * tile + g4, v);
      }
    }
  };
  auto zero = [&](auto& acc) {
    for (auto& row : acc)
      for (auto& v : row) v = splat4(0.f);
  };
  {
    f32x4 acc[3][NTT];
    zero(acc);
    ns_plain<3, 0, 2>(acc, pool, wt0, wt1, 27, c);
    NS_STAMP(14);
    store(w, acc[0]); store(w + 4, acc[1]); store(w + 8, acc[2]);
  }
  NS_STAMP(15);
  {
    f32x4 acc[2][NTT];
    zero(acc);
    ns_plain<2, 30 % NP, 2>(acc, pool, wt1, wt2, 27, c);
    store(w + 12, acc[0]); store(w + 16, acc[1]);
  }
  {
    f32x4 acc[2][NTT];
    zero(acc);
    const u32x4_t* const none[1] = {nullptr};
    ns_plain<2, 50 % NP, 0>(acc, pool, wt2, none, 27, c);
    store(w + 20, acc[0]); store(w + 24, acc[1]);
  }
  NS_STAMP(16);
  if constexpr ((NS_DIAG & 16) != 0) {
    if (blockIdx.x == 7 && lane == 0) {
      printf("NSSTAMP w%d:", w);
      for (int i = 1; i <= 16; ++i) printf(" %d:%llu", i, ns_stamp[i] - ns_stamp[0]);
      printf("\n");
    }
  }
}

}  // namespace

bool ns_enabled() {
  // MI355ASR_NS=1: ff_module_1 + qkv (when no layer in front rides in the launch) on the N-split kernel instead of the pair-pipelined
  // one.  OFF by default: measured 37.6 us against 35.2 (profiles/r06_ns_experiments.md) -- a 1 KB global_load_dwordx4 costs the
  // issuing wave ~65 cycles of issue, which is why fused_pp.hip gives the weight stream to loader waves of its own.  Read at
  // mi355asr_finalize_weights too: the plain-order fragments (2.4 MB per block) are only packed when the switch is on.
  static const bool on = [] { const char* v = getenv("MI355ASR_NS"); return v ? atoi(v) != 0 : false; }();
  return on;
}
int launch_ns_ff1_qkv(const Ff1QkvArgs& b, hipStream_t s) {
  if (!ns_enabled() || !b.ns_w1 || !b.ns_w2 || !b.ns_qkv || b.M <= 0 || b.pre_pp) return -1;
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)ns_ff1_qkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NsFf1Lds)) == hipSuccess;
  }();
  if (!attr) return -1;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL(ns_ff1_qkv_kernel, dim3((b.M + 63) / 64), dim3(BLOCK_THREADS), sizeof(NsFf1Lds), s, b);
  return 0;
}
