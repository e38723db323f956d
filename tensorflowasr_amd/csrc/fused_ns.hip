// N-split block kernels for dmodel 144, SMALL batches (round 6): one 16-token tile per workgroup, the hidden dimension and the
// column tiles of the plain layers split over the workgroup's eight waves.
//
// The pair-pipelined kernels (fused_pp.hip) give each consumer wave 16 tokens and the whole hidden dimension; a launch streams the
// block's 2.4 MB of weights through every workgroup's LDS ring, which is the right trade from ~4 000 tokens up (one workgroup per CU
// and 64 tokens each).  For ONE utterance -- test_asr.py's call pattern (test_asr.py:186-219) -- it leaves 4 of 256 CUs busy for 60 us
// per launch: a consumer wave walks the whole weight stream for its 16 tokens whatever the batch is.  Here the stream of a block is
// spread over (tiles x 8 waves): wave w computes hidden pairs w, w + 8, ... of  y += W2 act(W1aug [x ; 1])  and the column tiles w,
// w + 8, ... of the plain layers for the tile; a weight fragment is needed by exactly ONE wave, so it goes from L2 straight into that
// wave's registers (global_load_dwordx4, ten fragments in flight: no LDS ring, no loader waves, no per-slot barrier); every wave
// holds the tile's rows (identical registers in all waves: LayerNorm, scales and the operand split are computed redundantly, no
// operand exchange), and partial outputs meet in LDS once per chain (reduce-scatter + all-gather, fixed order: all waves continue
// with bit-identical rows).  What bounds a launch is ~65 cycles of ISSUE per 1 KB fragment load and wave (measured:
// profiles/r06_ns_experiments.md) -- which is also why the same split does not beat the loader-wave design at full batches (a
// four-wave, 64-token version of this file was built, measured at 37.6 against 35.2 us and removed; commit 6b61327 has it).
// Units are software-pipelined as in fused_pp.hip -- B(k - 1) | A(k + 1) | activation + split of pair k in <= 2-instruction slots
// behind the MFMAs (prep2_sched.inc) -- but as C++ templates: the compiler allocates registers and counts the waits.  One trap:
// pure VALU work that feeds nothing before the next phase is emitted in front of its first USE (sched_barrier pins the machine
// scheduler, not instruction selection), so every slot's results are tied to an empty asm volatile.
// Arithmetic, scales and fragment format are those of fused_pp.hip (two fp16 terms, three products, per-token power-of-two scales;
// api.hip: pack_half32); only the order in which a token's hidden features are summed differs.
// Reference semantics: asr/models/conformer_blocks.py:126-134, :164-170, :209-219, :259-265; layers/multihead_attention.py:151-188.
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
constexpr int KS = 5;           // 32-wide steps over K = 144 (+ the bias row 144)

#define NS_FENCE __builtin_amdgcn_sched_barrier(0)
// diagnostics (tools/build_variant.py NAME fused_ns.hip -DNS_DIAG=n): 32 = cycle stamps (s_memtime) at the stages of ns1_tail_kernel,
// printed for one workgroup (tools/sessions/r06_ns1_stamp.sh); 64 = timing only, WRONG results: the in-launch attention without its
// fp32 -> fp16-pair split of K / V (what pre-split fragments from the producer would save)
#ifndef NS_DIAG
#define NS_DIAG 0
#endif
#define NS1_STAMP(i) do { if constexpr ((NS_DIAG & 32) != 0) { if (blockIdx.x == 3 && blockIdx.y == 0) st1[i] = __builtin_readcyclecounter(); } } while (0)

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
// fragment pointers of a hidden pair, WAVE-UNIFORM (the lane is added as a 32-bit index at the load: scalar base + vector offset
// instead of 64-bit address arithmetic per fragment): b = its W2 fragments [9][2][64], a = its W1 fragments [KS][NT1][2][64] at tile
// 2 pair; lane = this lane's index
struct NsPtr { const u32x4_t *b, *a; unsigned lane; };
template <int NT1>
DEV u32x4_t ns_frag_a(const NsPtr& p, int q) {        // q < 20: step q / 4, (b0 lo, b1 lo, b0 hi, b1 hi)
  const int s = q >> 2, j = q & 3;
  return p.a[((s * NT1 + (j & 1)) * 2 + ((j >> 1) ^ 1)) * 64 + p.lane];
}

// ---- the workgroup's exchange areas ------------------------------------------------------------------------------------------------
constexpr int NP1 = 10;         // fragments in flight per wave here: eight waves per workgroup have 256 registers each
template <int NW>
struct Ns1Lds {
  f32x4 red[NW][KB][64];            // partial outputs (chains, the layer in front) / column tiles of a plain layer, one slab per wave
  f32x4 sum[KB][64];                // their sums
  float par[12][D];                 // the kernel's parameter vectors (LayerNorm gamma / beta, biases): one L2 round trip at the start
};

// partial outputs of all NW waves -> their sum, the same in every wave (fixed order 0, 1, ...); `live` waves contributed
template <int NW>
DEV void ns1_allreduce(f32x4 (&y)[KB], Ns1Lds<NW>& L, int w, int lane) {
  // reduce-scatter, then all-gather: wave w sums column tiles w, w + NW, ... over the NW partials (fixed order) and publishes them;
  // every wave then reads the nine sums -- 8 + 9 fragment reads per wave instead of 72 (the all-to-all moved 576 KB through LDS
  // per exchange: 4 - 8 k cycles by the stamps)
  __syncthreads();                                    // the previous exchange has been read by everyone
#pragma unroll
  for (int i = 0; i < KB; ++i) L.red[w][i][lane] = y[i];
  __syncthreads();
  for (int i = w; i < KB; i += NW) {
    f32x4 a = L.red[0][i][lane];
#pragma unroll
    for (int k = 1; k < NW; ++k) a += L.red[k][i][lane];
    L.sum[i][lane] = a;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = L.sum[i][lane];
}

// One phase of a chain for ONE token tile (operands in registers):
//   B part (HB): y[tile] += W2[pair b][tile] fB        27 MFMAs, tiles in groups of three so that an accumulator rests two MFMAs
//   A part (HA): hA[0..1] += W1[:, pair a] x            30 MFMAs
//   prep   (HP): fP = split(swish(hP[0..1]))             40 slots behind the MFMAs (all of them in a row when the phase has none)
// positions: B = per group (lo of three tiles, hi of three tiles), 18 + 2 idle; A = per step (b0 lo, b1 lo, b0 hi, b1 hi)
DEV u32x4_t ns1_frag_b(const NsPtr& p, int q) {        // q < 18
  const int g = q / 6, j = q % 6;
  return p.b[((3 * g + j % 3) * 2 + (j < 3 ? 1 : 0)) * 64 + p.lane];
}
template <bool HB, bool HA, bool HP, int NXT, int NT1>
DEV void ns1_phase(f32x4 (&y)[KB], f32x4 (&hA)[2], f32x4 (&hP)[2], const Split8& fB, Split8& fP, const Split8 (&xf)[KS],
                   u32x4_t (&pool)[NP1], const NsPtr& cur, const NsPtr& nxt, float k1, float ik2) {
  constexpr int NB = HB ? 20 : 0, NA = HA ? 20 : 0, NPOS = NB + NA;
  constexpr int MB = HB ? 27 : 0, MA = HA ? 30 : 0, M = MB + MA, SL = HP ? PREP2_SLOTS : 0;
  auto load_pos = [&](auto Q) {
    constexpr int q = decltype(Q)::value;
    if constexpr (q < NPOS) {
      if constexpr (q < NB) {
        if constexpr (q < 18) pool[q % NP1] = ns1_frag_b(cur, q);
      } else {
        pool[q % NP1] = ns_frag_a<NT1>(cur, q - NB);
      }
    } else if constexpr (NXT == 1) {
      if constexpr (q - NPOS < 18) pool[q % NP1] = ns1_frag_b(nxt, q - NPOS);
    } else if constexpr (NXT == 2) {
      pool[q % NP1] = ns_frag_a<NT1>(nxt, q - NPOS);
    }
  };
  auto release = [&](auto Q) { load_pos(std::integral_constant<int, decltype(Q)::value + NP1>()); };
  Prep2Ctx pc{hP[0], hP[1], fP, k1, ik2, 0.f, 0.f, 0.f, 0.f, 0u};
  auto slot = [&](auto S) {
    prep2_slot<decltype(S)::value>(pc);
    asm volatile("" : "+v"(pc.ta), "+v"(pc.tb), "+v"(pc.m0), "+v"(pc.m1), "+v"(pc.hp), "+v"(pc.lo), "+v"(pc.hi), "+v"(pc.out.t[0]), "+v"(pc.out.t[1]));
  };
  auto prep_after = [&](auto Mi) {
    constexpr int m = decltype(Mi)::value;
    if constexpr (SL > 0 && M > 0) {
      constexpr int lo = (m * SL) / M, hi = ((m + 1) * SL) / M;
      static_for<lo, hi>(slot);
    }
  };
  if constexpr (M == 0 && SL > 0) {                  // a wave with a single pair: nothing to hide the activation behind
    static_for<0, SL>(slot);
    static_assert(NPOS == 0 || true, "");
  }
  if constexpr (HB) {
    static_for<0, 3>([&](auto Gi) {
      constexpr int g = decltype(Gi)::value;
      static_for<0, 3>([&](auto Pi) {
        constexpr int pr = decltype(Pi)::value;                     // 0: lo x hi, 1: hi x lo, 2: hi x hi
        static_for<0, 3>([&](auto Ti) {
          constexpr int i = decltype(Ti)::value, tile = 3 * g + i;
          constexpr int m = (g * 3 + pr) * 3 + i;
          constexpr int q = 6 * g + (pr == 0 ? 0 : 3) + i;
          y[tile] = ns_mfma(pool[q % NP1], pr == 1 ? fB.t[1] : fB.t[0], y[tile]);
          prep_after(std::integral_constant<int, m>());
          if constexpr (i == 2 && pr == 0) { release(std::integral_constant<int, 6 * g>()); release(std::integral_constant<int, 6 * g + 1>()); release(std::integral_constant<int, 6 * g + 2>()); }
          if constexpr (i == 2 && pr == 2) {
            release(std::integral_constant<int, 6 * g + 3>()); release(std::integral_constant<int, 6 * g + 4>()); release(std::integral_constant<int, 6 * g + 5>());
            if constexpr (g == 2) { release(std::integral_constant<int, 18>()); release(std::integral_constant<int, 19>()); }
          }
          NS_FENCE;
        });
      });
    });
  }
  if constexpr (HA) {
    static_for<0, KS>([&](auto SI) {
      constexpr int st = decltype(SI)::value;
      static_for<0, 3>([&](auto Pi) {
        constexpr int pr = decltype(Pi)::value;
        static_for<0, 2>([&](auto Bi) {
          constexpr int b = decltype(Bi)::value;
          constexpr int m = MB + (st * 3 + pr) * 2 + b;
          constexpr int q = NB + 4 * st + (pr == 0 ? 0 : 2) + b;
          hA[b] = ns_mfma(pool[q % NP1], pr == 1 ? xf[st].t[1] : xf[st].t[0], hA[b]);
          prep_after(std::integral_constant<int, m>());
          if constexpr (b == 1 && pr == 0) { release(std::integral_constant<int, NB + 4 * st>()); release(std::integral_constant<int, NB + 4 * st + 1>()); }
          if constexpr (b == 1 && pr == 2) { release(std::integral_constant<int, NB + 4 * st + 2>()); release(std::integral_constant<int, NB + 4 * st + 3>()); }
          NS_FENCE;
        });
      });
    });
  }
}

// y += W2 act(W1aug [x ; 1]) for one tile; this wave's pairs w, w + NW, ... (n of them, possibly none) as a software pipeline with
// a run-time trip count (the roles of the two hidden / operand register sets are swapped by copies: 16 moves per unit)
// the first NP1 fragments of wave w's first pair: requested by the caller as early as it can (in front of the exchange / LayerNorm /
// operand split that precede a chain: a fragment takes an L2 round trip, and a chain that starts cold waits for it with idle pipes)
template <int P, int NW>
DEV void ns1_prime(u32x4_t (&pool)[NP1], const u32x4_t* w1, int w, int lane) {
  const NsPtr p0{nullptr, w1 + (size_t)(2 * min(w, P - 1)) * (2 * 64), (unsigned)lane};
  static_for<0, NP1>([&](auto Q) { constexpr int q = decltype(Q)::value; pool[q] = ns_frag_a<2 * P>(p0, q); });
}
template <int P, int NW>
DEV void ns1_chain(f32x4 (&y)[KB], const Split8 (&xf)[KS], u32x4_t (&pool)[NP1], const u32x4_t* w1, const u32x4_t* w2, int w, int lane, float k1, float ik2) {
  constexpr int NT1 = 2 * P;
  const int n = w < P ? (P - w + NW - 1) / NW : 0;
  if (n == 0) return;
  auto ptr = [&](int k) { const int pair = w + NW * k; return NsPtr{w2 + (size_t)pair * (KB * 2 * 64), w1 + (size_t)(2 * pair) * (2 * 64), (unsigned)lane}; };
  const f32x4 zero = splat4(0.f);
  f32x4 hacc[2] = {zero, zero}, hprep[2];
  Split8 fuse, fbuild;
  {
    const NsPtr p0 = ptr(0);
    if (n == 1) {
      ns1_phase<false, true, false, 1, NT1>(y, hacc, hprep, fuse, fbuild, xf, pool, p0, p0, k1, ik2);           // A(0); then its own W2
      hprep[0] = hacc[0]; hprep[1] = hacc[1];
      ns1_phase<false, false, true, 0, NT1>(y, hacc, hprep, fuse, fbuild, xf, pool, p0, p0, k1, ik2);          // prep(0)
      ns1_phase<true, false, false, 0, NT1>(y, hacc, hprep, fbuild, fuse, xf, pool, p0, p0, k1, ik2);          // B(0)
      return;
    }
    const NsPtr p1 = ptr(1);
    ns1_phase<false, true, false, 2, NT1>(y, hacc, hprep, fuse, fbuild, xf, pool, p0, p1, k1, ik2);             // A(0)
    hprep[0] = hacc[0]; hprep[1] = hacc[1]; hacc[0] = zero; hacc[1] = zero;
    ns1_phase<false, true, true, 1, NT1>(y, hacc, hprep, fuse, fbuild, xf, pool, p1, p0, k1, ik2);              // A(1) | prep(0)
    fuse = fbuild; hprep[0] = hacc[0]; hprep[1] = hacc[1];
  }
#pragma unroll 1
  for (int k = 2; k < n; ++k) {                       // B(k - 2; fuse) | A(k) | prep(k - 1) -> fbuild
    const NsPtr cu{ptr(k - 2).b, ptr(k).a, (unsigned)lane}, nx{ptr(k - 1).b, nullptr, (unsigned)lane};
    hacc[0] = zero; hacc[1] = zero;
    ns1_phase<true, true, true, 1, NT1>(y, hacc, hprep, fuse, fbuild, xf, pool, cu, nx, k1, ik2);
    fuse = fbuild; hprep[0] = hacc[0]; hprep[1] = hacc[1];
  }
  {
    const NsPtr cu{ptr(n - 2).b, nullptr, (unsigned)lane}, nx{ptr(n - 1).b, nullptr, (unsigned)lane};
    ns1_phase<true, false, true, 1, NT1>(y, hacc, hprep, fuse, fbuild, xf, pool, cu, nx, k1, ik2);              // B(n - 2) | prep(n - 1)
    ns1_phase<true, false, false, 0, NT1>(y, hacc, hprep, fbuild, fuse, xf, pool, nx, nx, k1, ik2);             // B(n - 1)
  }
}

// plain layer for one tile: acc[j] = W[:, tile_j] x for three column tiles; the fragments of DP steps are in flight (DP = 2: a pool
// of twelve, refilled two steps ahead; DP = 5: all thirty requested up front -- kernels with registers to spare)
template <int DP = 2>
DEV void ns1_plain3(f32x4 (&acc)[3], const u32x4_t* const (&wt)[3], int NT, const Split8 (&xf)[KS], unsigned lane) {      // wt: wave-uniform
  static_assert(DP == 2 || DP == KS, "pool depth");
  u32x4_t pl[DP][6];
  auto frag = [&](int j, int st, int term) { return wt[j][(unsigned)(st * NT) * (2u * 64u) + (unsigned)term * 64u + lane]; };
#pragma unroll
  for (int st = 0; st < DP; ++st)
#pragma unroll
    for (int j = 0; j < 3; ++j) { pl[st][j] = frag(j, st, 1); pl[st][3 + j] = frag(j, st, 0); }
  NS_FENCE;
  static_for<0, KS>([&](auto SI) {
    constexpr int st = decltype(SI)::value;
    static_for<0, 3>([&](auto Pi) {
      constexpr int pr = decltype(Pi)::value;
      static_for<0, 3>([&](auto J) {
        constexpr int j = decltype(J)::value;
        acc[j] = ns_mfma(pl[st % DP][(pr == 0 ? 0 : 3) + j], pr == 1 ? xf[st].t[1] : xf[st].t[0], acc[j]);
        if constexpr (st + DP < KS && j == 2) {
          if constexpr (pr == 0) { pl[st % DP][0] = frag(0, st + DP, 1); pl[st % DP][1] = frag(1, st + DP, 1); pl[st % DP][2] = frag(2, st + DP, 1); }
          if constexpr (pr == 2) { pl[st % DP][3] = frag(0, st + DP, 0); pl[st % DP][4] = frag(1, st + DP, 0); pl[st % DP][5] = frag(2, st + DP, 0); }
        }
        NS_FENCE;
      });
    });
  });
}

// the rows of this workgroup's tile: frame f0 + t of utterance b (T > 0: tiles never cross an utterance), or token 16 blockIdx.x + t
struct Ns1Tile { int tok; bool live; size_t row; };
DEV Ns1Tile ns1_tile(int M, int T, int t) {
  Ns1Tile r;
  if (T > 0) {
    const int f = blockIdx.x * 16 + t;
    r.live = f < T;
    r.tok = blockIdx.y * T + min(f, T - 1);
  } else {
    r.tok = blockIdx.x * 16 + t;
    r.live = r.tok < M;
    r.tok = min(r.tok, M - 1);
  }
  r.row = (size_t)r.tok * D;
  return r;
}
DEV void ns1_split_rows(Split8 (&xf)[KS], const f32x4 (&xs)[KB], int g4, float sx) {       // fused_pp.hip: split_operand
  const f32x4 s4 = splat4(sx);
#pragma unroll
  for (int s = 0; s < KS - 1; ++s) xf[s] = split8(xs[2 * s] * s4, xs[2 * s + 1] * s4);
  f32x4 oh = splat4(0.f);
  oh.x = g4 == 0 ? sx : 0.0f;
  xf[KS - 1] = split8(xs[KB - 1] * s4, oh);
}
DEV void ns1_ln(f32x4 (&xs)[KB], const float* ga, const float* be, int g4, float eps) {      // parameters from the LDS stash
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * lds4(ga, kb, g4) + lds4(be, kb, g4);
}
// up to twelve 144-float vectors into L.par (null pointers are skipped); the caller's next __syncthreads() publishes them
template <int NW, int N>
DEV void ns1_stash(Ns1Lds<NW>& L, const float* const (&src)[N]) {
  static_assert(N <= 12, "L.par");
  for (int i = threadIdx.x; i < N * D; i += NW * 64) {
    const int v = i / D;
    if (src[v]) L.par[v][i - v * D] = src[v][i - v * D];
  }
}

// ff_module_1 + q / k / v of the tile in xs (x0 rows, identical in every wave): stores x1 (wave 0) and qkv (each wave its tiles)
template <int NW>
DEV void ns1_ff1_qkv(const Ff1QkvArgs& a, Ns1Lds<NW>& L, f32x4 (&xs)[KB], const Ns1Tile& tl, int w, int lane, int g4, u32x4_t (&pool)[NP1]) {
  f32x4 y[KB];
  const float inv_fc = 1.0f / a.fc;
  Split8 xf[KS];
  {
    f32x4 r[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) r[i] = lds4(L.par[2], i, g4) + splat4(inv_fc) * xs[i];
    ns1_ln(xs, L.par[0], L.par[1], g4, a.eps);
    const NsTok tk = ns_chain_scales(a.pp_sc, ns_row_max(xs));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = w == 0 ? r[i] * splat4(tk.s2) : splat4(0.f);       // residual + bias ride in wave 0's partial
    ns1_split_rows(xf, xs, g4, tk.sx);
    ns1_chain<18, NW>(y, xf, pool, reinterpret_cast<const u32x4_t*>(a.ns_w1), reinterpret_cast<const u32x4_t*>(a.ns_w2), w, lane, tk.k1, tk.ik2);
    ns1_allreduce<NW>(y, L, w, lane);
#pragma unroll
    for (int i = 0; i < KB; ++i) { y[i] = splat4(a.fc * tk.inv2) * y[i]; xs[i] = y[i]; }      // x1 = x0 + fc (ffn + b2)
  }
  if (w == 0 && tl.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x1 + tl.row + 16 * i + g4, y[i]);
  }
  ns1_ln(xs, L.par[3], L.par[4], g4, a.eps);
  const float sx = pp_pow2_scale(ns_row_max(xs));
  ns1_split_rows(xf, xs, g4, sx);
  const float invq = pp_recip_pow2(a.pp_sw_qkv * sx);
  const u32x4_t* wq = reinterpret_cast<const u32x4_t*>(a.ns_qkv);
#pragma unroll 1
  for (int t0 = w; t0 < 27; t0 += 3 * NW) {           // column tiles t0, t0 + NW, t0 + 2 NW (past the end: tile 26 again, not stored)
    const u32x4_t* const wt[3] = {wq + (size_t)min(t0, 26) * 128, wq + (size_t)min(t0 + NW, 26) * 128, wq + (size_t)min(t0 + 2 * NW, 26) * 128};
    f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
    ns1_plain3(acc, wt, 27, xf, (unsigned)lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tile = t0 + j * NW;
      if (tile >= 27 || !tl.live) continue;
      const int q = tile / KB, i = tile - KB * q;
      const f32x4 v = acc[j] * splat4((q == 0 ? a.qscale : 1.0f) * invq);
      if (a.qkv_T > 0) {
        const int bq = tl.tok / a.qkv_T, tq = tl.tok - bq * a.qkv_T;
        float* plane = a.qkv + (size_t)q * a.M * D + ((size_t)bq * a.qkv_H * a.qkv_T + tq) * 36;
        const int f0 = 16 * i + g4, hq = f0 / 36;
        stg4(plane + (size_t)hq * a.qkv_T * 36 + (f0 - 36 * hq), v);
      } else {
        stg4(a.qkv + (size_t)tl.tok * (3 * D) + 16 * tile + g4, v);
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void ns1_ff1_qkv_kernel(Ff1QkvArgs a) {
  __shared__ __attribute__((aligned(16))) Ns1Lds<NW> L;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, t = lane & 15;
  const Ns1Tile tl = ns1_tile(a.M, 0, t);
  f32x4 xs[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x0 + tl.row + 16 * kb + g4);
  u32x4_t pool[NP1];
  ns1_prime<18, NW>(pool, reinterpret_cast<const u32x4_t*>(a.ns_w1), w, lane);
  const float* const par[5] = {a.ff_ln_g, a.ff_ln_b, a.ff_b2, a.att_ln_g, a.att_ln_b};
  ns1_stash<NW>(L, par);
  __syncthreads();
  ns1_ff1_qkv<NW>(a, L, xs, tl, w, lane, g4, pool);
}

// ---- attention of one 16-query tile (round 6: inside the out-projection launch of the small-batch path) ------------------------------
// multihead_attention.py:151-188 on attention_split_kernel's arithmetic (two fp16 terms; S^T = K Q^T, dims 32..35 on the fp32 MFMA;
// exp2 with log2 e folded into q; the score accumulators of two key tiles are the B operand of a 32-key step of V^T P^T), T <= 256.
// Head by head: all eight waves stage the head's K / V fragments (the next head's rows are already in registers), wave w takes key
// tiles 2 w, 2 w + 1 = step w of P V for the 16 queries, and the eight partial (max, sum, output) triples are combined through LDS
// into the tile's context rows.  A separate attention launch costs 16 us for one utterance (4 workgroups, ~11 us of launch floor).
constexpr int A_HS = 36, A_OT = 3;
struct Ns1AttnLds {
  float pm[2][8][16], pl[2][8][16]; // partial maxima / sums of the eight waves, per query, for two heads
  f32x4 po[2][8][A_OT][64];         // partial outputs (48 KB)
  float ctx[16][D];                 // the tile's context rows, all heads (9 KB)
};
template <int NW>
DEV void ns1_attention(Ns1AttnLds& A, const OutGluArgs& g, int b, int f0, int w, int lane, f32x4 (&xs)[KB]) {
  static_assert(NW == 8, "eight key-tile pairs = eight waves");
  // Nothing is staged: wave w is the ONLY reader of key tiles 2 w, 2 w + 1 (as the A operand of S^T = K Q^T: lane (kg, kc) = key
  // kc, dims 8 kg ..) and of step w of V (lane (kg, fc) = feature fc of the eight keys 32 w + 4 kg + {0..3}, + 16), so every lane
  // loads exactly its own fragment entries from q / k / v and splits them in registers.  Heads go two at a time: the loads of
  // both are in flight before the first product (a first version staged K / V head by head through LDS like attention_split_kernel
  // and was bound by four serial load latencies: no faster than the separate launch).
  const int gq = lane >> 4, g4 = gq * 4, c = lane & 15;
  const int T = g.a_T, H = g.a_H, ld = g.a_ldk;
  const float sq = g.a_sq, sk = g.a_sk, sv = g.a_sv;
  constexpr float SP = 16384.f, LOG2E = 1.4426950408889634f;
  const unsigned uld = (unsigned)ld;
  const int tq = min(f0 + c, T - 1);
  const int nkt = (T + 15) / 16;
  const bool have = 2 * w < nkt;                         // this wave's pair holds at least one real key
  const f32x4 inv_qk = splat4(1.0f / (sq * sk));
#pragma unroll 1
  for (int h0 = 0; h0 < H; h0 += 2) {
    f32x4 qlo[2], qhi[2], klo[2][2], khi[2][2];
    float qtl[2], ktl[2][2], ve[2][A_OT][8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = min(h0 + hh, H - 1);
      const float* qrow = g.a_head_major ? g.aq + (((size_t)b * H + h) * T + tq) * A_HS : g.aq + ((size_t)b * T + tq) * g.a_ldq + h * A_HS;
      qlo[hh] = ldg4(qrow + 8 * gq); qhi[hh] = ldg4(qrow + 8 * gq + 4); qtl[hh] = qrow[32 + gq];
      const size_t khead = g.a_head_major ? ((size_t)b * H + h) * T * A_HS : (size_t)b * T * ld + h * A_HS;
      const float* __restrict__ kbase = g.ak + khead;
      const float* __restrict__ vbase = g.av + khead;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int skey = 16 * (2 * w + k2) + c;
        const unsigned ko = (unsigned)min(skey, T - 1) * uld;
        const bool ok = skey < T;
        klo[hh][k2] = ok ? *reinterpret_cast<const f32x4*>(kbase + ko + 8u * gq) : splat4(0.f);
        khi[hh][k2] = ok ? *reinterpret_cast<const f32x4*>(kbase + ko + 8u * gq + 4u) : splat4(0.f);
        ktl[hh][k2] = ok ? kbase[ko + 32u + gq] : 0.f;
      }
#pragma unroll
      for (int ot = 0; ot < A_OT; ++ot) {
        const int f = 16 * ot + c, key0 = 32 * w + 4 * gq;
        const unsigned vo = (unsigned)key0 * uld + (unsigned)f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int dk = 16 * (j >> 2) + (j & 3);
          ve[hh][ot][j] = (f < A_HS && key0 + dk < T) ? vbase[vo + (unsigned)dk * uld] : 0.f;      // padding features / keys: exact zeros
        }
      }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x4 sc[2], o[A_OT] = {splat4(0.f), splat4(0.f), splat4(0.f)};
      float mw = -INFINITY, lw = 0.f;
      if (have) {
        const Split8 qf = split8(qlo[hh] * splat4(LOG2E * sq), qhi[hh] * splat4(LOG2E * sq));
        const float qt = qtl[hh] * (LOG2E * sq);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int kt = 2 * w + k2;
          Split8 kf;
          if constexpr ((NS_DIAG & 64) != 0) { kf.t[0] = __builtin_bit_cast(u32x4_t, klo[hh][k2]); kf.t[1] = __builtin_bit_cast(u32x4_t, khi[hh][k2]); }   // timing only
          else kf = split8(klo[hh][k2] * splat4(sk), khi[hh][k2] * splat4(sk));
          f32x4 acc = splat4(0.f);
          acc = ns_mfma(kf.t[1], qf.t[0], acc);
          acc = ns_mfma(kf.t[0], qf.t[1], acc);
          acc = ns_mfma(kf.t[0], qf.t[0], acc);
          sc[k2] = mfma4(ktl[hh][k2] * sk, qt, acc) * inv_qk;
          const int kb = 16 * kt + g4;
          sc[k2].x = (kb + 0 < T) ? sc[k2].x : -INFINITY;
          sc[k2].y = (kb + 1 < T) ? sc[k2].y : -INFINITY;
          sc[k2].z = (kb + 2 < T) ? sc[k2].z : -INFINITY;
          sc[k2].w = (kb + 3 < T) ? sc[k2].w : -INFINITY;
          mw = fmaxf(mw, fmaxf(fmaxf(sc[k2].x, sc[k2].y), fmaxf(sc[k2].z, sc[k2].w)));
        }
        mw = group_max(mw);                            // finite: key tile 2 w holds at least one real key
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          sc[k2].x = __builtin_amdgcn_exp2f(sc[k2].x - mw);
          sc[k2].y = __builtin_amdgcn_exp2f(sc[k2].y - mw);
          sc[k2].z = __builtin_amdgcn_exp2f(sc[k2].z - mw);
          sc[k2].w = __builtin_amdgcn_exp2f(sc[k2].w - mw);
          lw += (sc[k2].x + sc[k2].y) + (sc[k2].z + sc[k2].w);
        }
        lw = group_sum(lw);
        const Split8 pf = split8(sc[0] * splat4(SP), sc[1] * splat4(SP));
#pragma unroll
        for (int i = 0; i < A_OT; ++i) {
          const f32x4 lo = {ve[hh][i][0], ve[hh][i][1], ve[hh][i][2], ve[hh][i][3]}, hi = {ve[hh][i][4], ve[hh][i][5], ve[hh][i][6], ve[hh][i][7]};
          Split8 vf;
          if constexpr ((NS_DIAG & 64) != 0) { vf.t[0] = __builtin_bit_cast(u32x4_t, lo); vf.t[1] = __builtin_bit_cast(u32x4_t, hi); }   // timing only
          else vf = split8(lo * splat4(sv), hi * splat4(sv));
          o[i] = ns_mfma(vf.t[1], pf.t[0], o[i]);
          o[i] = ns_mfma(vf.t[0], pf.t[1], o[i]);
          o[i] = ns_mfma(vf.t[0], pf.t[0], o[i]);
        }
      }
      if (gq == 0) { A.pm[hh][w][c] = mw; A.pl[hh][w][c] = lw; }
#pragma unroll
      for (int i = 0; i < A_OT; ++i) A.po[hh][w][i][lane] = o[i];
    }
    __syncthreads();                                   // the partials of both heads are written
    if (w < 2 && h0 + w < H) {                         // wave 0 / 1 combines head h0 / h0 + 1
      const int h = h0 + w;
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) mx = fmaxf(mx, A.pm[w][k][c]);
      float den = 0.f;
      f32x4 acc[A_OT] = {splat4(0.f), splat4(0.f), splat4(0.f)};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float sc_k = __builtin_amdgcn_exp2f(A.pm[w][k][c] - mx);          // waves without keys: exp2(-inf) = 0
        den += A.pl[w][k][c] * sc_k;
#pragma unroll
        for (int i = 0; i < A_OT; ++i) acc[i] += A.po[w][k][i][lane] * splat4(sc_k);
      }
      const float inv = (1.0f / den) * (1.0f / (SP * sv));
#pragma unroll
      for (int i = 0; i < A_OT; ++i)
        if (16 * i + g4 < A_HS) *reinterpret_cast<f32x4*>(&A.ctx[c][h * A_HS + 16 * i + g4]) = acc[i] * splat4(inv);
    }
    __syncthreads();                                   // combined: the partial areas are free, the context rows of these heads written
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = *reinterpret_cast<const f32x4*>(&A.ctx[c][16 * kb + g4]);
  __syncthreads();                                     // ... and read: the area is reused by the exchange below
}

// [attention +] out-projection + residual + LayerNorm + pw_conv_1 + GLU (pp_out_glu_kernel, one tile per workgroup): ctx (or q, k,
// v), x1 -> x2, u.  Grid (ceil(T / 16), utterances).
template <int NW>
union Ns1OgLds {
  Ns1Lds<NW> l;
  Ns1AttnLds a;
};
template <int NW, bool ATTN>
__global__ __launch_bounds__(NW * 64) void ns1_out_glu_kernel(OutGluArgs a, int T) {
  __shared__ __attribute__((aligned(16))) Ns1OgLds<NW> U;
  Ns1Lds<NW>& L = U.l;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, t = lane & 15;
  const Ns1Tile tl = ns1_tile(a.M, T, t);
  f32x4 xs[KB], x2[KB];
  if constexpr (ATTN) ns1_attention<NW>(U.a, a, blockIdx.y, blockIdx.x * 16, w, lane, xs);
  else {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.ctx + tl.row + 16 * kb + g4);
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) x2[kb] = ldg4(a.x1 + tl.row + 16 * kb + g4);
  {
    const float* const par[2] = {a.cv_ln_g, a.cv_ln_b};
    ns1_stash<NW>(L, par);                             // published by the barrier in front of the out projection's exchange
  }
  Split8 xf[KS];
  {
    const float sx = pp_pow2_scale(ns_row_max(xs));
    ns1_split_rows(xf, xs, g4, sx);
    const f32x4 inv = splat4(pp_recip_pow2(a.pp_sw_out * sx));
    const u32x4_t* wo = reinterpret_cast<const u32x4_t*>(a.ns_out);
    // nine column tiles: wave w takes tiles w, w + NW, w + 2 NW (< 9); every wave needs the whole row afterwards
#pragma unroll
    for (int i = 0; i < KB; ++i) L.red[0][i][lane] = splat4(0.f);        // (all waves write the same zeros: tiles nobody owns do not exist)
    __syncthreads();
    if (w < KB) {
      const u32x4_t* const wt[3] = {wo + (size_t)min(w, 8) * 128, wo + (size_t)min(w + NW, 8) * 128, wo + (size_t)min(w + 2 * NW, 8) * 128};
      f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
      ns1_plain3(acc, wt, 9, xf, (unsigned)lane);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (w + j * NW < KB) L.red[0][w + j * NW][lane] = acc[j] * inv;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KB; ++i) { x2[i] += L.red[0][i][lane]; xs[i] = x2[i]; }        // x2 = x1 + attention (+ bias: row 144)
  }
  if (w == 0 && tl.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x2 + tl.row + 16 * i + g4, x2[i]);
  }
  ns1_ln(xs, L.par[0], L.par[1], g4, a.eps);
  const float sx = pp_pow2_scale(ns_row_max(xs));
  ns1_split_rows(xf, xs, g4, sx);
  const float inv = pp_recip_pow2(a.pp_sw_pw1 * sx);
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(a.ns_pw1);
  // value tile i and gate tile 9 + i in the same wave; wave w takes i = w, w + NW (< 9): one ns1_plain3 holds (value, gate, -)
#pragma unroll 1
  for (int i = w; i < KB; i += NW) {
    const u32x4_t* const wt[3] = {wp + (size_t)i * 128, wp + (size_t)(KB + i) * 128, wp + (size_t)i * 128};
    f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
    ns1_plain3(acc, wt, 18, xf, (unsigned)lane);
    if (tl.live) {
      const f32x4 va = acc[0] * splat4(inv), vb = acc[1] * splat4(inv);
      const f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
      stg4(a.u + tl.row + 16 * i + g4, o);
    }
  }
}

// depthwise conv (k = 32) of the tile's 16 frames from u in HBM: 47 window rows + the taps through LDS, 16 x 144 outputs by the
// workgroup's threads, back through LDS in row order; every wave picks up all rows.  conformer_blocks.py:205 ('same': 15 / 16
// zeros) and chunk_conformer_blocks.py:262 ('causal': 31 in front): rows outside the utterance are the padding.
constexpr int N1_K = 32, N1_ROWS = 16 + N1_K - 1;
struct Ns1DwLds { float win[N1_ROWS][D]; float wt[N1_K][D]; float out[16][D]; };
template <int NW>
DEV void ns1_dwconv(Ns1DwLds& S, const TailFf2Args& a, f32x4 (&xs)[KB], int lane, int g4, int t) {
  constexpr int NT_ = NW * 64;
  const int tid = threadIdx.x;
  const int T = a.dw_T, f0 = blockIdx.x * 16 - a.dw_pad;
  const float* __restrict__ ub = a.dw_u + (size_t)blockIdx.y * T * D;
  // every global load first, then the LDS writes (one memory latency for the window and the taps, not one per loop iteration)
  constexpr int NWIN = N1_ROWS * (D / 4), NTAP = N1_K * (D / 4), RW = (NWIN + NT_ - 1) / NT_, RT_ = (NTAP + NT_ - 1) / NT_;
  f32x4 sw[RW], st[RT_];
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int i = tid + k * NT_, r = i / (D / 4), c4 = i - r * (D / 4), f = f0 + r;
    sw[k] = (i < NWIN && f >= 0 && f < T) ? ldg4(ub + (size_t)f * D + 4 * c4) : splat4(0.f);
  }
#pragma unroll
  for (int k = 0; k < RT_; ++k) {
    const int i = tid + k * NT_;
    st[k] = i < NTAP ? ldg4(a.dw_wd + 4 * i) : splat4(0.f);
  }
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int i = tid + k * NT_;
    if (i < NWIN) *reinterpret_cast<f32x4*>(&S.win[0][0] + 4 * i) = sw[k];
  }
#pragma unroll
  for (int k = 0; k < RT_; ++k) {
    const int i = tid + k * NT_;
    if (i < NTAP) *reinterpret_cast<f32x4*>(&S.wt[0][0] + 4 * i) = st[k];
  }
  __syncthreads();
  for (int i = tid; i < 16 * (D / 4); i += NT_) {
    const int fr = i / (D / 4), c4 = i - fr * (D / 4);
    f32x4 acc = splat4(0.f);
#pragma unroll 8
    for (int j = 0; j < N1_K; ++j)
      acc += *reinterpret_cast<const f32x4*>(&S.win[fr + j][4 * c4]) * *reinterpret_cast<const f32x4*>(&S.wt[j][4 * c4]);
    *reinterpret_cast<f32x4*>(&S.out[fr][4 * c4]) = acc;
  }
  __syncthreads();
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = *reinterpret_cast<const f32x4*>(&S.out[t][16 * kb + g4]);
}

// depthwise conv + conv-module tail + ff_module_2 + block LayerNorm [+ ff_module_1 + qkv of the next block]
// (pp_block_kernel<true, FF1, 0, true>: u and x2 from HBM), one tile per workgroup; grid (ceil(T / 16), utterances)
template <int NW, bool FF1>
__global__ __launch_bounds__(NW * 64) void ns1_tail_kernel(TailFf2Args a, Ff1QkvArgs b) {
  __shared__ __attribute__((aligned(16))) Ns1Lds<NW> L;
  __shared__ __attribute__((aligned(16))) Ns1DwLds S;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, t = lane & 15;
  unsigned long long st1[12] = {};
  NS1_STAMP(0);
  const Ns1Tile tl = ns1_tile(a.M, a.dw_T, t);
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + tl.row + 16 * kb + g4);
  {
    const float* const par[11] = {FF1 ? b.ff_ln_g : nullptr, FF1 ? b.ff_ln_b : nullptr, FF1 ? b.ff_b2 : nullptr, FF1 ? b.att_ln_g : nullptr,
                                  FF1 ? b.att_ln_b : nullptr, a.pw2_b, a.ff_ln_g, a.ff_ln_b, a.ff_b2, a.ln_g, a.ln_b};
    ns1_stash<NW>(L, par);                             // published by the barriers of the depthwise conv
  }
  u32x4_t pool[NP1], pool2[NP1];
  ns1_prime<9, NW>(pool, reinterpret_cast<const u32x4_t*>(a.ns_cv_w1), w, lane);            // in flight under the depthwise conv
  ns1_dwconv<NW>(S, a, xs, lane, g4, t);
  NS1_STAMP(1);
  Split8 xf[KS];
  {
    const NsTok tk = ns_chain_scales(a.pp_sc[0], ns_row_max(xs));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = w == 0 ? (y[i] + lds4(L.par[5], i, g4)) * splat4(tk.s2) : splat4(0.f);
    ns1_split_rows(xf, xs, g4, tk.sx);
    NS1_STAMP(2);
    ns1_chain<9, NW>(y, xf, pool, reinterpret_cast<const u32x4_t*>(a.ns_cv_w1), reinterpret_cast<const u32x4_t*>(a.ns_cv_w2), w, lane, tk.k1, tk.ik2);
    ns1_prime<18, NW>(pool2, reinterpret_cast<const u32x4_t*>(a.ns_ff_w1), w, lane);         // ff_module_2's first fragments: under the exchange + LayerNorm
    NS1_STAMP(3);
    ns1_allreduce<NW>(y, L, w, lane);
    NS1_STAMP(4);
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = y[i] * splat4(tk.inv2);                             // x3 = x2 + conv module
  }
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = y[i];
  ns1_ln(xs, L.par[6], L.par[7], g4, a.eps);
  {
    const NsTok tk = ns_chain_scales(a.pp_sc[1], ns_row_max(xs));
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = w == 0 ? (lds4(L.par[8], i, g4) + splat4(inv_fc) * y[i]) * splat4(tk.s2) : splat4(0.f);
    ns1_split_rows(xf, xs, g4, tk.sx);
    NS1_STAMP(5);
    ns1_chain<18, NW>(y, xf, pool2, reinterpret_cast<const u32x4_t*>(a.ns_ff_w1), reinterpret_cast<const u32x4_t*>(a.ns_ff_w2), w, lane, tk.k1, tk.ik2);
    if constexpr (FF1) ns1_prime<18, NW>(pool, reinterpret_cast<const u32x4_t*>(b.ns_w1), w, lane);      // the next block's ff_module_1
    NS1_STAMP(6);
    ns1_allreduce<NW>(y, L, w, lane);
    NS1_STAMP(7);
#pragma unroll
    for (int i = 0; i < KB; ++i) y[i] = splat4(a.fc * tk.inv2) * y[i];
  }
  ns1_ln(y, L.par[9], L.par[10], g4, a.eps);                                                // block-final LayerNorm
  if (a.y && w == 0 && tl.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.y + tl.row + 16 * i + g4, y[i]);
  }
  NS1_STAMP(8);
  if constexpr (FF1) ns1_ff1_qkv<NW>(b, L, y, tl, w, lane, g4, pool);
  NS1_STAMP(9);
  if constexpr ((NS_DIAG & 32) != 0) {
    if (blockIdx.x == 3 && blockIdx.y == 0 && lane == 0 && (w == 0 || w == 7)) {
      printf("NS1STAMP w%d:", w);
      for (int i = 1; i <= 9; ++i) printf(" %d:%llu", i, st1[i] - st1[0]);
      printf("\n");
    }
  }
}

// Subsampling Dense (conformer_blocks.py:93-96; pp_sublinear_kernel's arithmetic: every 144-wide chunk of a row under its own
// power-of-two scale, the bias in row 144 of chunk 0) for one tile: the CHUNKS are split over the waves -- wave w takes chunks w,
// w + NW, ... with all nine column tiles each, partial rows meet in LDS.  ns: plain fragments [5][9 chunks][2][64].
template <int NW>
__global__ __launch_bounds__(NW * 64) void ns1_sublinear_kernel(StreamGemmArgs a, const u32x4_t* __restrict__ ns, float sw, int chunks) {
  __shared__ __attribute__((aligned(16))) Ns1Lds<NW> L;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, t = lane & 15;
  const int tok = blockIdx.x * 16 + t;
  const bool live = tok < a.M;
  const float* __restrict__ xrow = a.x + (size_t)min(tok, a.M - 1) * a.K + g4;
  const int NT = KB * chunks;
  f32x4 y[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = splat4(0.f);
#pragma unroll 1
  for (int f = w; f < chunks; f += NW) {
    f32x4 xs[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(xrow + (size_t)f * D + 16 * kb);
    Split8 xf[KS];
    const float sx = pp_pow2_scale(ns_row_max(xs));
    ns1_split_rows(xf, xs, g4, sx);
    const f32x4 inv = splat4(pp_recip_pow2(sw * sx));
    const u32x4_t* wp = ns + (size_t)(KB * f) * 128;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const u32x4_t* const wt[3] = {wp + (size_t)(3 * g) * 128, wp + (size_t)(3 * g + 1) * 128, wp + (size_t)(3 * g + 2) * 128};
      f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
      ns1_plain3<KS>(acc, wt, NT, xf, (unsigned)lane);
#pragma unroll
      for (int j = 0; j < 3; ++j) y[3 * g + j] += acc[j] * inv;
    }
  }
  ns1_allreduce<NW>(y, L, w, lane);
  if (w == 0 && live) {
    float* yrow = a.y + (size_t)tok * a.ldy + g4;
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(yrow + 16 * i, y[i]);
  }
}

// CTC class head of one tile (pp_head_kernel's contract: logits = x W + b over `groups` x nine column tiles, per-frame arg-max --
// lowest class among equal maxima -- and / or maximum and / or the logits), column tiles w, w + NW, ... per wave
template <int NW>
__global__ __launch_bounds__(NW * 64) void ns1_head_kernel(GemmArgs a, const u32x4_t* __restrict__ ns, float sw, int groups) {
  __shared__ float bv[NW][16];
  __shared__ int bi[NW][16];
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, t = lane & 15;
  const Ns1Tile tl = ns1_tile(a.M, 0, t);
  f32x4 xs[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x + tl.row + 16 * kb + g4);
  Split8 xf[KS];
  const float sx = pp_pow2_scale(ns_row_max(xs));
  ns1_split_rows(xf, xs, g4, sx);
  const float inv = pp_recip_pow2(sw * sx);
  const int NT = KB * groups;
  const u32x4_t* wp = ns;
  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  const bool want_max = a.argmax_out != nullptr || a.maxval_out != nullptr;
  float* yrow = a.y ? a.y + (size_t)tl.tok * a.ldy : nullptr;
#pragma unroll 1
  for (int t0 = w; t0 < NT; t0 += 3 * NW) {
    const u32x4_t* const wt[3] = {wp + (size_t)min(t0, NT - 1) * 128, wp + (size_t)min(t0 + NW, NT - 1) * 128, wp + (size_t)min(t0 + 2 * NW, NT - 1) * 128};
    f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
    ns1_plain3(acc, wt, NT, xf, (unsigned)lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tile = t0 + j * NW, f0 = 16 * tile + g4;
      if (tile >= NT || 16 * tile >= a.n_valid) continue;
      const f32x4 v = acc[j] * splat4(inv);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      if (want_max) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (f0 + q < a.n_valid && vv[q] > best_v) { best_v = vv[q]; best_i = f0 + q; }
      }
      if (yrow && tl.live) {
        if (f0 + 3 < a.n_valid && (a.ldy & 3) == 0) stg4(yrow + f0, v);
        else
#pragma unroll
          for (int q = 0; q < 4; ++q) if (f0 + q < a.n_valid) yrow[f0 + q] = vv[q];
      }
    }
  }
  if (!want_max) return;
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float ov = __shfl_xor(best_v, off);
    const int oi = __shfl_xor(best_i, off);
    if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
  }
  if (lane < 16) { bv[w][lane] = best_v; bi[w][lane] = best_i; }
  __syncthreads();
  if (w == 0 && lane < 16 && tl.live) {
#pragma unroll
    for (int k = 1; k < NW; ++k) {
      const float ov = bv[k][lane];
      const int oi = bi[k][lane];
      if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
    }
    if (a.argmax_out) a.argmax_out[tl.tok] = best_i;
    if (a.maxval_out) a.maxval_out[tl.tok] = best_v;
  }
}

}  // namespace

// ---- small batches: one tile per workgroup ----------------------------------------------------------------------------------------------
namespace {
constexpr int NS1_W = 8;       // waves per workgroup (two per SIMD: one wave's fragment loads issue while the other's MFMAs run)
int ns1_max_rows() {
  // MI355ASR_NS1_MAX_M=n: the one-tile-per-workgroup kernels up to n rows (0: never).  Default 4096: 256 tiles = one workgroup per CU
  // (measured at 250 ... 4000 rows: 46 - 50 us per block against 60 - 62; beyond, the workgroups of a launch no longer run at once)
  static const int n = (int)mi355_env("MI355ASR_NS1_MAX_M", 4096);
  return n;
}
}  // namespace
bool ns1_rows_ok(int M) { return M > 0 && M <= ns1_max_rows(); }
int launch_ns1_ff1_qkv(const Ff1QkvArgs& b, hipStream_t s) {
  if (!ns1_rows_ok(b.M) || !b.ns_w1 || !b.ns_w2 || !b.ns_qkv || b.pre_pp || b.xq_pe) return -1;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL((ns1_ff1_qkv_kernel<NS1_W>), dim3((b.M + 15) / 16), dim3(NS1_W * 64), 0, s, b);
  return 0;
}
// what launch_pp_og_tail_ff1 / _ff2 do in one launch, as two: [attention +] out-projection + GLU (x2 -> g.x2, u -> g.u), then
// depthwise conv + tail [+ the next block's ff_module_1 + qkv].  a.dw_u is set to g.u here; a.x2 must be g.x2.
bool ns1_block_ok(const TailFf2Args& a, const Ff1QkvArgs* b, const OutGluArgs& g) {
  if (!ns1_rows_ok(a.M) || a.M != g.M || !g.ns_out || !g.ns_pw1 || !a.ns_cv_w1 || !a.ns_cv_w2 || !a.ns_ff_w1 || !a.ns_ff_w2 || a.head_pp ||
      !g.x2 || !g.u || !g.x1 || !a.dw_wd || a.dw_T <= 0 || a.M % a.dw_T != 0 || (a.dw_pad != 15 && a.dw_pad != 31))
    return false;
  if (b && (!b->ns_w1 || !b->ns_w2 || !b->ns_qkv || b->pre_pp || b->xq_pe || b->M != a.M)) return false;
  return true;
}
bool ns1_attn_ok(int hs, const AttnArgs& at) {
  // MI355ASR_NS1_ATTN=0: the attention of a small-batch block as its own launch (attention_split_kernel)
  static const bool on = mi355_env("MI355ASR_NS1_ATTN", 1) != 0;
  return on && hs == A_HS && at.Tq == at.Tk && at.Tk > 16 && at.Tk <= 256 && at.win_front < 0 && at.H >= 1 && at.H * A_HS == at.D && at.D == D &&
         at.h2_sq > 0.f && at.h2_sk > 0.f && at.h2_sv > 0.f && at.ldq % 4 == 0 && at.ldk % 4 == 0 && at.q_off == 0;
}
int launch_ns1_og_tail(const TailFf2Args& a, const Ff1QkvArgs* b, const OutGluArgs& g, hipStream_t s) {
  if (!ns1_block_ok(a, b, g) || (!g.attn && !g.ctx)) return -1;
  if (g.attn && (!g.aq || !g.ak || !g.av || g.a_T != a.dw_T)) return -1;
  note_scheme(SCHEME_F16X2);
  const dim3 grid((a.dw_T + 15) / 16, a.M / a.dw_T);
  if (g.attn) hipLaunchKernelGGL((ns1_out_glu_kernel<NS1_W, true>), grid, dim3(NS1_W * 64), 0, s, g, a.dw_T);
  else hipLaunchKernelGGL((ns1_out_glu_kernel<NS1_W, false>), grid, dim3(NS1_W * 64), 0, s, g, a.dw_T);
  TailFf2Args k = a;
  k.dw_u = g.u; k.x2 = g.x2;
  if (b) hipLaunchKernelGGL((ns1_tail_kernel<NS1_W, true>), grid, dim3(NS1_W * 64), 0, s, k, *b);
  else hipLaunchKernelGGL((ns1_tail_kernel<NS1_W, false>), grid, dim3(NS1_W * 64), 0, s, k, Ff1QkvArgs{});
  return 0;
}
int launch_ns1_head(const GemmArgs& a, const float* ns, float sw, int groups, hipStream_t s) {
  if (!ns1_rows_ok(a.M) || !ns || groups < 1 || a.n_valid > 144 * groups) return -1;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL((ns1_head_kernel<NS1_W>), dim3((a.M + 15) / 16), dim3(NS1_W * 64), 0, s, a, reinterpret_cast<const u32x4_t*>(ns), sw, groups);
  return 0;
}
int launch_ns1_sublinear(const StreamGemmArgs& a, const float* ns, float sw, hipStream_t s) {
  if (!ns1_rows_ok(a.M) || !ns || a.NT != KB || a.K % D != 0 || a.K < D || (a.ldy & 3) != 0) return -1;
  note_scheme(SCHEME_F16X2);
  hipLaunchKernelGGL((ns1_sublinear_kernel<NS1_W>), dim3((a.M + 15) / 16), dim3(NS1_W * 64), 0, s, a, reinterpret_cast<const u32x4_t*>(ns), sw, a.K / D);
  return 0;
}
