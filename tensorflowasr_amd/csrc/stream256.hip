// The whole block stack of the streaming encoder in ONE launch (bf16 mode, dmodel 256; BASELINE config 3, round 5).
// Reference: StreamingConformerEncoder runs every chunk through its ConformerBlocks on its own (conformer_blocks.py:574-594 reshapes
// [B, T, d] into [B * chunks, T / chunks, d]; blocks :240-262, FFModule :99-134, MHSAModule :137-172, ConvModule :182-218): a
// chunk's 13 frames never see another chunk's, so the stack is a per-chunk chain of ~40 tiny layers.
//
// Why: at 64 chunks x 13 rows a layer is one 16-row tile per chunk; round 4's path ran 32 launches of 5-17 us per step for the four
// blocks (0.39 ms of config 3's 0.90), each of them a dispatch + a dependent chain of memory round trips, none of them arithmetic.
// Here ONE workgroup of eight waves owns ONE chunk (<= 16 rows) for all blocks:
//   * the residual stream lives in registers: wave w holds features [32 w, 32 w + 32) of the 16 tokens (two accumulator-layout
//     tiles, f32);
//   * every GEMM is Y^T = W^T X^T on v_mfma_f32_16x16x32_bf16 with the weights streamed straight from L2 as 1 KB fragments of the
//     one-term slab-ring packs (gemm_ring.hip; what chain256_bf16_kernel reads), in batches of eight fragments, three batches ahead
//     of their MFMAs -- the stream never stops: the last batches of a layer request the first of the next (and of the next block);
//     barriers are bare s_barrier after s_waitcnt lgkmcnt(0), so the weight loads stay in flight across them;
//   * activations cross waves through LDS as bf16 operand fragments: the accumulator tiles of a producer ARE the operand of the
//     consumer (normalised rows, FFN hidden, attention context, depthwise output, conv hidden): two adjacent 16-feature tiles of a
//     wave = one 32-wide k-step; as f32 rows (strides 260 / 772 floats: conflict-free 16-byte accesses) only where tokens mix
//     (q / k / v for the attention, the GLU output for the depthwise conv);
//   * LayerNorm statistics: every wave takes mean and centred second moment of its own 32 features per token (two passes over
//     registers), the eight partial pairs meet in LDS (2 KB, one barrier) and are merged with the pairwise update
//     M2 = sum M2_w + 32 sum (mean_w - mean)^2 -- as stable as two passes over the row, 1 / 8 of the LDS traffic of reading it
//     (the first version read the whole row in every wave: 1.1 us per LayerNorm, six per block);
//   * attention over the chunk's <= 16 keys in f32 on the fp32 MFMA (wave = head x output half: S^T = K Q^T as 16 MFMAs, softmax
//     over the accumulator's four keys x four lane groups, O^T = V^T P^T with the accumulator as the B operand), depthwise conv on the
//     VALU from the f32 rows.
// Same arithmetic as the layer-at-a-time bf16 path: GEMM operands rounded to nearest-even bf16, f32 accumulation, f32 LayerNorm (two
// passes) / softmax / GLU / swish / BN / residuals; the summation order along K differs from gemm16_kernel's four-way K split, so a
// hidden value on a bf16 rounding boundary may fall the other way (the bound tests/ hold the bf16 mode to).
// What bounds it: a workgroup pulls a block's 3.4 MB of bf16 weights through one CU's L2 port (~145 GB/s with 16-byte loads,
// tools/ubench/l2_pull.hip): ~24 us per block; 64 workgroups use a quarter of the chip -- the other CUs have nothing to do at 832 rows.
// Measured (profiles/r05_stream256.md): 156 us for the four blocks (the 32 launches it replaces: 389 us by events) = 95 us of weight
// stream + 61 us that the same kernel takes WITHOUT its weight loads (16 barriers per block 14, swish / sigmoid 6, LayerNorm
// statistics + attention 10, the dependent LDS-write -> barrier -> LDS-read -> MFMA chains of 64 phases the rest): the two do not
// overlap, and neither a third batch in flight nor requests held back until after a layer's epilogue changes that by a
// microsecond -- whatever a wave has requested is delivered at the port's rate, and during a phase no wave requests anything.
// Dedicated loader waves with an LDS ring are what would keep the port busy through the phases; the ring would need ~100 KB.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "launch.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

#ifndef MI355ASR_S256_DIAG
#define MI355ASR_S256_DIAG 0
#endif
// timing experiments (tools/build_variant.py ... -DMI355ASR_S256_DIAG=n; WRONG results): bit 0 = every block reads block 0's
// weights (3.4 MB: L2-resident), 1 = no weight loads at all, 2 = no attention, 3 = no row statistics (one barrier less per LayerNorm),
// 4 = no s_barrier (the LDS waits stay), 5 = no swish / sigmoid, 6 = no MFMAs (one VALU op per batch instead; operand reads stay)
constexpr int S_DG = MI355ASR_S256_DIAG;
constexpr int S_NW = 8;          // waves per workgroup
constexpr int S_D = 256;
constexpr int S_RSF = 260;       // f32 row stride (floats): 260 = 4 (mod 64) -> sixteen rows' 16-byte pieces tile the 64 banks
constexpr int S_QSF = 772;       // q | k | v row stride
constexpr int S_NBUF = 3;        // weight batches (eight 1 KB fragments per wave) in flight / being multiplied

DEV unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2_t)); }   // v_cvt_pk_bf16_f32: nearest even
DEV u32x2_t bf16x4(f32x4 v) { return u32x2_t{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)}; }
// one 32-wide k-step of a token's operand: features 32 s + 4 g + {0..3} and 32 s + 16 + 4 g + {0..3} (the slot order of the ring packs)
DEV u32x4_t operand(f32x4 lo, f32x4 hi) {
  const u32x2_t a = bf16x4(lo), b = bf16x4(hi);
  return u32x4_t{a.x, a.y, b.x, b.y};
}
DEV f32x4 mma(u32x4_t w, u32x4_t x, f32x4 c) {
  if constexpr (S_DG & 64) { c.x += __builtin_bit_cast(float, w.x ^ x.x); return c; }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
DEV f32x4 act_swish(f32x4 v) { if constexpr (S_DG & 32) return v; else return swish4(v); }
DEV float act_sigmoid(float v) { if constexpr (S_DG & 32) return v; else return fast_sigmoid(v); }
DEV unsigned frag_at(unsigned tile, unsigned step, unsigned steps) { return (((tile >> 3) * steps + step) * 8 + (tile & 7)) * 64; }
DEV void bar_lds() {            // this wave's LDS traffic is done; workgroup barrier; no fence: global loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!(S_DG & 16)) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct Lds {
  float rows[16][S_RSF];        // 16 KB  GLU output rows for the depthwise conv
  f32x2 stat[2][S_NW][16];      //  2 KB  per-wave (mean, M2) of a token's 32 features, two buffers taken in turn
  u32x4_t hid[32][64];          // 32 KB  hidden activations as operand fragments (FFN: 32 k-steps; conv module: 16)
  float qkv[16][S_QSF];         // 48 KB  q (scaled) | k | v of the chunk
  u32x4_t xop[8][64];           //  8 KB  the normalised rows as operand fragments
  u32x4_t ctxf[8][64];          //  8 KB  attention context as operand fragments
  u32x4_t dwf[8][64];           //  8 KB  depthwise conv output as operand fragments
};

// addressing discipline: a UNIFORM pointer (scalar registers) + one of two per-lane unsigned offsets (lane, 4 g) -- the loads take
// the scalar-base form and nothing per-lane is worth hoisting out of the block loop (a per-lane index per fragment was: 348 spills)
DEV f32x4 ldu(const float* __restrict__ p, unsigned off) { return *reinterpret_cast<const f32x4*>(p + off); }

// batch t of a layer's weight stream for this wave: G k-steps x NT tiles (NT * G <= 8 fragments of 1 KB).  wl = the layer's pack
// advanced to this wave's first fragment (uniform, changes with the block); off(i, step) = a compile-time fragment offset: a
// fragment's address is a scalar add away from wl (a per-wave run-time offset per fragment was hoisted out of the block loop by
// the hundred and spilled)
template <int NT, int G, class OFF>
DEV void wload(const u32x4_t* __restrict__ wl, int t, OFF off, unsigned lane, u32x4_t (&d)[8]) {
#pragma unroll
  for (int u = 0; u < G; ++u)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const u32x4_t* __restrict__ p = wl + off(i, t * G + u);
      if constexpr (!(S_DG & 2)) d[u * NT + i] = p[lane];
    }
}
// ring-pack geometry ([N / 128 chunks][K / 32 steps][8 tiles][64 lanes] fragments of 16 bytes) for a wave that owns NT consecutive
// tiles, NT in {2, 4, 8}: its first fragment, and fragment (tile i, step) from there
DEV unsigned wave_frag0(unsigned nt, unsigned wave, unsigned steps) { const unsigned t0 = nt * wave; return (t0 >> 3) * steps * 512 + (t0 & 7) * 64; }
struct OffPlain { constexpr unsigned operator()(int i, int step) const { return (unsigned)step * 512u + (unsigned)i * 64u; } };
// q | k | v (48 tiles): wave w owns tile w of each of the six chunks -- tiles 8 i + w: i < 2 are q tiles for every wave
struct OffQkv { constexpr unsigned operator()(int i, int step) const { return ((unsigned)i * 8u + (unsigned)step) * 512u; } };
// pw_conv_1 (GLU pack: a chunk = four value tiles + their four gate tiles): value tiles 2 w, 2 w + 1, then their gates
struct OffGlu { constexpr unsigned operator()(int i, int step) const { return (unsigned)step * 512u + (unsigned)((i & 1) + 4 * (i >> 1)) * 64u; } };

// acc[i] += W[:, tile i]^T X^T over KS k-steps.  The weight stream is one sequence of batches across layers and blocks, three
// batches ahead of the multiplications in three register buffers: batch t of this layer sits in wa[(ROT + t) % 3] (requested by
// whoever ran three batches earlier); as soon as its MFMAs have issued, the buffer takes batch t + 3 -- this layer's (ld) or, past
// its end, the following layers' (ldn(j): batch j counted from the next layer's first).  (First version: two batches ahead,
// requested BEFORE the MFMAs of batch t -- a third less in flight, and nothing but those two batches to cover the phases between
// the GEMMs: the stream's 95 us and the phases' 61 us simply added up.)
// pre() runs just before the first request that belongs to a later layer: vector-memory loads return in order (one counter), so
// the small operands of this layer's epilogue and of the next layer's prologue (biases, LayerNorm gamma / beta, depthwise taps) are
// requested THERE -- asked for where they are used they would sit behind the batches of weights in flight and every epilogue would
// wait for the whole queue (a memory round trip per layer, ~14 per block).
template <int NT, int G, int KS, int ROT, class BOP, class LD, class LDN, class PRE>
DEV void gemm(f32x4 (&acc)[NT], u32x4_t (&wa)[S_NBUF][8], BOP bop, LD ld, LDN ldn, PRE pre) {
  constexpr int NB = KS / G;
  static_assert(NT * G <= 8 && S_NBUF == 3, "batches");
#pragma unroll
  for (int t = 0; t < NB; ++t) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const u32x4_t x = bop(t * G + u);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = mma(wa[(ROT + t) % S_NBUF][u * NT + i], x, acc[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t == (NB > S_NBUF ? NB - S_NBUF : 0)) pre();
    if (t + S_NBUF < NB) ld(t + S_NBUF, wa[(ROT + t) % S_NBUF]);
    else ldn(t + S_NBUF - NB, wa[(ROT + t) % S_NBUF]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
struct LnP { f32x4 g[2], b[2]; };     // gamma / beta of this wave's 32 features

template <int KSZ>
__global__ __launch_bounds__(S_NW * 64) void stream256_kernel(S256Args a) {
  __shared__ __attribute__((aligned(16))) Lds L;
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned g4 = (lane >> 4) * 4, c = lane & 15;
  const int T = a.T;
  const float eps = a.eps;
  const size_t row0 = (size_t)blockIdx.x * T;
  auto load_ln = [&](const float* __restrict__ gamma, const float* __restrict__ beta, LnP& p) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { p.g[j] = ldu(gamma + 32 * wave + 16 * j, g4); p.b[j] = ldu(beta + 32 * wave + 16 * j, g4); }
  };
  // rows past T repeat the last one: finite values that are never keys, never conv taps, never stored
  f32x4 xr[2];
  LnP lnp;                      // the parameters of the NEXT LayerNorm, requested a layer ahead
  {
    const unsigned roff = min(c, (unsigned)(T - 1)) * S_D + g4;
    const float* __restrict__ xw = a.x + row0 * S_D + 32 * wave;
    xr[0] = ldu(xw, roff);
    xr[1] = ldu(xw + 16, roff);
    load_ln(a.blk[0].ff_ln_g[0], a.blk[0].ff_ln_b[0], lnp);
  }
  u32x4_t wa[S_NBUF][8];
  if constexpr (S_DG & 2) {
#pragma unroll
    for (int t = 0; t < S_NBUF; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) wa[t][i] = u32x4_t{lane, lane, lane, lane};
  }
  {
    const u32x4_t* w = reinterpret_cast<const u32x4_t*>(a.blk[0].ff_w1[0]) + wave_frag0(8, wave, 8);
#pragma unroll
    for (int t = 0; t < S_NBUF; ++t) wload<8, 1>(w, t, OffPlain{}, lane, wa[t]);
  }
  int par = 0;
  // LayerNorm statistics of the residual rows (Keras semantics: biased variance, eps inside the root): this wave's 32 features per
  // token in two passes over registers, the eight waves' (mean, M2) merged pairwise
  auto row_stats = [&](float& mean, float& rstd) {
    if constexpr (S_DG & 8) { mean = xr[0].x; rstd = xr[1].y; return; }
    const float mw = group_sum(((xr[0].x + xr[0].y) + (xr[0].z + xr[0].w)) + ((xr[1].x + xr[1].y) + (xr[1].z + xr[1].w))) * (1.0f / 32);
    const f32x4 d0 = xr[0] - splat4(mw), d1 = xr[1] - splat4(mw);
    const float m2 = group_sum(((d0.x * d0.x + d0.y * d0.y) + (d0.z * d0.z + d0.w * d0.w)) + ((d1.x * d1.x + d1.y * d1.y) + (d1.z * d1.z + d1.w * d1.w)));
    if (lane < 16) L.stat[par][wave][c] = f32x2{mw, m2};
    bar_lds();
    f32x2 st[S_NW];
#pragma unroll
    for (int w = 0; w < S_NW; ++w) st[w] = L.stat[par][w][c];
    par ^= 1;
    float sm = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < S_NW; ++w) { sm += st[w].x; q += st[w].y; }
    mean = sm * (1.0f / S_NW);
    float dm = 0.f;
#pragma unroll
    for (int w = 0; w < S_NW; ++w) { const float e = st[w].x - mean; dm += e * e; }
    rstd = 1.0f / sqrtf((q + 32.0f * dm) * (1.0f / S_D) + eps);
  };
  // LayerNorm of the residual rows as the operand fragments of the next GEMM (L.xop)
  auto ln_operand = [&](const LnP& p) {
    float mean, rstd;
    row_stats(mean, rstd);
    f32x4 xn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) xn[j] = (xr[j] - splat4(mean)) * splat4(rstd) * p.g[j] + p.b[j];
    L.xop[wave][lane] = operand(xn[0], xn[1]);
    bar_lds();
  };
  auto xop_at = [&](int s) { return L.xop[s][lane]; };

  for (int bi = 0; bi < a.nblocks; ++bi) {
    const S256Block& B = a.blk[(S_DG & 1) ? 0 : bi];
    const bool more = bi + 1 < a.nblocks;
    const S256Block& BN = a.blk[(S_DG & 1) ? 0 : (more ? bi + 1 : bi)];
    auto pack = [](const void* p) { return reinterpret_cast<const u32x4_t*>(p); };
    const u32x4_t* __restrict__ w_f1a = pack(B.ff_w1[0]) + wave_frag0(8, wave, 8);
    const u32x4_t* __restrict__ w_f1b = pack(B.ff_w2[0]) + wave_frag0(2, wave, 32);
    const u32x4_t* __restrict__ w_qkv = pack(B.qkv_w) + wave * 64;
    const u32x4_t* __restrict__ w_out = pack(B.out_w) + wave_frag0(2, wave, 8);
    const u32x4_t* __restrict__ w_pw1 = pack(B.pw1_w) + (wave >> 1) * (8 * 512) + (wave & 1) * 128;
    const u32x4_t* __restrict__ w_pc1 = pack(B.pc_w1) + wave_frag0(4, wave, 8);
    const u32x4_t* __restrict__ w_pw2 = pack(B.pw2_w) + wave_frag0(2, wave, 16);
    const u32x4_t* __restrict__ w_f2a = pack(B.ff_w1[1]) + wave_frag0(8, wave, 8);
    const u32x4_t* __restrict__ w_f2b = pack(B.ff_w2[1]) + wave_frag0(2, wave, 32);
    const u32x4_t* __restrict__ w_nxt = pack(BN.ff_w1[0]) + wave_frag0(8, wave, 8);
    auto ld_ff1 = [&](const u32x4_t* w) { return [=](int t, u32x4_t (&d)[8]) { wload<8, 1>(w, t, OffPlain{}, lane, d); }; };
    auto ld_ff2 = [&](const u32x4_t* w) { return [=](int t, u32x4_t (&d)[8]) { wload<2, 4>(w, t, OffPlain{}, lane, d); }; };
    auto ld_qkv = [=](int t, u32x4_t (&d)[8]) { wload<6, 1>(w_qkv, t, OffQkv{}, lane, d); };
    auto ld_out = [=](int t, u32x4_t (&d)[8]) { wload<2, 4>(w_out, t, OffPlain{}, lane, d); };
    auto ld_pw1 = [=](int t, u32x4_t (&d)[8]) { wload<4, 2>(w_pw1, t, OffGlu{}, lane, d); };
    auto ld_pc1 = [=](int t, u32x4_t (&d)[8]) { wload<4, 2>(w_pc1, t, OffPlain{}, lane, d); };
    auto ld_pw2 = [=](int t, u32x4_t (&d)[8]) { wload<2, 4>(w_pw2, t, OffPlain{}, lane, d); };
    auto ld_next = [=](int t, u32x4_t (&d)[8]) { if (more) wload<8, 1>(w_nxt, t, OffPlain{}, lane, d); };

    // FFModule i: xr += fc * (swish(LN(x) W1 + b1) W2 + b2); pre2 = what to request before the stream moves on to the next layer
    auto ff_module = [&](auto rot, int i, auto ld1, auto ld2, auto ldn, auto pre2) {
      constexpr int ROT = decltype(rot)::value;
      ln_operand(lnp);
      f32x4 h[8], b1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = splat4(0.f);
      gemm<8, 1, 8, ROT>(h, wa, xop_at, ld1, ld2, [&] {
#pragma unroll
        for (int j = 0; j < 8; ++j) b1[j] = ldu(B.ff_b1[i] + 128 * wave + 16 * j, g4);
      });
#pragma unroll
      for (int p = 0; p < 4; ++p) L.hid[4 * wave + p][lane] = operand(act_swish(h[2 * p] + b1[2 * p]), act_swish(h[2 * p + 1] + b1[2 * p + 1]));
      bar_lds();
      f32x4 y[2] = {splat4(0.f), splat4(0.f)}, b2[2];
      gemm<2, 4, 32, (ROT + 8) % S_NBUF>(y, wa, [&](int s) { return L.hid[s][lane]; }, ld2, ldn, [&] {
#pragma unroll
        for (int j = 0; j < 2; ++j) b2[j] = ldu(B.ff_b2[i] + 32 * wave + 16 * j, g4);
        pre2();
      });
#pragma unroll
      for (int j = 0; j < 2; ++j) xr[j] = xr[j] + splat4(a.fc) * (y[j] + b2[j]);
    };

    // ---- ff_module_1 (batches 0 .. 15 of the block's stream)
    ff_module(std::integral_constant<int, 0>{}, 0, ld_ff1(w_f1a), ld_ff2(w_f1b), ld_qkv, [&] { load_ln(B.att_ln_g, B.att_ln_b, lnp); });

    // ---- mhsa_module: x += (softmax(q k^T) v) Wo + bo, q / k / v = LN(x) Wqkv + b (q scaled)
    ln_operand(lnp);
    {
      f32x4 q[6], qb[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) q[j] = splat4(0.f);
      auto ldn_qkv = [&](int j, u32x4_t (&d)[8]) { if (j < 2) ld_out(j, d); else ld_pw1(j - 2, d); };      // (the out-projection has two batches)
      gemm<6, 1, 8, 1>(q, wa, xop_at, ld_qkv, ldn_qkv, [&] {
#pragma unroll
        for (int j = 0; j < 6; ++j) qb[j] = ldu(B.qkv_b + 16 * wave + 128 * j, g4);
      });
#pragma unroll
      for (int j = 0; j < 6; ++j) {                            // tile 8 j + wave: j < 2 = the q tiles
        f32x4 v = q[j] + qb[j];
        if (j < 2) v = v * splat4(a.qscale);
        *reinterpret_cast<f32x4*>(&L.qkv[c][128 * j + 16 * wave + g4]) = v;
      }
    }
    bar_lds();
    if constexpr (!(S_DG & 4)) {
      // wave = head (wave & 3) x output half (wave >> 2).  S^T[key][query] = K Q^T: lane (g, c) reads row c of K and of Q, dims
      // 16 s + 4 g + {0..3}: sixteen v_mfma_f32_16x16x4_f32; its accumulator holds keys 4 g + {0..3} of query c.
      const unsigned hh = wave & 3, oh = wave >> 2;
      f32x4 st = splat4(0.f);
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&L.qkv[c][S_D + 64 * hh + 16 * sb + g4]);
        const f32x4 qf = *reinterpret_cast<const f32x4*>(&L.qkv[c][64 * hh + 16 * sb + g4]);
        st = mma_kblock(kf, qf, st);
      }
      const int k0 = (int)g4;
      st.x = k0 + 0 < T ? st.x : -INFINITY; st.y = k0 + 1 < T ? st.y : -INFINITY;
      st.z = k0 + 2 < T ? st.z : -INFINITY; st.w = k0 + 3 < T ? st.w : -INFINITY;
      const float mx = group_max(fmaxf(fmaxf(st.x, st.y), fmaxf(st.z, st.w)));          // key 0 is always real: finite
      const f32x4 e = {__expf(st.x - mx), __expf(st.y - mx), __expf(st.z - mx), __expf(st.w - mx)};     // exp(-inf) = 0 past T
      const float inv = 1.0f / group_sum((e.x + e.y) + (e.z + e.w));
      // O^T[feature][query] = V^T P^T: the accumulator IS the B operand (key 4 g + r in MFMA r); A = V[key 4 g + r][feature lane & 15]
      f32x4 o[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* vp = &L.qkv[g4][2 * S_D + 64 * hh + 32 * oh + 16 * t + c];
        o[t] = mfma4(vp[0], e.x, o[t]);
        o[t] = mfma4(vp[S_QSF], e.y, o[t]);
        o[t] = mfma4(vp[2 * S_QSF], e.z, o[t]);
        o[t] = mfma4(vp[3 * S_QSF], e.w, o[t]);
      }
      // features 64 h + 32 oh + {16 t + 4 g + r} of token c: k-step 2 h + oh of the out-projection, this lane's own slots
      L.ctxf[2 * hh + oh][lane] = operand(o[0] * splat4(inv), o[1] * splat4(inv));
    }
    bar_lds();
    {
      f32x4 y[2] = {splat4(0.f), splat4(0.f)}, ob[2];
      gemm<2, 4, 8, 0>(y, wa, [&](int s) { return L.ctxf[s][lane]; }, ld_out, ld_pw1, [&] {
#pragma unroll
        for (int j = 0; j < 2; ++j) ob[j] = ldu(B.out_b + 32 * wave + 16 * j, g4);
        load_ln(B.cv_ln_g, B.cv_ln_b, lnp);
      });
#pragma unroll
      for (int j = 0; j < 2; ++j) xr[j] = xr[j] + (y[j] + ob[j]);
    }

    // ---- conv_module: x += pw2(swish(BN(pointwise(depthwise(GLU(pw1(LN(x))))))))
    ln_operand(lnp);
    f32x4 dwt[KSZ][2];           // the depthwise taps of this wave's 32 channels
    {
      f32x4 u[4], pb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = splat4(0.f);
      gemm<4, 2, 8, 2>(u, wa, xop_at, ld_pw1, ld_pc1, [&] {
#pragma unroll
        for (int j = 0; j < 2; ++j) { pb[j] = ldu(B.pw1_b + 32 * wave + 16 * j, g4); pb[2 + j] = ldu(B.pw1_b + S_D + 32 * wave + 16 * j, g4); }
#pragma unroll
        for (int k = 0; k < KSZ; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j) dwt[k][j] = ldu(B.dw_w + k * S_D + 32 * wave + 16 * j, g4);
      });
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 va = u[j] + pb[j], vb = u[2 + j] + pb[2 + j];
        const f32x4 o = {va.x * act_sigmoid(vb.x), va.y * act_sigmoid(vb.y), va.z * act_sigmoid(vb.z), va.w * act_sigmoid(vb.w)};
        *reinterpret_cast<f32x4*>(&L.rows[c][32 * wave + 16 * j + g4]) = o;
      }
    }
    bar_lds();
    {
      f32x4 dv[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
      for (int k = 0; k < KSZ; ++k) {
        const int tt = (int)c + k - a.pad_left;
        const bool in = tt >= 0 && tt < T;
        const int tr = min(max(tt, 0), 15);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 uv = *reinterpret_cast<const f32x4*>(&L.rows[tr][32 * wave + 16 * j + g4]);
          dv[j] += (in ? uv : splat4(0.f)) * dwt[k][j];
        }
      }
      L.dwf[wave][lane] = operand(dv[0], dv[1]);
    }
    bar_lds();
    {
      f32x4 h[4], pcb[4], bns[4], bnt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = splat4(0.f);
      gemm<4, 2, 8, 0>(h, wa, [&](int s) { return L.dwf[s][lane]; }, ld_pc1, ld_pw2, [&] {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pcb[j] = ldu(B.pc_b1 + 64 * wave + 16 * j, g4);
          bns[j] = ldu(B.bn_s + 64 * wave + 16 * j, g4);
          bnt[j] = ldu(B.bn_t + 64 * wave + 16 * j, g4);
        }
      });
#pragma unroll
      for (int p = 0; p < 2; ++p)
        L.hid[2 * wave + p][lane] = operand(act_swish((h[2 * p] + pcb[2 * p]) * bns[2 * p] + bnt[2 * p]),
                                            act_swish((h[2 * p + 1] + pcb[2 * p + 1]) * bns[2 * p + 1] + bnt[2 * p + 1]));
    }
    bar_lds();
    {
      f32x4 y[2] = {splat4(0.f), splat4(0.f)}, p2b[2];
      gemm<2, 4, 16, 1>(y, wa, [&](int s) { return L.hid[s][lane]; }, ld_pw2, ld_ff1(w_f2a), [&] {
#pragma unroll
        for (int j = 0; j < 2; ++j) p2b[j] = ldu(B.pw2_b + 32 * wave + 16 * j, g4);
        load_ln(B.ff_ln_g[1], B.ff_ln_b[1], lnp);
      });
#pragma unroll
      for (int j = 0; j < 2; ++j) xr[j] = xr[j] + (y[j] + p2b[j]);
    }

    // ---- ff_module_2, then the block's LayerNorm
    LnP lnf;
    ff_module(std::integral_constant<int, 2>{}, 1, ld_ff1(w_f2a), ld_ff2(w_f2b), ld_next, [&] {
      load_ln(B.ln_g, B.ln_b, lnf);
      load_ln(BN.ff_ln_g[0], BN.ff_ln_b[0], lnp);          // (the last block: its own again, unused)
    });
    {
      float mean, rstd;
      row_stats(mean, rstd);
#pragma unroll
      for (int j = 0; j < 2; ++j) xr[j] = (xr[j] - splat4(mean)) * splat4(rstd) * lnf.g[j] + lnf.b[j];
    }
  }
  if ((int)c < T) {
    float* __restrict__ yw = a.y + row0 * S_D + 32 * wave;
    *reinterpret_cast<f32x4*>(yw + (c * S_D + g4)) = xr[0];
    *reinterpret_cast<f32x4*>(yw + 16 + (c * S_D + g4)) = xr[1];
  }
}

}  // namespace

// -1: not this kernel's shape
int launch_stream256(const S256Args& a, hipStream_t s) {
  if (!stream256_shape_ok(a.B, a.T, a.nblocks, a.ksz)) return -1;
  note_scheme(SCHEME_BF16);
  hipLaunchKernelGGL(stream256_kernel<5>, dim3(a.B), dim3(S_NW * 64), 0, s, a);
  return 0;
}
