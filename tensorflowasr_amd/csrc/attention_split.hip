// Full (unmasked) multi-head attention for utterances of up to 256 frames, head size 36, on the bf16 matrix pipe with
// exactly split operands (round 2; replaces attention_lds_kernel on the offline dmodel-144 path).
// Reference: asr/models/layers/multihead_attention.py:151-188, called from conformer_blocks.py:164-170.
//
// Why: attention_lds_kernel runs 336 v_mfma_f32_16x16x4_f32 per wave (32 cycles each) and stages K / V^T twice per
// (utterance, head) -- once per 128-query workgroup: 35 us per launch, 2.5x the algorithmic HBM traffic.  Here
//   * one workgroup of 16 waves = all (<= 256) queries of one (utterance, head): K and V are staged ONCE, as MFMA
//     fragments: every fp32 value is written as three bf16 terms (exact: 3 x 8 significand bits), a fragment read is one
//     conflict-free ds_read_b128;
//   * S^T = K Q^T: dims 0..31 as one 16x16x32 bf16 step = six MFMAs over the term pairs (i + j <= 2, smallest first;
//     the dropped pairs are below 2^-24 of a product), dims 32..35 as one fp32 16x16x4 MFMA: 128 cycles per 16-key tile
//     instead of 288;
//   * softmax in registers as before (all <= 256 scores of a query live in its four lanes, exp2 with log2 e folded into q);
//   * O^T = V^T P^T: the S^T accumulator of two key tiles IS the B operand of a 32-key step once split (the k-slot order
//     of a step is chosen to match: slot 8g + j <-> key 32s + 4g + j, slot 8g + 4 + j <-> key 32s + 16 + 4g + j, and V is
//     staged in that order): 18 MFMAs of 16 cycles per 32 keys instead of 24 of 32.
// 4352 matrix-pipe cycles per wave instead of 10 752.  Grid (ceil(Tq / 256), H, B): at 64 x 10 s exactly one workgroup
// per CU, four waves per SIMD.
#include <cstdlib>

#include "common.h"
#include "env.h"
#include "launch.h"

namespace {

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
// c += A B with both operands split: the six term pairs with i + j <= 2, smallest first
DEV f32x4 mma_split(const u32x4_t (&a)[3], const Split8& b, f32x4 c) {
  c = mma32(a[2], b.t[0], c);
  c = mma32(a[1], b.t[1], c);
  c = mma32(a[0], b.t[2], c);
  c = mma32(a[1], b.t[0], c);
  c = mma32(a[0], b.t[1], c);
  c = mma32(a[0], b.t[0], c);
  return c;
}

// Two-term scheme (see subconv.hip / fused_pp.hip): hi + lo fp16 terms of values already multiplied by their power-of-two
// scale, three products per fragment pair.  Split8::t[2] is unused then.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
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
template <int TM>
DEV Split8 split_tm(f32x4 lo, f32x4 hi) {
  if constexpr (TM == 2) return split8h(lo, hi); else return split8(lo, hi);
}
// c += A B: TM = 3 the six bf16 term pairs, TM = 2 the three fp16 ones (lo x hi, hi x lo, hi x hi)
template <int TM>
DEV f32x4 mma_tm(const u32x4_t (&a)[3], const Split8& b, f32x4 c) {
  if constexpr (TM == 2) {
    c = mma32h(a[1], b.t[0], c);
    c = mma32h(a[0], b.t[1], c);
    c = mma32h(a[0], b.t[0], c);
    return c;
  } else {
    return mma_split(a, b, c);
  }
}

#ifndef MI355ASR_ATTN_DIAG
#define MI355ASR_ATTN_DIAG 0
#endif
// timing experiments (tools/build_variant.py ... -DMI355ASR_ATTN_DIAG=n; WRONG results): bit 0 = no Q K^T products, 1 = no exp2,
// 2 = no P V products (and no split of P), 3 = no V fragment writes, 4 = no operand loads of K / V
constexpr int ADG = MI355ASR_ATTN_DIAG;
constexpr int HS = 36;
constexpr int TPK = 256;        // keys held in LDS
constexpr int NKT = TPK / 16;   // key tiles
constexpr int NST = TPK / 32;   // 32-key steps of P V
constexpr int OT = 3;           // output feature tiles (48 >= 36)
constexpr int AW = 16;          // waves = query tiles per workgroup
constexpr int ATH = AW * 64;

template <int TM>
__global__ __launch_bounds__(ATH) void attention_split_kernel(AttnArgs a) {
  static_assert(TM == 3 || ADG == 0, "the timing variants are those of the three-term kernel");
  __shared__ __attribute__((aligned(16))) u32x4_t Kf[TM][NKT][64];       // 48 KB  K fragments (dims 0..31), TM terms
  __shared__ __attribute__((aligned(16))) float Kt[NKT][64];              //  4 KB  K[key][32 + g] for the fp32 tail step
  __shared__ __attribute__((aligned(16))) u32x4_t Vf[TM][NST][OT][64];    // 72 KB  V^T fragments in step order, TM terms
  // two-term scheme: operands times their power-of-two scale, the score / output accumulators in units of sq sk / 2^14 sv
  const float sq = TM == 2 ? a.h2_sq : 1.f, sk = TM == 2 ? a.h2_sk : 1.f, sv = TM == 2 ? a.h2_sv : 1.f;
  constexpr float SP = TM == 2 ? 16384.f : 1.f;                          // probabilities lie in [0, 1]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int T = a.Tk, TQ = a.Tq;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = a.ldk, D = a.D;
  // token-major: row (b T + t) of ld floats, head h at columns [36 h, 36 h + 36); head-major (round 5): row ((b H + h) T + t) of 36
  const size_t khead = a.head_major ? ((size_t)b * a.H + h) * T * HS : (size_t)b * T * ld + h * HS;
  const float* __restrict__ kbase = a.k + khead;
  const float* __restrict__ vbase = a.v + khead;

  // ---- this lane's query fragment: query c of tile qt, dims 8g..8g+7 and 32 + g (log2 e folded in: softmax uses exp2)
  const int qt = blockIdx.x * AW + wv;
  const int tq = qt * 16 + c;
  const float* qrow = a.head_major ? a.q + (((size_t)b * a.H + h) * TQ + min(tq, TQ - 1)) * HS
                                   : a.q + ((size_t)b * TQ + min(tq, TQ - 1)) * a.ldq + h * HS;
  constexpr float LOG2E = 1.4426950408889634f;
  const f32x4 qlo = ldg4(qrow + 8 * g), qhi = ldg4(qrow + 8 * g + 4);
  const float qtl = qrow[32 + g] * (LOG2E * sq);

  // ---- stage K: thread (tile wv, lane) owns exactly one fragment triple; all global loads first, then the LDS writes
  const int skey = 16 * wv + c;
  const float* krow = kbase + (size_t)min(skey, T - 1) * ld;
  f32x4 klo = ldg4(krow + 8 * g), khi = ldg4(krow + 8 * g + 4);
  float ktl = krow[32 + g];
  // ---- stage V (round 3): a thread owns whole 16-byte fragment entries -- entry (step s, feature tile ot, lane (kg, fc)) =
  // feature 16 ot + fc of the eight keys 32 s + 4 kg + {0..3} and 32 s + 16 + 4 kg + {0..3}, in that k-slot order (see the
  // header) -- loads its eight values, splits them and writes three conflict-free ds_write_b128.  (Round 2 walked V row-major
  // and scattered 36 ds_write_b16 per thread into the fragments: 31 % of the kernel's LDS cycles were bank conflicts.)
  constexpr int NVE = NST * OT * 64, NVT = (NVE + ATH - 1) / ATH;      // 1536 entries, two rounds of the 1024 threads
  float ve[NVT][8];
#pragma unroll
  for (int it = 0; it < NVT; ++it) {
    const int en = tid + it * ATH;
    const int s = en / (OT * 64), rem = en - s * (OT * 64), ot = rem >> 6, l = rem & 63, kg = l >> 4, fc = l & 15;
    const int f = 16 * ot + fc;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 32 * s + 16 * (j >> 2) + 4 * kg + (j & 3);
      // padded keys and the padding features 36..47 must be exact zeros (0 x garbage = NaN)
      ve[it][j] = (!(ADG & 16) && en < NVE && f < HS && key < T) ? vbase[(size_t)key * ld + f] : 0.f;
    }
  }
  if (skey >= T) { klo = splat4(0.f); khi = splat4(0.f); ktl = 0.f; }
  {
    const Split8 kf = split_tm<TM>(klo * splat4(sk), khi * splat4(sk));
    Kf[0][wv][lane] = kf.t[0];
    Kf[1][wv][lane] = kf.t[1];
    if constexpr (TM == 3) Kf[2][wv][lane] = kf.t[2];
    Kt[wv][lane] = ktl * sk;
  }
  __syncthreads();                     // K fragments staged; the V rows are still in flight / in registers
  const bool active = qt * 16 < TQ;    // (waves without queries still stage their share of V)
  const int nkt = (T + 15) / 16;       // key tiles that hold at least one real key (uniform)
  const Split8 qf = split_tm<TM>(qlo * splat4(LOG2E * sq), qhi * splat4(LOG2E * sq));
  const f32x4 inv_qk = splat4(1.0f / (sq * sk));                        // powers of two

  // ---- S^T = K Q^T: lane holds S^T[key = 16 kt + 4 g + j][query c] in log2 units
  f32x4 sc[NKT];
  auto qk_tile = [&](int kt) {
    const u32x4_t kf[3] = {Kf[0][kt][lane], Kf[1][kt][lane], Kf[TM - 1][kt][lane]};
    const float kl = Kt[kt][lane];
    if constexpr (ADG & 1) { sc[kt] = splat4(kl) + __builtin_bit_cast(f32x4, kf[0]); return; }
    f32x4 acc = mma_tm<TM>(kf, qf, splat4(0.f));
    sc[kt] = mfma4(kl, qtl, acc);
    if constexpr (TM == 2) sc[kt] = sc[kt] * inv_qk;
  };
  const bool full = (nkt == NKT);
  if (active) {
    if (full) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) qk_tile(kt);
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) qk_tile(kt); else sc[kt] = splat4(-INFINITY);
      }
    }
  }

  // ---- the V fragments behind the Q K^T products (round 3: the loads were issued before the K staging, so their latency no
  // longer sits in front of the first MFMA; the LDS writes drain under the softmax)
#pragma unroll
  for (int it = 0; it < NVT; ++it) {
    const int en = tid + it * ATH;
    if (en < NVE && !(ADG & 8)) {
      const f32x4 lo = {ve[it][0], ve[it][1], ve[it][2], ve[it][3]}, hi = {ve[it][4], ve[it][5], ve[it][6], ve[it][7]};
      const Split8 vf = split_tm<TM>(lo * splat4(sv), hi * splat4(sv));
      u32x4_t* dst = &Vf[0][0][0][0] + en;                         // [t][s][ot][lane]: en = (s * OT + ot) * 64 + lane
      dst[0] = vf.t[0];
      dst[NVE] = vf.t[1];
      if constexpr (TM == 3) dst[2 * NVE] = vf.t[2];
    }
  }
  if (!active) { __syncthreads(); return; }

  // ---- softmax over keys: keys >= T are masked, which only the last real tile can contain
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt < nkt && 16 * kt + 16 > T) {
      const int kb = 16 * kt + g4;
      sc[kt].x = (kb + 0 < T) ? sc[kt].x : -INFINITY;
      sc[kt].y = (kb + 1 < T) ? sc[kt].y : -INFINITY;
      sc[kt].z = (kb + 2 < T) ? sc[kt].z : -INFINITY;
      sc[kt].w = (kb + 3 < T) ? sc[kt].w : -INFINITY;
    }
    mx = fmaxf(mx, fmaxf(fmaxf(sc[kt].x, sc[kt].y), fmaxf(sc[kt].z, sc[kt].w)));
  }
  mx = group_max(mx);                  // every query sees key 0, so mx is finite
  float psum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if constexpr (!(ADG & 2)) {
      sc[kt].x = __builtin_amdgcn_exp2f(sc[kt].x - mx);      // exp2(-inf) = 0 for masked keys
      sc[kt].y = __builtin_amdgcn_exp2f(sc[kt].y - mx);
      sc[kt].z = __builtin_amdgcn_exp2f(sc[kt].z - mx);
      sc[kt].w = __builtin_amdgcn_exp2f(sc[kt].w - mx);
    }
    psum += (sc[kt].x + sc[kt].y) + (sc[kt].z + sc[kt].w);
  }

  __syncthreads();                     // V fragments staged
  // ---- O^T[feat][query] += V^T[feat][key] P^T[key][query], 32 keys per step
  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = splat4(0.f);
  const int nst = (nkt + 1) / 2;
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    if constexpr (ADG & 4) { o[0] += sc[2 * s]; o[1] += sc[2 * s + 1]; }
    else if (s < nst) {
      const Split8 pf = split_tm<TM>(sc[2 * s] * splat4(SP), sc[2 * s + 1] * splat4(SP));
#pragma unroll
      for (int i = 0; i < OT; ++i) {
        const u32x4_t vf[3] = {Vf[0][s][i][lane], Vf[1][s][i][lane], Vf[TM - 1][s][i][lane]};
        o[i] = mma_tm<TM>(vf, pf, o[i]);
      }
    }
  }
  const float inv = (1.0f / group_sum(psum)) * (1.0f / (SP * sv));      // the second factor is a power of two
  if (tq < TQ) {
    float* orow = a.ctx + ((size_t)b * TQ + tq) * D + h * HS;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      if (16 * i + g4 < HS) stg4(orow + 16 * i + g4, o[i] * splat4(inv));
    }
  }
}


// ---- more than 256 keys (round 6) ------------------------------------------------------------------------------------------------------
// multihead_attention.py:151-188 has no length limit; utterances beyond 10.24 s used to fall to the fp32-MFMA kernels (35 us per
// launch already at 250 frames).  A workgroup of 16 or 8 waves = 256 / 128 queries of one (utterance, head) walks the keys in
// blocks of 256 (in quarters / halves: one softmax update each) with an online softmax: per block the K / V fragments are staged as above (the next block's rows are requested into
// registers before this block's products, so their latency hides under the MFMAs), S^T = K Q^T for the block's 16 key tiles, the
// running maximum m and the per-lane partial sums are updated (p = exp2(s - m_new), everything carried so far times exp2(m_old -
// m_new): the factor belongs to the lane's query, so the output accumulators -- features x this query -- scale per lane), then
// O^T += V^T P^T.  Two workgroup barriers per block (fragments staged / fragments free).  Grid (ceil(Tq / 256 or 128), H, B).
constexpr int LKB = 256;         // keys per block, worked through in halves / quarters (32 / 16 score registers)
constexpr int LKT = LKB / 16;    // 16 key tiles
constexpr int LST = LKB / 32;    // 8 steps of P V
// LW = waves = query tiles per workgroup.  16 (256 queries, 128 registers: the block in quarters, its rows fetched at the top of the
// iteration) stages K / V once per 256 queries and keeps four waves per SIMD; 8 (128 queries, 256 registers: halves, the next block's
// rows prefetched into registers under the products) has twice the workgroups of 0.62 of the duration (measured, T = 375 ... 1000):
// the launcher takes whichever fills the chip's rounds better (T = 375, 42 utterances: 38 against 47 us; T = 750, 21: 49 against 60).
template <int TM, int LW>
__global__ __launch_bounds__(LW * 64) void attention_split_long_kernel(AttnArgs a) {
  constexpr int LTH = LW * 64;
  constexpr bool LPRE = LW <= 8;
  __shared__ __attribute__((aligned(16))) u32x4_t Kf[TM][LKT][64];
  __shared__ __attribute__((aligned(16))) float Kt[LKT][64];
  __shared__ __attribute__((aligned(16))) u32x4_t Vf[TM][LST][OT][64];
  const float sq = TM == 2 ? a.h2_sq : 1.f, sk = TM == 2 ? a.h2_sk : 1.f, sv = TM == 2 ? a.h2_sv : 1.f;
  constexpr float SP = TM == 2 ? 16384.f : 1.f;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int T = a.Tk, TQ = a.Tq;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = a.ldk, D = a.D;
  const size_t khead = a.head_major ? ((size_t)b * a.H + h) * T * HS : (size_t)b * T * ld + h * HS;
  const float* __restrict__ kbase = a.k + khead;
  const float* __restrict__ vbase = a.v + khead;
  const int qt = blockIdx.x * LW + wv;
  const int tq = qt * 16 + c;
  const float* qrow = a.head_major ? a.q + (((size_t)b * a.H + h) * TQ + min(tq, TQ - 1)) * HS
                                   : a.q + ((size_t)b * TQ + min(tq, TQ - 1)) * a.ldq + h * HS;
  constexpr float LOG2E = 1.4426950408889634f;
  const f32x4 qlo = ldg4(qrow + 8 * g), qhi = ldg4(qrow + 8 * g + 4);
  const float qtl = qrow[32 + g] * (LOG2E * sq);
  const Split8 qf = split_tm<TM>(qlo * splat4(LOG2E * sq), qhi * splat4(LOG2E * sq));
  const f32x4 inv_qk = splat4(1.0f / (sq * sk));
  const bool active = qt * 16 < TQ;

  constexpr int NKE = LKT * 64, NKR = (NKE + LTH - 1) / LTH;             // K fragment entries (tile, lane): 896, two rounds
  constexpr int NVE = LST * OT * 64, NVR = (NVE + LTH - 1) / LTH;        // V fragment entries: 1344, three rounds
  f32x4 klo[NKR], khi[NKR];
  float ktl[NKR];
  float ve[NVR][8];
  // rows of the key block that starts at kb0 into registers; keys >= T and the padding features are exact zeros.  32-bit element
  // offsets from the (uniform) head base: one address register per load instead of two, and nothing per-load for the compiler to
  // carry around the loop (with 64-bit pointers it kept all thirty addresses live: 256 registers and a spill)
  const unsigned uld = (unsigned)ld;
  auto fetch = [&](int kb0) {
#pragma unroll
    for (int it = 0; it < NKR; ++it) {
      const int en = tid + it * LTH, kt = en >> 6, l = en & 63, kg = l >> 4, kc = l & 15;
      const int skey = kb0 + 16 * kt + kc;
      const unsigned ko = (unsigned)min(skey, T - 1) * uld + 8u * kg;
      const bool ok = en < NKE && skey < T;
      klo[it] = ok ? *reinterpret_cast<const f32x4*>(kbase + ko) : splat4(0.f);
      khi[it] = ok ? *reinterpret_cast<const f32x4*>(kbase + ko + 4u) : splat4(0.f);
      ktl[it] = ok ? kbase[(unsigned)min(skey, T - 1) * uld + 32u + kg] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NVR; ++it) {
      const int en = tid + it * LTH;
      const int s = en / (OT * 64), rem = en - s * (OT * 64), ot = rem >> 6, l = rem & 63, kg = l >> 4, fc = l & 15;
      const int f = 16 * ot + fc;
      const int key0 = kb0 + 32 * s + 4 * kg;
      const unsigned vo = (unsigned)key0 * uld + (unsigned)f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int dk = 16 * (j >> 2) + (j & 3);
        ve[it][j] = (en < NVE && f < HS && key0 + dk < T) ? vbase[vo + (unsigned)dk * uld] : 0.f;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < NKR; ++it) {
      const int en = tid + it * LTH;
      if (en < NKE) {
        const Split8 kf = split_tm<TM>(klo[it] * splat4(sk), khi[it] * splat4(sk));
        (&Kf[0][0][0])[en] = kf.t[0];
        (&Kf[1][0][0])[en] = kf.t[1];
        if constexpr (TM == 3) (&Kf[2][0][0])[en] = kf.t[2];
        (&Kt[0][0])[en] = ktl[it] * sk;
      }
    }
#pragma unroll
    for (int it = 0; it < NVR; ++it) {
      const int en = tid + it * LTH;
      if (en < NVE) {
        const f32x4 lo = {ve[it][0], ve[it][1], ve[it][2], ve[it][3]}, hi = {ve[it][4], ve[it][5], ve[it][6], ve[it][7]};
        const Split8 vf = split_tm<TM>(lo * splat4(sv), hi * splat4(sv));
        u32x4_t* dst = &Vf[0][0][0][0] + en;
        dst[0] = vf.t[0];
        dst[NVE] = vf.t[1];
        if constexpr (TM == 3) dst[2 * NVE] = vf.t[2];
      }
    }
  };

  float m = -INFINITY, psum = 0.f;
  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = splat4(0.f);
  if constexpr (LPRE) fetch(0);
#pragma unroll 1
  for (int kb0 = 0; kb0 < T; kb0 += LKB) {
    if constexpr (!LPRE) fetch(kb0);
    stage();
    __syncthreads();                                   // this block's fragments staged
    if constexpr (LPRE) {
      if (kb0 + LKB < T) fetch(kb0 + LKB);             // the next block's rows: in flight under the products
    }
    if (active) {
      const int nk = min(T - kb0, LKB), nkt = (nk + 15) / 16;
      // the block in two halves of eight key tiles (one online-softmax update each): 32 score registers instead of 64 leave room
      // for the next block's rows
      constexpr int NH = LPRE ? 2 : 4, HT = LKT / NH, HS2 = LST / NH;      // (16-wave workgroups have 128 registers: quarters)
#pragma unroll
      for (int hb = 0; hb < NH; ++hb) {
        if (hb * HT < nkt) {
          f32x4 sc[HT];
#pragma unroll
          for (int k8 = 0; k8 < HT; ++k8) {
            const int kt = hb * HT + k8;
            if (kt < nkt) {
              const u32x4_t kf[3] = {Kf[0][kt][lane], Kf[1][kt][lane], Kf[TM - 1][kt][lane]};
              f32x4 acc = mma_tm<TM>(kf, qf, splat4(0.f));
              sc[k8] = mfma4(Kt[kt][lane], qtl, acc);
              if constexpr (TM == 2) sc[k8] = sc[k8] * inv_qk;
            } else {
              sc[k8] = splat4(-INFINITY);
            }
          }
          float bm = -INFINITY;
#pragma unroll
          for (int k8 = 0; k8 < HT; ++k8) {
            const int kt = hb * HT + k8;
            if (kt < nkt && 16 * kt + 16 > nk) {
              const int kk = 16 * kt + g4;
              sc[k8].x = (kk + 0 < nk) ? sc[k8].x : -INFINITY;
              sc[k8].y = (kk + 1 < nk) ? sc[k8].y : -INFINITY;
              sc[k8].z = (kk + 2 < nk) ? sc[k8].z : -INFINITY;
              sc[k8].w = (kk + 3 < nk) ? sc[k8].w : -INFINITY;
            }
            bm = fmaxf(bm, fmaxf(fmaxf(sc[k8].x, sc[k8].y), fmaxf(sc[k8].z, sc[k8].w)));
          }
          const float mn = fmaxf(m, group_max(bm));        // finite: the half holds at least one real key (hb * HT < nkt)
          const float alpha = __builtin_amdgcn_exp2f(m - mn);      // very first half: exp2(-inf) = 0
          m = mn;
          float bs = 0.f;
#pragma unroll
          for (int k8 = 0; k8 < HT; ++k8) {
            sc[k8].x = __builtin_amdgcn_exp2f(sc[k8].x - mn);
            sc[k8].y = __builtin_amdgcn_exp2f(sc[k8].y - mn);
            sc[k8].z = __builtin_amdgcn_exp2f(sc[k8].z - mn);
            sc[k8].w = __builtin_amdgcn_exp2f(sc[k8].w - mn);
            bs += (sc[k8].x + sc[k8].y) + (sc[k8].z + sc[k8].w);
          }
          psum = psum * alpha + bs;
#pragma unroll
          for (int i = 0; i < OT; ++i) o[i] = o[i] * splat4(alpha);
          const int nst = (nkt + 1) / 2;
#pragma unroll
          for (int s4 = 0; s4 < HS2; ++s4) {
            const int s = hb * HS2 + s4;
            if (s < nst) {
              const Split8 pf = split_tm<TM>(sc[2 * s4] * splat4(SP), sc[2 * s4 + 1] * splat4(SP));
#pragma unroll
              for (int i = 0; i < OT; ++i) {
                const u32x4_t vf[3] = {Vf[0][s][i][lane], Vf[1][s][i][lane], Vf[TM - 1][s][i][lane]};
                o[i] = mma_tm<TM>(vf, pf, o[i]);
              }
            }
          }
        }
      }
    }
    __syncthreads();                                   // every wave is done with this block's fragments
  }
  if (!active) return;
  const float inv = (1.0f / group_sum(psum)) * (1.0f / (SP * sv));
  if (tq < TQ) {
    float* orow = a.ctx + ((size_t)b * TQ + tq) * D + h * HS;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      if (16 * i + g4 < HS) stg4(orow + 16 * i + g4, o[i] * splat4(inv));
    }
  }
}

}  // namespace

static bool attn_long_on() {
  // MI355ASR_ATTN_LONG=0: more than 256 keys on the fp32-MFMA kernels (attention_lds_kernel / attention_kernel) as before round 6
  static const bool on = mi355_env("MI355ASR_ATTN_LONG", 1) != 0;
  return on;
}
bool attention_split_applicable(int hs, const AttnArgs& a) {
  return hs == HS && a.win_front < 0 && (a.Tk <= TPK || attn_long_on()) && a.Tk > 16 && a.Tq > 16 && a.ldk % 4 == 0 && a.ldq % 4 == 0;
}

static bool attn_three_env() {
  // MI355ASR_ATTN_TERMS=3: the three-term bf16 kernel also where the operand bounds are known
  static const bool three = mi355_env("MI355ASR_ATTN_TERMS", -1) == 3;
  return three;
}
bool attention_split_two_term(int hs, const AttnArgs& a) {
  return attention_split_applicable(hs, a) && a.h2_sq > 0.f && a.h2_sk > 0.f && a.h2_sv > 0.f && !attn_three_env() && ADG == 0;
}

int launch_attention_split(int hs, const AttnArgs& a, hipStream_t s) {
  if (!attention_split_applicable(hs, a)) return -1;
  const int qtiles = (a.Tq + 15) / 16;
  const bool two = attention_split_two_term(hs, a);
  if (a.head_major && (!two || a.ldq != HS || a.ldk != HS)) return -1;       // head-major operands: this kernel's two-term form only
  note_scheme(two ? SCHEME_F16X2 : SCHEME_BF16X3);
  const dim3 grid((qtiles + AW - 1) / AW, a.H, a.B);
  if (a.Tk > TPK) {                      // key blocks of 256 with an online softmax; 256 or 128 queries per workgroup
    static const int ncu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
    const long n16 = (long)((qtiles + 15) / 16) * a.H * a.B, n8 = (long)((qtiles + 7) / 8) * a.H * a.B;
    const double c16 = (double)((n16 + ncu - 1) / ncu), c8 = 0.62 * (double)((n8 + ncu - 1) / ncu);      // rounds x duration of a round
    if (c16 <= c8) {
      const dim3 gl((qtiles + 15) / 16, a.H, a.B);
      if (two) hipLaunchKernelGGL((attention_split_long_kernel<2, 16>), gl, dim3(1024), 0, s, a);
      else hipLaunchKernelGGL((attention_split_long_kernel<3, 16>), gl, dim3(1024), 0, s, a);
    } else {
      const dim3 gl((qtiles + 7) / 8, a.H, a.B);
      if (two) hipLaunchKernelGGL((attention_split_long_kernel<2, 8>), gl, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((attention_split_long_kernel<3, 8>), gl, dim3(512), 0, s, a);
    }
    return 0;
  }
  // (Round 6 measured two 8-wave workgroups per CU -- 128 queries each, the fragments trimmed to 72 KB, 110 VGPRs, bit-identical --
  // against this one 16-wave workgroup: 20.7 us both, profiles/r06_attention_pair_ab.jsonl; the kernel is in commit 33c4385.)
  if (two)
    hipLaunchKernelGGL(attention_split_kernel<2>, grid, dim3(ATH), 0, s, a);
  else
    hipLaunchKernelGGL(attention_split_kernel<3>, grid, dim3(ATH), 0, s, a);
  return 0;
}
