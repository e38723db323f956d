// Full (unmasked) multi-head attention for head size 64 and up to 288 keys on the fp16 matrix pipe with two-term operands (round 5):
// the dmodel-256 configurations -- the streaming configuration's global CTC decoder over 64 x 260 history frames (BASELINE config 3)
// and conformerM offline (250 frames).  Reference: asr/models/layers/multihead_attention.py:151-188 (conformer_blocks.py:164-170).
//
// Why: attention_lds_kernel<64, 272> runs these shapes on v_mfma_f32_16x16x4_f32 -- 4 600 MFMAs of 32 cycles per (utterance, head)
// for Q K^T and as many for P V: 74 us per launch at 64 x 260, MFMA-bound.  The scheme of attention_split_kernel<2> (head size 36):
// q, k, v carry a static bound (LayerNorm output x the projection's column sums, api.hip), so each is hi + lo fp16 of the value
// times a power of two, three v_mfma_f32_16x16x32_f16 per fragment pair, fp32 accumulation and softmax: 2^-22 of the operand
// bounds, i.e. fp32-grade results (no rounding of the model's arithmetic: the bf16 mode's oracle is untouched) at a fifth of the
// matrix-pipe cycles.
//   * one workgroup of nine waves = one (utterance, head); K and V of the head are staged ONCE as fragments (144 KB of LDS);
//   * wave w takes the query tiles w and w + 9 (up to eighteen tiles of sixteen queries) one after the other;
//   * S^T = K Q^T (two 32-dim steps), softmax in registers (all keys of a query live in its four lanes), O^T = V^T P^T with the
//     split score accumulator of two key tiles as the B operand of a 32-key step (the k-slot order of attention_split.hip).
// Measured (BASELINE config 3, 64 streams x 260 frames x 4 heads): 74 -> ~31 us per launch, the config-3 step 0.733 -> 0.689 ms;
// 1e-6 from the fp64 oracle through a whole block like the fp32-MFMA kernels (tests).  What is left is not matrix work (two query
// tiles per wave ~5 us): 34 MB of k / v requested by 256 workgroups at once, V through 4-byte loads (its fragments hold one
// feature of eight keys), one workgroup per CU.
#include <cstdlib>

#include "common.h"
#include "env.h"
#include "launch.h"

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct Split2 { u32x4_t hi, lo; };

DEV unsigned pk_f16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t)); }
DEV Split2 split8h(f32x4 lo, f32x4 hi) {      // eight values -> hi + lo fp16 terms (the values already carry their power-of-two scale)
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned d0[4], d1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d0[k] = pk_f16(v[2 * k], v[2 * k + 1]);
    const f16x2_t h = __builtin_bit_cast(f16x2_t, d0[k]);
    d1[k] = pk_f16(v[2 * k] - (float)h.x, v[2 * k + 1] - (float)h.y);
  }
  return Split2{u32x4_t{d0[0], d0[1], d0[2], d0[3]}, u32x4_t{d1[0], d1[1], d1[2], d1[3]}};
}
DEV f32x4 mma32h(u32x4_t a, u32x4_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// c += A B with both operands split: lo x hi, hi x lo, hi x hi (smallest first)
DEV f32x4 mma2(u32x4_t ah, u32x4_t al, const Split2& b, f32x4 c) {
  c = mma32h(al, b.hi, c);
  c = mma32h(ah, b.lo, c);
  return mma32h(ah, b.hi, c);
}

constexpr int HS = 64;
constexpr int NKT = 18;         // key tiles of 16 (288 keys)
constexpr int NST = NKT / 2;    // 32-key steps of P V
constexpr int OT = 4;           // output feature tiles
constexpr int KS = 2;           // 32-dim steps of Q K^T
constexpr int AW = 9;           // waves: eighteen query tiles = two per wave (260 frames are seventeen: with eight waves one of them
                                // had three, the critical path of the workgroup); the 2304 K and 2304 V fragment entries = 4 + 4 per thread
constexpr int ATH = AW * 64;
constexpr int MAXQT = 2 * AW;   // query tiles a workgroup takes

__global__ __launch_bounds__(ATH) void attention_split64_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4_t Kf[2][KS][NKT][64];      // 72 KB  K fragments, hi / lo
  __shared__ __attribute__((aligned(16))) u32x4_t Vf[2][NST][OT][64];      // 72 KB  V^T fragments in step order, hi / lo
  // operands times their power-of-two scale; the score / output accumulators in units of sq sk / 2^14 sv
  const float sq = a.h2_sq, sk = a.h2_sk, sv = a.h2_sv;
  constexpr float SP = 16384.f;                                            // probabilities lie in [0, 1]
  constexpr float LOG2E = 1.4426950408889634f;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int T = a.Tk, TQ = a.Tq;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = a.ldk, D = a.D;
  const float* __restrict__ kbase = a.k + (size_t)b * T * ld + h * HS;
  const float* __restrict__ vbase = a.v + (size_t)b * T * ld + h * HS;

  // ---- stage K and V: every global load of the workgroup is requested before the first is used (a load per loop trip costs a
  // memory latency each: ten trips were half of the first version's 35 us), then split and written as fragments.
  // K entry (tile kt, step ks, lane (g, c)) = key 16 kt + c, dims 32 ks + 8 g + {0..7}; keys past T are exact zeros.
  // V entry (step s, feature tile ot, lane (kg, fc)) = feature 16 ot + fc of the keys 32 s + 4 kg + {0..3} and 32 s + 16 + 4 kg +
  // {0..3}, in that k-slot order (the order in which two score tiles sit side by side, see attention_split.hip)
  constexpr int NKE = KS * NKT * 64, NKI = (NKE + ATH - 1) / ATH;
  constexpr int NVE = NST * OT * 64, NVI = (NVE + ATH - 1) / ATH;
  f32x4 kr[NKI][2];
  float vr[NVI][8];
#pragma unroll
  for (int it = 0; it < NKI; ++it) {
    const int en = tid + it * ATH, l = en & 63, r = en >> 6, kt = r % NKT, ks = r / NKT;
    const int key = 16 * kt + (l & 15), d0 = 32 * ks + 8 * (l >> 4);
    const bool in = en < NKE && key < T;
    const float* p = kbase + (size_t)(in ? key : 0) * ld + d0;
    kr[it][0] = ldg4(p); kr[it][1] = ldg4(p + 4);
    if (!in) { kr[it][0] = splat4(0.f); kr[it][1] = splat4(0.f); }
  }
#pragma unroll
  for (int it = 0; it < NVI; ++it) {
    const int en = tid + it * ATH, l = en & 63, r = en >> 6, ot = r % OT, s = r / OT, kg = l >> 4, f = 16 * ot + (l & 15);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 32 * s + 16 * (j >> 2) + 4 * kg + (j & 3);
      const bool in = en < NVE && key < T;                               // padded keys: exact zeros (0 x garbage = NaN)
      const float v = vbase[(size_t)(in ? key : 0) * ld + f];
      vr[it][j] = in ? v : 0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < NKI; ++it) {
    const int en = tid + it * ATH;
    if (en < NKE) {
      const Split2 f = split8h(kr[it][0] * splat4(sk), kr[it][1] * splat4(sk));
      (&Kf[0][0][0][0])[en] = f.hi;                                      // [term][ks][kt][lane]: en = (ks NKT + kt) 64 + lane
      (&Kf[1][0][0][0])[en] = f.lo;
    }
  }
#pragma unroll
  for (int it = 0; it < NVI; ++it) {
    const int en = tid + it * ATH;
    if (en < NVE) {
      const Split2 f2 = split8h(f32x4{vr[it][0], vr[it][1], vr[it][2], vr[it][3]} * splat4(sv), f32x4{vr[it][4], vr[it][5], vr[it][6], vr[it][7]} * splat4(sv));
      (&Vf[0][0][0][0])[en] = f2.hi;                                     // [term][s][ot][lane]: en = (s OT + ot) 64 + lane
      (&Vf[1][0][0][0])[en] = f2.lo;
    }
  }
  __syncthreads();

  const int nkt = (T + 15) / 16, nst = (nkt + 1) / 2;
  const int nqt = (TQ + 15) / 16;
  const f32x4 inv_qk = splat4(1.0f / (sq * sk));                           // powers of two
  for (int qt = blockIdx.x * MAXQT + wv; qt < min(nqt, (int)(blockIdx.x + 1) * MAXQT); qt += AW) {
    const int tq = qt * 16 + c;
    const float* qrow = a.q + ((size_t)b * TQ + min(tq, TQ - 1)) * a.ldq + h * HS;
    Split2 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[ks] = split8h(ldg4(qrow + 32 * ks + 8 * g) * splat4(LOG2E * sq), ldg4(qrow + 32 * ks + 8 * g + 4) * splat4(LOG2E * sq));
    // ---- S^T = K Q^T: lane holds S^T[key = 16 kt + 4 g + j][query c] in log2 units
    f32x4 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        f32x4 acc = splat4(0.f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mma2(Kf[0][ks][kt][lane], Kf[1][ks][kt][lane], qf[ks], acc);
        sc[kt] = acc * inv_qk;
      } else {
        sc[kt] = splat4(-INFINITY);
      }
    }
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
      sc[kt].x = __builtin_amdgcn_exp2f(sc[kt].x - mx);      // exp2(-inf) = 0 for masked keys
      sc[kt].y = __builtin_amdgcn_exp2f(sc[kt].y - mx);
      sc[kt].z = __builtin_amdgcn_exp2f(sc[kt].z - mx);
      sc[kt].w = __builtin_amdgcn_exp2f(sc[kt].w - mx);
      psum += (sc[kt].x + sc[kt].y) + (sc[kt].z + sc[kt].w);
    }
    // ---- O^T[feat][query] += V^T[feat][key] P^T[key][query], 32 keys per step
    f32x4 o[OT];
#pragma unroll
    for (int i = 0; i < OT; ++i) o[i] = splat4(0.f);
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      if (s < nst) {
        const Split2 pf = split8h(sc[2 * s] * splat4(SP), sc[2 * s + 1] * splat4(SP));
#pragma unroll
        for (int i = 0; i < OT; ++i) o[i] = mma2(Vf[0][s][i][lane], Vf[1][s][i][lane], pf, o[i]);
      }
    }
    const float inv = (1.0f / group_sum(psum)) * (1.0f / (SP * sv));      // the second factor is a power of two
    if (tq < TQ) {
      float* orow = a.ctx + ((size_t)b * TQ + tq) * D + h * HS;
#pragma unroll
      for (int i = 0; i < OT; ++i) stg4(orow + 16 * i + g4, o[i] * splat4(inv));
    }
  }
}

}  // namespace

bool attention_split64_applicable(int hs, const AttnArgs& a) {
  // MI355ASR_ATTN64_SPLIT=0: the fp32-MFMA kernels (attention_lds_kernel<64, 272> / attention_kernel<64>) as before
  static const bool on = mi355_env("MI355ASR_ATTN64_SPLIT", 1) != 0;
  return on && hs == HS && a.win_front < 0 && !a.head_major && a.Tk <= 16 * NKT && a.Tk > 32 && a.Tq > 16 && a.ldk % 4 == 0 && a.ldq % 4 == 0 &&
         a.h2_sq > 0.f && a.h2_sk > 0.f && a.h2_sv > 0.f;
}

int launch_attention_split64(int hs, const AttnArgs& a, hipStream_t s) {
  if (!attention_split64_applicable(hs, a)) return -1;
  note_scheme(SCHEME_F16X2);
  const int qtiles = (a.Tq + 15) / 16;
  hipLaunchKernelGGL(attention_split64_kernel, dim3((qtiles + MAXQT - 1) / MAXQT, a.H, a.B), dim3(ATH), 0, s, a);
  return 0;
}
