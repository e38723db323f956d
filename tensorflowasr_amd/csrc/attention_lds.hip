// Full (unmasked) multi-head self-attention for utterances of up to 256 encoder frames -- the shape of the offline
// ConformerCTC path (T = 250 at 10 s).  Reference: asr/models/layers/multihead_attention.py:151-188 (scores =
// q k^T with q pre-scaled by 1/sqrt(hs), softmax over keys, no mask, no positional term), called from
// conformer_blocks.py:164-170.
//
// One workgroup = 8 waves = 8 query tiles (128 queries) of one (utterance, head); two workgroups share a CU
// (2 x 74 KB of LDS), i.e. 4 waves per SIMD, so that one wave's staging / softmax / LDS latency is covered by
// the other waves' MFMAs.  K (row-major) and V (transposed)
// of that head are staged in LDS once per workgroup -- the eight waves read every K/V fragment from there instead
// of each streaming them from L2 with 15 vector-memory instructions per key tile (attention_kernel in blocks.hip,
// which stays for band attention, T > 256 and head size 64).  Per wave:
//   S^T[key][query] = K Q^T        9 MFMAs per 16-key tile (hs = 36 = 9 k-steps of 4), scores of all <=256 keys
//                                  stay in registers (16 x float4)
//   softmax over keys              one pass: max, exp2 (log2 e folded into q), sum -- no running rescale
//   O^T[feat][query] = V^T P^T     12 MFMAs per key tile; P^T is the S^T accumulator itself (same lane mapping)
// The 48-wide feature padding of O (3 tiles for 36 features) is the only wasted matrix work.
#include <cstdlib>

#include "common.h"
#include "launch.h"

namespace {

constexpr int TP_MAX = 272;    // most keys held in LDS: 256 (offline 10 s), or 272 at head size 64 (the streaming CTC's 260 frames)
constexpr int AW = 8;          // waves (query tiles) per workgroup
constexpr int ATH = AW * 64;

template <int HS, int TP>       // TP = padded key count held in LDS
__global__ __launch_bounds__(ATH, 2) void attention_lds_kernel(AttnArgs a) {
  constexpr int VS = TP + 4;           // row stride of V^T (floats): 16-byte aligned rows, 2-way bank conflicts at most
  constexpr int FB = HS / 16;          // full 16-wide feature blocks
  constexpr int TS = (HS % 16) / 4;    // tail k-steps (feature = 16*FB + 4*ts + g)
  constexpr int OT = (HS + 15) / 16;   // output feature tiles
  constexpr int C4 = HS / 4;           // float4 chunks per row
  constexpr int NKT = TP / 16;
  __shared__ __attribute__((aligned(16))) float Ks[TP * HS];
  __shared__ __attribute__((aligned(16))) float Vt[HS * VS];

  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int qt = blockIdx.x * AW + (threadIdx.x >> 6);
  const int T = a.Tk, TQ = a.Tq;       // keys / queries per utterance
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = a.ldk, D = a.D;
  const float* __restrict__ kbase = a.k + (size_t)b * T * ld + h * HS;
  const float* __restrict__ vbase = a.v + (size_t)b * T * ld + h * HS;

  // ---- this lane's query fragment (log2 e folded in: softmax uses exp2)
  const int tq = qt * 16 + c;
  const float* qrow = a.q + ((size_t)b * TQ + min(tq, TQ - 1)) * a.ldq + h * HS;
  constexpr float LOG2E = 1.4426950408889634f;
  f32x4 q4[FB > 0 ? FB : 1];
  float qs[TS > 0 ? TS : 1];
#pragma unroll
  for (int s = 0; s < FB; ++s) q4[s] = ldg4(qrow + 16 * s + g4) * splat4(LOG2E);
#pragma unroll
  for (int s = 0; s < TS; ++s) qs[s] = qrow[16 * FB + 4 * s + g] * LOG2E;

  // ---- stage K and V^T of this (utterance, head); rows past T are zero
  // (all loads first, then the LDS writes: a load -> write loop would expose one L2 round trip per iteration)
  constexpr int NIT = (TP * C4 + ATH - 1) / ATH;
  f32x4 kv[NIT], vv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = min((int)threadIdx.x + it * ATH, TP * C4 - 1);
    const int key = idx / C4, ch = idx - key * C4;
    const size_t row = (size_t)min(key, T - 1) * ld + 4 * ch;
    kv[it] = ldg4(kbase + row);
    vv[it] = ldg4(vbase + row);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = min((int)threadIdx.x + it * ATH, TP * C4 - 1);    // tail threads rewrite the last chunk
    const int key = idx / C4, ch = idx - key * C4;
    const bool ok = key < T;
    const f32x4 k = ok ? kv[it] : splat4(0.f), v = ok ? vv[it] : splat4(0.f);
    *reinterpret_cast<f32x4*>(&Ks[key * HS + 4 * ch]) = k;
    Vt[(4 * ch + 0) * VS + key] = v.x;
    Vt[(4 * ch + 1) * VS + key] = v.y;
    Vt[(4 * ch + 2) * VS + key] = v.z;
    Vt[(4 * ch + 3) * VS + key] = v.w;
  }
  __syncthreads();
  if (qt * 16 >= TQ) return;
  const int nkt = (T + 15) / 16;       // key tiles that hold at least one real key (uniform)

  // ---- S^T = K Q^T
  f32x4 sc[NKT];
  const float* kl = Ks + c * HS + g4;            // + 16*kt*HS + 16*f
  const float* kt_tail = Ks + c * HS + 16 * FB + g;
  auto qk_tile = [&](int kt) {
    f32x4 k4[FB > 0 ? FB : 1];
    float ks[TS > 0 ? TS : 1];
#pragma unroll
    for (int f = 0; f < FB; ++f) k4[f] = *reinterpret_cast<const f32x4*>(kl + 16 * kt * HS + 16 * f);
#pragma unroll
    for (int f = 0; f < TS; ++f) ks[f] = kt_tail[16 * kt * HS + 4 * f];
#pragma unroll
    for (int f = 0; f < FB; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) sc[kt] = mfma4(k4[f][j], q4[f][j], sc[kt]);
#pragma unroll
    for (int f = 0; f < TS; ++f) sc[kt] = mfma4(ks[f], qs[f], sc[kt]);
  };
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) sc[kt] = splat4(0.f);
  // the common case (T > 240) is straight-line code, so the LDS reads of later tiles are scheduled under the MFMAs
  // of earlier ones; shorter utterances take the branchy path (tiles past nkt are skipped)
  const bool full = (nkt == NKT);
  if (full) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) qk_tile(kt);
  } else {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt < nkt) qk_tile(kt);
  }
  // lane holds S^T[key = 16*kt + 4*g + j][query c] (in log2 units)

  // ---- softmax over keys: keys >= T are masked, which only the last real tile can contain
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt < nkt) {
      if (16 * kt + 16 > T) {
        const int kb = 16 * kt + g4;
        sc[kt].x = (kb + 0 < T) ? sc[kt].x : -INFINITY;
        sc[kt].y = (kb + 1 < T) ? sc[kt].y : -INFINITY;
        sc[kt].z = (kb + 2 < T) ? sc[kt].z : -INFINITY;
        sc[kt].w = (kb + 3 < T) ? sc[kt].w : -INFINITY;
      }
      mx = fmaxf(mx, fmaxf(fmaxf(sc[kt].x, sc[kt].y), fmaxf(sc[kt].z, sc[kt].w)));
    }
  }
  mx = group_max(mx);                  // every query sees key 0, so mx is finite
  float psum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt < nkt) {
      sc[kt].x = __builtin_amdgcn_exp2f(sc[kt].x - mx);
      sc[kt].y = __builtin_amdgcn_exp2f(sc[kt].y - mx);
      sc[kt].z = __builtin_amdgcn_exp2f(sc[kt].z - mx);
      sc[kt].w = __builtin_amdgcn_exp2f(sc[kt].w - mx);
      psum += (sc[kt].x + sc[kt].y) + (sc[kt].z + sc[kt].w);
    }
  }

  // ---- O^T[feat][query] += V^T[feat][key] P^T[key][query]
  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = splat4(0.f);
  const float* vl[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) vl[i] = Vt + min(16 * i + c, HS - 1) * VS + g4;   // rows >= HS: results discarded
  auto pv_tile = [&](int kt) {
    f32x4 v4[OT];
#pragma unroll
    for (int i = 0; i < OT; ++i) v4[i] = *reinterpret_cast<const f32x4*>(vl[i] + 16 * kt);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < OT; ++i) o[i] = mfma4(v4[i][j], sc[kt][j], o[i]);
  };
  if (full) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) pv_tile(kt);
  } else {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt < nkt) pv_tile(kt);
  }
  const float inv = 1.0f / group_sum(psum);
  if (tq < TQ) {
    float* orow = a.ctx + ((size_t)b * TQ + tq) * D + h * HS;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      if (16 * i + g4 < HS) stg4(orow + 16 * i + g4, o[i] * splat4(inv));
    }
  }
}

}  // namespace

bool attention_lds_applicable(int HS, const AttnArgs& a) {
  // head size 64 (ConformerM / L): K and V^T take 130 KB, one workgroup per CU.
  return (HS == 36 || HS == 64) && a.win_front < 0 && a.Tk <= (HS == 64 ? TP_MAX : 256) && a.Tk > 16 && a.Tq > 16;
}

int launch_attention_lds(int HS, const AttnArgs& a, hipStream_t s) {
  if (!attention_lds_applicable(HS, a)) return -1;
  const int qtiles = (a.Tq + 15) / 16;
  dim3 grid((qtiles + AW - 1) / AW, a.H, a.B);
  if (HS == 64 && a.Tk > 256) hipLaunchKernelGGL((attention_lds_kernel<64, TP_MAX>), grid, dim3(ATH), 0, s, a);
  else if (HS == 64) hipLaunchKernelGGL((attention_lds_kernel<64, 256>), grid, dim3(ATH), 0, s, a);
  else hipLaunchKernelGGL((attention_lds_kernel<36, 256>), grid, dim3(ATH), 0, s, a);
  return 0;
}
