// ConvSubsampling convs for dmodel 144 (conformer_blocks.py:76-92):
//   Conv2D(1 -> d, 3x3, stride (st1, 2)) + ReLU  ->  Conv2D(d -> d, 3x3, stride 2) + ReLU
// as one implicit GEMM over K = 9*d with rows = conv2 output positions (b, t2, f2).  Same algorithm as
// subconv_kernel in frontend.hip (conv1 is recomputed in registers from a 7x7 mel window, so the [B,T1,F1,d] conv1
// activation -- 737 MB at B = 64 -- never exists); this version is built on the register weight stream of
// wstream.h:
//   * conv2 weights two batches ahead, SGPR-based addressing;
//   * the conv1 taps / biases live in LDS (one copy per workgroup);
//   * the conv1 evaluation of k-block q+1 (36 FMAs + ReLU + mask) is spread over the five fenced MFMA groups of
//     k-block q, so its VALU work issues in the matrix pipe's shadow instead of between two MFMA batches;
//   * 256-register budget (two waves per SIMD, accumulators stay in VGPRs).
// K is ordered (channel block cb, kt, kf): one set of conv1 taps serves nine k-blocks.
#include "common.h"
#include "launch.h"
#include "wstream.h"

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;   // 9
constexpr int NB = KB;

struct SubCtx {
  float win[7][7];
  unsigned valid;             // bit kt*3+kf: conv1 position (2*t2+kt-pt2, 2*f2+kf-pf2) lies inside [0,T1) x [0,F1)
  int g4;
  const float *p_w1, *p_b1;   // LDS
};

struct SubState {
  f32x4 xf;          // conv1 operand of the next k-block to be multiplied
  f32x4 tw[2][2];    // conv1 taps (LDS -> registers) of the next hook
};

DEV f32x4 rd_tap(const float* w1c, int tp) {
  int off = tp * D;
  asm volatile("" : "+v"(off));          // opaque: otherwise the reads are CSE'd across q and pinned in 36 VGPRs
  return *reinterpret_cast<const f32x4*>(w1c + off);
}
// scalar FMAs on purpose: a float4 * splat is lowered to v_pk_fma_f32 with the window value duplicated into a
// register pair, and hipcc keeps all 49 duplicated pairs live (98 VGPRs -> scratch spills)
DEV void tap(f32x4& v, const SubCtx& cx, int q, int tp, const f32x4 w) {
  const int kt = q / 3, kf = q % 3, i = tp / 3, j = tp % 3;
  float m = cx.win[2 * kt + i][2 * kf + j];
  asm volatile("" : "+v"(m));
  v.x = __builtin_fmaf(m, w.x, v.x); v.y = __builtin_fmaf(m, w.y, v.y);
  v.z = __builtin_fmaf(m, w.z, v.z); v.w = __builtin_fmaf(m, w.w, v.w);
}
DEV f32x4 finish(f32x4 v, const SubCtx& cx, int q) {
  const bool ok = (cx.valid >> q) & 1u;            // conv2's zero padding of the conv1 activation
  v.x = ok ? fmaxf(v.x, 0.f) : 0.f; v.y = ok ? fmaxf(v.y, 0.f) : 0.f;
  v.z = ok ? fmaxf(v.z, 0.f) : 0.f; v.w = ok ? fmaxf(v.w, 0.f) : 0.f;
  return v;
}

// The nine k-blocks (kt, kf) of channel block cb.  Enters with buffer parity CUR0 (weights and taps), leaves with
// CUR0 ^ 1.  The conv1 operand of the k-block after the current one -- (cb, q+1), or (cb+1, 0) during q = 8 -- is
// evaluated in the hooks of the current one: taps two at a time, read from LDS (broadcast reads, 4 addresses per
// wave) one fence group before the FMAs that use them.
template <int CUR0>
DEV void channel_block(f32x4 (&acc)[NB], WStream<NB>& ws, SubState& st, unsigned l16, const SubCtx& cx, int cb,
                       const f32x4* __restrict__ w2) {
  const int cbn = min(cb + 1, KB - 1);
  const float* w1c = cx.p_w1 + 16 * cb + cx.g4;
  const float* w1n = cx.p_w1 + 16 * cbn + cx.g4;
  const f32x4 b1v = lds4(cx.p_b1, cb, cx.g4), b1n = lds4(cx.p_b1, cbn, cx.g4);
  static_for<0, 9>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    constexpr int qn = (q + 1) % 9;                                // k-block whose operand is being prepared
    const int sidx = cb * 9 + q;                                   // batch index in the conv2 weight stream
    const f32x4* p1 = w2 + (size_t)min(sidx + 1, 9 * KB - 1) * (KB * 64);
    const f32x4* p2 = w2 + (size_t)min(sidx + 2, 9 * KB - 1) * (KB * 64);
    f32x4 xn = (q < 8) ? b1v : b1n;
    batch_step<(CUR0 + q) & 1>(acc, st.xf, ws, l16, p1, p2, [&](auto GI) {
      constexpr int gi = decltype(GI)::value;
      constexpr int h = CUR0 + q * 5 + gi;                         // hook counter: its taps sit in tw[h & 1]
      tap(xn, cx, qn, 2 * gi, st.tw[h & 1][0]);
      if constexpr (gi < 4) tap(xn, cx, qn, 2 * gi + 1, st.tw[h & 1][1]);
      if constexpr (gi == 4) xn = finish(xn, cx, qn);
      // fetch the taps of the next hook: same target while gi < 4, else the k-block after it
      constexpr int gn = (gi + 1) % 5;
      const float* wsrc = (q < 7 || (q == 7 && gi < 4)) ? w1c : w1n;
      st.tw[(h + 1) & 1][0] = rd_tap(wsrc, 2 * gn);
      if constexpr (gn < 4) st.tw[(h + 1) & 1][1] = rd_tap(wsrc, 2 * gn + 1);
    });
    st.xf = xn;
  });
}

__global__ __launch_bounds__(BLOCK_THREADS, 2) void subconv144_kernel(SubConvArgs a) {
  __shared__ __attribute__((aligned(16))) float p_w1[9 * D], p_b1[D], p_b2[D];
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int P = a.B * a.T2 * a.F2;

  SubCtx cx;
  cx.g4 = g4; cx.p_w1 = p_w1; cx.p_b1 = p_b1;
  const int pos = wid * 16 + c;
  {
    const int p = min(pos, P - 1);
    const int b = p / (a.T2 * a.F2);
    const int r = p % (a.T2 * a.F2);
    const int t2 = r / a.F2, f2 = r % a.F2;
    const int tm0 = 4 * t2 - 2 * a.pt2 - a.pt1;
    const int fm0 = 4 * f2 - 2 * a.pf2 - a.pf1;
    const float* __restrict__ mb = a.mel + (size_t)b * a.F * a.NM;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int tm = tm0 + i;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int fm = fm0 + j;
        cx.win[i][j] = (tm >= 0 && tm < a.F && fm >= 0 && fm < a.NM) ? mb[(size_t)tm * a.NM + fm] : 0.f;
      }
    }
    cx.valid = 0;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int kf = 0; kf < 3; ++kf) {
        const int t1 = 2 * t2 + kt - a.pt2, f1 = 2 * f2 + kf - a.pf2;
        const unsigned ok = (unsigned)((t1 >= 0) & (t1 < a.T1) & (f1 >= 0) & (f1 < a.F1));
        cx.valid |= ok << (kt * 3 + kf);
      }
  }
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.w2p);
  WStream<NB> ws;
  ws.lane16 = (unsigned)lane * 16u;
  stream_begin(ws, w2, w2 + (size_t)KB * 64);
  stash(p_w1, a.w1, 9 * D); stash(p_b1, a.b1, D); stash(p_b2, a.b2, D);
  __syncthreads();
  if ((size_t)wid * 16 >= (size_t)P) return;

  f32x4 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = lds4(p_b2, n, g4);
  // operand of the very first k-block (channel block 0, tap position 0) and the taps of the first hook
  SubState st;
  {
    const float* w1c = p_w1 + g4;
    f32x4 v = lds4(p_b1, 0, g4);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) tap(v, cx, 0, tp, rd_tap(w1c, tp));
    st.xf = finish(v, cx, 0);
    st.tw[0][0] = rd_tap(w1c, 0);
    st.tw[0][1] = rd_tap(w1c, 1);
  }
  // nine channel blocks of nine batches each; the buffer parity repeats every two blocks
#pragma unroll 1
  for (int cb = 0; cb + 1 < KB; cb += 2) {
    const unsigned l16 = fresh_lane16(ws.lane16);
    channel_block<0>(acc, ws, st, l16, cx, cb, w2);
    channel_block<1>(acc, ws, st, l16, cx, cb + 1, w2);
  }
  {
    const unsigned l16 = fresh_lane16(ws.lane16);
    channel_block<0>(acc, ws, st, l16, cx, KB - 1, w2);
  }
  if (pos < P) {
    float* orow = a.out + (size_t)pos * D;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      f32x4 v = acc[n];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      stg4(orow + 16 * n + g4, v);
    }
  }
}

}  // namespace

int launch_subconv144(const SubConvArgs& a, hipStream_t s) {
  const int P = a.B * a.T2 * a.F2;
  const int tiles = (P + 15) / 16;
  hipLaunchKernelGGL(subconv144_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
