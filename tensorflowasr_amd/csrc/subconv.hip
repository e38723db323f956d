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
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "env.h"
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


// ---------------------------------------------------------------------------------------------------------
// Split-bf16 variant (default): the same implicit GEMM on v_mfma_f32_16x16x32_bf16 with every fp32 operand written as
// the exact sum of three bf16 terms (see leaf.hip: the six term pairs with i + j <= 2, smallest first, are as accurate
// as an fp32 FMA chain, at 6 x 16 cycles per 32 k-slots instead of 8 x 32), and -- because the matrix pipe is then
// 2.7x faster than any per-wave weight stream can feed -- the conv2 weights shared by the eight waves of a workgroup
// through a double-buffered LDS slab instead of one register stream per wave.
//   * workgroup = 256 consecutive positions (t2, f2) of one utterance = 8 waves x 2 row tiles; grid (ceil(T2 F2 / 256), B)
//     (B = 64, T2 F2 = 5000: 1280 workgroups = exactly five per CU);
//   * the mel patch of the tile (59 frames x 83 bins) is staged in LDS once; conv1 is evaluated from it per lane;
//   * k-slots: one 32-wide MFMA step = (channel block of 16, tap pair (2 p, 2 p + 1)): lane group g holds conv1
//     channels 4 g .. 4 g + 3 at tap 2 p (slots 0..3) and at tap 2 p + 1 (slots 4..7); the ninth tap is paired with
//     zeros (45 steps for 40.5 steps' worth of K: 10 % idle slots);
//   * per step a 27 KB weight slab (9 column tiles x 3 terms) goes global -> registers -> LDS behind the MFMAs of the
//     previous step; waves 0..3 compute the conv1 values + split of step s before its MFMAs, waves 4..7 (their SIMD
//     partners) those of step s + 1 after the MFMAs of step s: one wave's VALU phase always faces the other's MFMAs;
//   * the split is by truncation (x & 0xffff0000, remainder exact), packed with v_perm_b32: 11 VALU per value pair.
constexpr int SCW = 8, SCT = SCW * 64, SRT = 2, SPOSG = SCW * 16 * SRT;   // 256 positions per workgroup
constexpr int NPAIR = 4;                           // regular MFMA steps per channel block of 16: taps (0,1) (2,3) (4,5) (6,7)
// round 3: the ninth taps of TWO channel blocks share one step (slots 0..3 = tap 8 of block 2 i, 4..7 = tap 8 of block
// 2 i + 1): 4 KB + ceil(KB / 2) steps instead of 5 KB (dmodel 144: 41 instead of 45 -- the round-2 kernel paired every ninth
// tap with zeros)
constexpr int ninth_steps(int kb) { return (kb + 1) / 2; }
constexpr int MELP = 8192;                         // floats of LDS for the mel patch

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Round 5: the weight slabs of subconv_split_ring_kernel travel through a ring of SRING LDS slots, fetched SRING - 1 steps
// ahead.  With the two-term operands a step's work (conv1 VALU + 54 MFMAs per wave) had shrunk below the latency of one slab's
// DMA (~1.3 us when every workgroup asks L2 for the same lines), and with a double buffer -- slab s + 1 requested in step s and
// waited for at its end -- every step lasted (conv1 phase) + (DMA latency): 3600 cycles for 2 x 864 of matrix work per SIMD.
// -DMI355ASR_SUBCONV_RING=2 builds the double buffer (A / B timing).
// MI355ASR_SUBCONV_C1I (with conv1 on the matrix pipe): 1 = every wave produces the operand of step s + 1 in pieces between the
// fragment groups of step s's MFMAs (no early / late split); 0 = the operand as its own phase (waves 0..3 before, 4..7 behind).
#ifndef MI355ASR_SUBCONV_C1I
#define MI355ASR_SUBCONV_C1I 1
#endif
#ifndef MI355ASR_SUBCONV_RING
#define MI355ASR_SUBCONV_RING 4
#endif
constexpr int SRING = MI355ASR_SUBCONV_RING;
static_assert(SRING >= 2 && (SRING & (SRING - 1)) == 0, "ring slots: a power of two");
// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: the slab DMAs are the only vector-memory operations in flight in the step
// loop and return in order, so "at most n outstanding" = "everything but the newest n pieces has landed"
DEV void wait_vmcnt(int n) {
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0f72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0f79); break;
    case 12: __builtin_amdgcn_s_waitcnt(0x0f7c); break;
    default: __builtin_amdgcn_s_waitcnt(0x0f70); break;      // any other count: wait for everything (always safe)
  }
}

// 16 bytes per lane, global -> LDS without a register round trip; `lds` is the wave's (uniform) base, lane i writes
// lds + 16 i.  Completion is counted by vmcnt.
DEV void dma16(const u32x4* gsrc, u32x4* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// A weight fragment from the LDS slab, as inline asm: a compiler-visible LDS read makes hipcc wait for vmcnt(0) first
// whenever an LDS-DMA is in flight (it cannot tell the buffer being read from the one being written), which exposed the
// latency of the next slab's DMA in every step.  Waits are counted by hand (LDS returns in order).
template <int OFF>
DEV u32x4 lds_read16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// Round 5: the conv1 evaluation's VALU phase was TWICE the MFMA phase of a step (3600 cycles per step interval against 2 x 864
// of matrix work per SIMD): 36 ds_read_b32 per step, each waited for with lgkmcnt(0) in front of its four v_fmac_f32.  V2:
// the mel patch in plain row-major order so that a lane's window columns are one or two vector reads per row (all of a tap's
// reads issued before its first multiply-add), two v_pk_fma_f32 per mel value instead of four v_fmac_f32, ReLU + zero padding as
// one v_med3_f32, the fp16 lo term as v_fma_mixlo / mixhi_f16.  Same multiply-adds in the same order: bit-identical results.
// -DMI355ASR_CONV1_V2=0 builds the round-3 evaluation (A / B timing).
#ifndef MI355ASR_CONV1_V2
#define MI355ASR_CONV1_V2 1
#endif
#define MI355ASR_CONV1_PK MI355ASR_CONV1_V2
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SplitFrag { u32x4 t[3]; };                  // 8 k-slots x 3 terms

// exact three-term bf16 split of eight fp32 values (slots 0..3 = lo, 4..7 = hi), two values per dword
DEV SplitFrag split8(f32x4 lo, f32x4 hi) {
  float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  SplitFrag f;
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
  return f;
}

// Two-term scheme (round 3): x * scale = hi + lo with hi = fp16(x * scale), lo = fp16(x * scale - hi), both round-to-nearest:
// |x * scale - hi - lo| <= 2^-22 |x * scale| as long as lo is a normal fp16 (|x * scale| >= 2^-2; below that the error is
// 2^-25 absolute -- the scale puts the layer's bound at 2^15).  a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi: three MFMAs instead
// of six, the dropped a_lo b_lo and the representation errors are of the size of the terms the six-product bf16 scheme
// drops (measured: the same distance from the fp64 oracle; Ootomo & Yokota's error-corrected fp16 GEMM without its
// second accumulator, which the power-of-two scales make unnecessary here).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
DEV unsigned pk_f16(float a, float b) {      // v_cvt_pk_f16_f32: both halves round to nearest
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
DEV SplitFrag split8h(f32x4 lo, f32x4 hi) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  SplitFrag f;
  unsigned d0[4], d1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d0[k] = pk_f16(v[2 * k], v[2 * k + 1]);
    [[maybe_unused]] const f16x2 h = __builtin_bit_cast(f16x2, d0[k]);
#if MI355ASR_CONV1_PK
    // lo = fp16(v - hi) as v_fma_mixlo / mixhi_f16: the fp16 hi is read in place (op_sel picks the half), v - hi is formed in
    // fp32 (exact) and rounded once into the destination half -- the same value as v_cvt_f32_f16 + v_sub_f32 + v_cvt_pk_f16_f32,
    // two instructions per pair instead of five
    unsigned lo2;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo2) : "v"(d0[k]), "v"(v[2 * k]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo2) : "v"(d0[k]), "v"(v[2 * k + 1]));
    d1[k] = lo2;
#else
    d1[k] = pk_f16(v[2 * k] - (float)h.x, v[2 * k + 1] - (float)h.y);
#endif
  }
  f.t[0] = u32x4{d0[0], d0[1], d0[2], d0[3]};
  f.t[1] = u32x4{d1[0], d1[1], d1[2], d1[3]};
  f.t[2] = u32x4{0u, 0u, 0u, 0u};
  return f;
}
template <int TM>
DEV SplitFrag split_terms(f32x4 lo, f32x4 hi) {
  if constexpr (TM == 2) return split8h(lo, hi); else return split8(lo, hi);
}
template <int TM>
DEV f32x4 mma_terms(u32x4 w, u32x4 x, f32x4 c) {
  if constexpr (TM == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

// PK: conv1's multiply-adds as v_pk_fma_f32 (operands in aligned register pairs).  dmodel 144 only: the 128-channel-chunk
// kernels of dmodel 256 / 512 hold two weight-fragment groups per step and have no room for the pairs (they would spill).
template <int RTN, bool PK_ = true>
struct SplitLane {
  static constexpr bool PK = PK_;
  int mb[RTN];                // float offset of the lane's 7x7 mel window in the LDS patch, per row tile
  unsigned valid[RTN];        // bit kt*3+kf: conv1 position inside [0,T1) x [0,F1)
};

// conv1 + ReLU (+ conv2's zero padding) at tap q for channels 16 cb + 4 g .. + 3 of row tile rt
// Mel patch layout (round 3): row stride RS.row floats, and inside a row the bins de-interleaved by four -- bin c of the patch
// sits at (c & 3) * RS.seg + (c >> 2).  The sixteen positions of a row tile read bins 4 f2 + m (m = 2 kf + j fixed per
// instruction), i.e. CONSECUTIVE floats f2 + const: no bank conflicts (the plain layout read stride-4 floats: two-way
// conflicts, 25 % of the kernel's LDS cycles); RS.row = 4 seg + pad with 4 RS.row = 20 (mod 32), so that a tile that wraps
// into the next output row (f2: 19 -> 0, + 4 patch rows) continues on the next banks as well.
struct PatchGeom { int row, seg; };
#if MI355ASR_CONV1_V2
// the 3 x 3 mel window of tap Q for one position: plain patch rows, the lane's window starts at bin 4 f2 of its row (16-byte
// aligned); columns 2 kf .. 2 kf + 2 of three rows: kf = 0 -> the first three floats of one b128 / b96, kf = 2 -> of the one
// behind it, kf = 1 -> a b64 + a b32.  Issued as a block: one exposed LDS latency per window instead of nine.
struct Win9 { float m[3][3]; };
template <int Q, int DIAG = 0, class SL>
DEV Win9 conv1_window(const float* melp, PatchGeom RS, const SL& sl, int rt) {
  constexpr int kt = Q / 3, kf = Q % 3;
  int off = sl.mb[rt];
  asm volatile("" : "+v"(off));            // opaque: otherwise the window reads are hoisted out of the channel-block loop
  const float* mp = melp + off + (2 * kt) * RS.row;
  Win9 w;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float* rp = mp + i * RS.row;
    if constexpr (DIAG == 5) {
#pragma unroll
      for (int j = 0; j < 3; ++j) w.m[i][j] = __builtin_bit_cast(float, off + i * RS.row + 2 * kf + j);
    } else if constexpr (kf == 1) {
      const f32x2 a = *reinterpret_cast<const f32x2*>(rp + 2);
      w.m[i][0] = a.x; w.m[i][1] = a.y; w.m[i][2] = rp[4];
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(rp + 2 * kf);
      w.m[i][0] = a.x; w.m[i][1] = a.y; w.m[i][2] = a.z;
    }
  }
  return w;
}
// conv1 + ReLU (+ conv2's zero padding) at tap Q for channels 16 cb + 4 g .. + 3 from the window: the nine multiply-adds per
// channel in the round-3 order (rows, then columns), two channels per v_pk_fma_f32; ReLU and the padding as one v_med3_f32:
// median(v, 0, +inf) = max(v, 0), median(v, 0, 0) = 0
template <int Q, class SL>
DEV f32x4 conv1_eval(const Win9& win, const SL& sl, int rt, const f32x4 (&w1r)[9], f32x4 b1v) {
  f32x2 lo2 = {b1v.x, b1v.y}, hi2 = {b1v.z, b1v.w};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 w = w1r[i * 3 + j];
      const float m = win.m[i][j];
      if constexpr (SL::PK) {
        const f32x2 mm = {m, m};
        lo2 = __builtin_elementwise_fma(mm, f32x2{w.x, w.y}, lo2);
        hi2 = __builtin_elementwise_fma(mm, f32x2{w.z, w.w}, hi2);
      } else {
        lo2.x = __builtin_fmaf(m, w.x, lo2.x); lo2.y = __builtin_fmaf(m, w.y, lo2.y);
        hi2.x = __builtin_fmaf(m, w.z, hi2.x); hi2.y = __builtin_fmaf(m, w.w, hi2.y);
      }
    }
  const bool ok = (sl.valid[rt] >> Q) & 1u;
  const float lim = ok ? __builtin_inff() : 0.f;
  return f32x4{__builtin_amdgcn_fmed3f(lo2.x, 0.f, lim), __builtin_amdgcn_fmed3f(lo2.y, 0.f, lim),
               __builtin_amdgcn_fmed3f(hi2.x, 0.f, lim), __builtin_amdgcn_fmed3f(hi2.y, 0.f, lim)};
}
template <int Q, int DIAG = 0, class SL>
DEV f32x4 conv1_at(const float* melp, PatchGeom RS, const SL& sl, int rt, const f32x4 (&w1r)[9], f32x4 b1v) {
  return conv1_eval<Q>(conv1_window<Q, DIAG>(melp, RS, sl, rt), sl, rt, w1r, b1v);
}
#else
template <int Q, int DIAG = 0, class SL>
DEV f32x4 conv1_at(const float* melp, PatchGeom RS, const SL& sl, int rt, const f32x4 (&w1r)[9], f32x4 b1v) {
  constexpr int kt = Q / 3, kf = Q % 3;
  f32x4 v = b1v;
  int off = sl.mb[rt];
  asm volatile("" : "+v"(off));            // opaque: otherwise the window reads are hoisted out of the channel-block loop
  const float* mp = melp + off + (2 * kt) * RS.row;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int mc = 2 * kf + j, col = (mc & 3) * RS.seg + (mc >> 2);
      const float m = DIAG == 5 ? __builtin_bit_cast(float, off + i * RS.row + col) : mp[i * RS.row + col];
      const f32x4 w = w1r[i * 3 + j];
      v.x = __builtin_fmaf(m, w.x, v.x); v.y = __builtin_fmaf(m, w.y, v.y);
      v.z = __builtin_fmaf(m, w.z, v.z); v.w = __builtin_fmaf(m, w.w, v.w);
    }
  const bool ok = (sl.valid[rt] >> Q) & 1u;
  v.x = ok ? fmaxf(v.x, 0.f) : 0.f; v.y = ok ? fmaxf(v.y, 0.f) : 0.f;
  v.z = ok ? fmaxf(v.z, 0.f) : 0.f; v.w = ok ? fmaxf(v.w, 0.f) : 0.f;
  return v;
}
#endif

template <int PAIR, int DIAG = 0, int TM = 3, int RTN, class SL>
DEV void frags_for(SplitFrag (&xf)[RTN], const float* melp, PatchGeom RS, const SL& sl, const f32x4 (&w1r)[9], f32x4 b1v) {
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) {
    f32x4 lo, hi;
#if MI355ASR_CONV1_V2
    if constexpr (SL::PK) {                // both windows of the row tile requested before the first multiply-add
      const Win9 wlo = conv1_window<2 * PAIR, DIAG>(melp, RS, sl, rt), whi = conv1_window<2 * PAIR + 1, DIAG>(melp, RS, sl, rt);
      lo = conv1_eval<2 * PAIR>(wlo, sl, rt, w1r, b1v);
      hi = conv1_eval<2 * PAIR + 1>(whi, sl, rt, w1r, b1v);
    } else
#endif
    {
      lo = conv1_at<2 * PAIR, DIAG>(melp, RS, sl, rt, w1r, b1v);
      hi = conv1_at<2 * PAIR + 1, DIAG>(melp, RS, sl, rt, w1r, b1v);
    }
    if constexpr (DIAG == 6) {               // no split: the raw bits as three "terms"
      const u32x4 l = __builtin_bit_cast(u32x4, lo), h = __builtin_bit_cast(u32x4, hi);
      xf[rt].t[0] = l; xf[rt].t[1] = h; xf[rt].t[2] = l ^ h;
    } else {
      xf[rt] = split_terms<TM>(lo, hi);
    }
  }
}

// operand of a ninth-tap step: conv1 at tap 8 for channel block cbA (slots 0..3) and cbA + 1 (slots 4..7; zeros past the
// last block); the conv1 taps of the two blocks pass through the same registers one after the other
template <int DIAG, int TM, class LT, int RTN, class SL>
DEV void frags_ninth(SplitFrag (&xf)[RTN], const float* melp, PatchGeom RS, const SL& sl, f32x4 (&w1r)[9], const float* p_b1,
                     int g4, int cbA, int KBn, LT&& load_taps) {
  f32x4 lo[RTN], hi[RTN];
  load_taps(cbA);
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) lo[rt] = conv1_at<8, DIAG>(melp, RS, sl, rt, w1r, lds4(p_b1, cbA, g4));
  if (cbA + 1 < KBn) {
    load_taps(cbA + 1);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = conv1_at<8, DIAG>(melp, RS, sl, rt, w1r, lds4(p_b1, cbA + 1, g4));
  } else {
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = splat4(0.f);
  }
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) xf[rt] = split_terms<TM>(lo[rt], hi[rt]);
}

// ---- conv1 on the matrix pipe (round 5; two-term kernel of dmodel 144: template flag C1M) ----------------------------------------
// The ablations of the two-term kernel (profiles/r05_subconv_ablation.md): MFMA floor 185 us, kernel without its conv1 / split
// VALU work 266 us, with it 390 -- the ~220 VALU instructions per step and wave do NOT hide behind the partner wave's 54 MFMAs
// (a SIMD issues about two VALU instructions per 16-cycle MFMA for free, and needs eight).  conv1 is a 9-tap dot product per
// (position, channel): as a matrix product W1^T [16 channels x K] . window [K x 16 positions] with K = 32 slots = (window row
// i = 0..2 in lane group g = i, eight patch columns 4 f2 .. 4 f2 + 7 per row; group 3 is zero) its result arrives in exactly
// the accumulator layout the conv2 operand is built from (lane (g, c): channels 4 g .. 4 g + 3 of position c).  The mel patch is
// staged ONCE per workgroup as two fp16 planes (hi + lo of mel x 2^m: the two-term representation every other layer uses), so the
// B operand of a tap is one ds_read2_b64 per plane -- the column offset 2 kf of the tap is carried by the A operand: the taps'
// weights (hi + lo of w x 2^w, built per channel block from the staged kernel) sit at slots 2 kf .. 2 kf + 2 of each row, zeros
// elsewhere.  Three MFMAs per (row tile, tap) instead of 36 multiply-adds and nine window reads; the bias enters as the
// accumulator's initial value.  66 MFMAs and ~50 VALU instructions per step and wave instead of 54 and ~220.
// Arithmetic: hi x hi + hi x lo + lo x hi in fp32 -- conv1 is now a two-term product like conv2 (2^-22 of its operand bounds),
// no longer an exact fp32 FMA chain; MI355ASR_SUBCONV_C1M=0 keeps the VALU evaluation.
constexpr int C1_ROWB = 192;                       // bytes of one patch row in a plane: 96 fp16 bins (4 F2 + 4 <= 96)
constexpr int C1_ZERO_ROWS = 8;                    // rows of zeros behind the planes (lane group 3, and rows past a tap's window)
template <int RTN>
struct C1Lane {
  static constexpr bool PK = true;
  unsigned bh[RTN], bl[RTN];                       // LDS byte address of the lane's window row (tap row 0) in the hi / lo plane
  unsigned valid[RTN];
};
template <int KT>
DEV u32x4 c1_read(unsigned addr) {                 // eight fp16 bins of window row 2 KT + g: 16 bytes at an 8-byte aligned address
  u32x4 v;
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(KT * (2 * C1_ROWB / 8)), "n"(KT * (2 * C1_ROWB / 8) + 1));
  return v;
}
struct C1Taps { unsigned ph, qh, pl, ql; };        // (w0, w1), (w2, 0) of this lane's (channel, window row) as fp16 hi and lo
template <int KF>
DEV u32x4 c1_place(unsigned p, unsigned q) {       // the three weights at slots 2 KF .. 2 KF + 2 of the lane's eight
  if constexpr (KF == 0) return u32x4{p, q, 0u, 0u};
  else if constexpr (KF == 1) return u32x4{0u, p, q, 0u};
  else return u32x4{0u, 0u, p, q};
}
DEV f32x4 c1_mma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// ReLU + conv2's zero padding (v_med3_f32) and the step from conv1's accumulator unit (2^m 2^w) to the conv2 operand scale
template <int Q, class SL>
DEV f32x4 c1_finish(f32x4 v, const SL& sl, int rt, float kmul) {
  const float lim = ((sl.valid[rt] >> Q) & 1u) ? __builtin_inff() : 0.f;
  return f32x4{__builtin_amdgcn_fmed3f(v.x, 0.f, lim) * kmul, __builtin_amdgcn_fmed3f(v.y, 0.f, lim) * kmul,
               __builtin_amdgcn_fmed3f(v.z, 0.f, lim) * kmul, __builtin_amdgcn_fmed3f(v.w, 0.f, lim) * kmul};
}
// the operand of a tap-pair step: taps 2 PAIR (k-slots 0..3) and 2 PAIR + 1 (slots 4..7) of the channel block whose taps are in tp
template <int PAIR, int RTN, class SL>
DEV void frags_for_mm(SplitFrag (&xf)[RTN], const SL& sl, const C1Taps& tp, f32x4 c0v, float kmul) {
  constexpr int QA = 2 * PAIR, QB = 2 * PAIR + 1;
  u32x4 wh[RTN][2], wlo[RTN][2];
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) {
    wh[rt][0] = c1_read<QA / 3>(sl.bh[rt]); wlo[rt][0] = c1_read<QA / 3>(sl.bl[rt]);
    wh[rt][1] = c1_read<QB / 3>(sl.bh[rt]); wlo[rt][1] = c1_read<QB / 3>(sl.bl[rt]);
  }
  const u32x4 ah[2] = {c1_place<QA % 3>(tp.ph, tp.qh), c1_place<QB % 3>(tp.ph, tp.qh)};
  const u32x4 al[2] = {c1_place<QA % 3>(tp.pl, tp.ql), c1_place<QB % 3>(tp.pl, tp.ql)};
  f32x4 v[RTN][2];
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) { v[rt][0] = c0v; v[rt][1] = c0v; }
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[rt][0]), "+v"(wlo[rt][0]), "+v"(wh[rt][1]), "+v"(wlo[rt][1]));
  // smallest products first; the 2 RTN accumulators take turns
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) { v[rt][0] = c1_mma(al[0], wh[rt][0], v[rt][0]); v[rt][1] = c1_mma(al[1], wh[rt][1], v[rt][1]); }
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) { v[rt][0] = c1_mma(ah[0], wlo[rt][0], v[rt][0]); v[rt][1] = c1_mma(ah[1], wlo[rt][1], v[rt][1]); }
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) { v[rt][0] = c1_mma(ah[0], wh[rt][0], v[rt][0]); v[rt][1] = c1_mma(ah[1], wh[rt][1], v[rt][1]); }
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt)
    xf[rt] = split8h(c1_finish<QA>(v[rt][0], sl, rt, kmul), c1_finish<QB>(v[rt][1], sl, rt, kmul));
}
// the operand of a ninth-tap step: tap 8 of channel block cbA (slots 0..3) and cbA + 1 (slots 4..7; zeros past the last block)
template <int RTN, class SL, class LT>
DEV void frags_ninth_mm(SplitFrag (&xf)[RTN], const SL& sl, C1Taps& tp, const float* p_b1, int g4, int cbA, int KBn, float kmul, LT&& load_taps) {
  u32x4 wh[RTN], wlo[RTN];
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) { wh[rt] = c1_read<2>(sl.bh[rt]); wlo[rt] = c1_read<2>(sl.bl[rt]); }
  f32x4 lo[RTN], hi[RTN];
  load_taps(cbA);
  {
    const u32x4 ah = c1_place<2>(tp.ph, tp.qh), al = c1_place<2>(tp.pl, tp.ql);
    const f32x4 c0v = lds4(p_b1, cbA, g4);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[rt]), "+v"(wlo[rt]));
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) lo[rt] = c1_mma(al, wh[rt], c0v);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) lo[rt] = c1_mma(ah, wlo[rt], lo[rt]);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) lo[rt] = c1_mma(ah, wh[rt], lo[rt]);
  }
  if (cbA + 1 < KBn) {
    load_taps(cbA + 1);
    const u32x4 ah = c1_place<2>(tp.ph, tp.qh), al = c1_place<2>(tp.pl, tp.ql);
    const f32x4 c0v = lds4(p_b1, cbA + 1, g4);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = c1_mma(al, wh[rt], c0v);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = c1_mma(ah, wlo[rt], hi[rt]);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = c1_mma(ah, wh[rt], hi[rt]);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = c1_finish<8>(hi[rt], sl, rt, kmul);
  } else {
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) hi[rt] = splat4(0.f);
  }
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) xf[rt] = split8h(c1_finish<8>(lo[rt], sl, rt, kmul), hi[rt]);
}

// DIAG != 0: timing experiments only (results are wrong): 1 = no conv1 / split work, 2 = one weight-fragment read per
// step instead of nine, 3 = no slab traffic (global -> LDS), 4 = no barrier in the step loop, 5 = conv1 without its mel
// reads from LDS, 6 = no operand split, 7 = three products per fragment triple instead of six (a two-term operand scheme's MFMA count),
// 8 = no MFMAs (one VALU op per fragment pair instead), 9 = one weight-fragment read per step instead of eighteen.  Round 5: with
// the two-term weights present the variants are those of the two-term kernel.
// DM = dmodel (conv1 channels = conv2 in / out channels), NBW = output column tiles of a workgroup: all nine for dmodel
// 144; eight (128 channels) for 256 / 512, the chunks on grid.z -- conv1 is then recomputed per chunk, the same VALU to
// MFMA ratio per step as at 144.  Weight fragments: [chunk][step][NBW tiles][3 terms][64 lanes][8].
// TM = 3: bf16 terms (exact), six products; TM = 2: fp16 terms, three products (see split8h): a.w2h, conv1 scaled by a.h_scale
// (folded into the staged conv1 kernel and bias: relu(s v) = s relu(v)), accumulators in units of h_scale * h_wscale.
// RTN row tiles of 16 positions per wave: two for whole utterances; one for the streaming shapes (260 positions per chunk: the
// second 256-position workgroup of a chunk would hold four positions and take as long as the first)
template <int DIAG, int DM, int NBW, int TM = 3, int RTN = SRT, bool C1M = false>
__global__ __launch_bounds__(SCT, 2) void subconv_split_ring_kernel(SubConvArgs a, PatchGeom RS, int rows) {
  static_assert(DIAG == 0 || RTN == SRT, "the timing variants run two row tiles per wave");
  static_assert(!C1M || (TM == 2 && (DM == 144 || DM == 256)), "conv1 on the matrix pipe: the two-term kernels of dmodel 144 / 256");
  constexpr int KB = DM / 16, NB = NBW, NI = ninth_steps(KB), NK32 = KB * NPAIR + NI, SLABF = NBW * TM * 64, D = DM;
  const int c0 = blockIdx.z * NBW;               // first output column tile of this workgroup
  __shared__ __attribute__((aligned(16))) u32x4 wl[SRING][SLABF];      // the slab ring: slab s in slot s % SRING
  __shared__ __attribute__((aligned(16))) float melp[MELP];
  __shared__ __attribute__((aligned(16))) float p_w1[9 * D], p_b1[D], p_b2[D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int b = blockIdx.y, r0 = blockIdx.x * (SCW * 16 * RTN), PU = a.T2 * a.F2;
  const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(TM == 2 ? a.w2h : a.w2s) + (size_t)blockIdx.z * NK32 * SLABF;
  float s1 = TM == 2 ? a.h_scale : 1.f;
  if (TM == 2 && a.h_melmax) {                   // features without a static bound: the scale from the UTTERANCE's own maximum
    const float bound = fmaf(a.h_l1, __uint_as_float(a.h_melmax[b]), a.h_bmax);
    const int e = (int)((__float_as_uint(bound) >> 23) & 255u);                // bound in [2^(e - 127), 2^(e - 126))
    s1 = __uint_as_float((unsigned)(127 + min(60, max(-60, 141 - e))) << 23);   // bound * s1 < 2^15
  }
  const float s2 = TM == 2 ? s1 * a.h_wscale : 1.f;
  // C1M: the mel planes carry mel * sm (static bound, or this utterance's run-time maximum), the conv1 kernel w * sw; conv1's
  // accumulator is in units of sm * sw and reaches the conv2 operand scale s1 through kmul (all powers of two)
  float sm = 1.f, kmul = 1.f;
  if constexpr (C1M) {
    sm = a.c1_mscale;
    if (a.h_melmax) {
      const int e = (int)((a.h_melmax[b] >> 23) & 255u);
      sm = __uint_as_float((unsigned)(127 + min(60, max(-60, 141 - e))) << 23);
    }
    kmul = s1 / (sm * a.c1_wscale);
  }
  constexpr int NQ = (SLABF + SCT - 1) / SCT;
  u32x4 nw[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int idx = threadIdx.x + SCT * q;
    nw[q] = u32x4{0u, 0u, 0u, 0u};
    if (idx < SLABF) nw[q] = wg[idx];
  }
  // mel patch: rows tm_base .. + rows, bins fm_base .. + RS (zero outside the utterance / the mel range)
  const int t2a = r0 / a.F2;
  const int tm_base = 4 * t2a - 2 * a.pt2 - a.pt1, fm_base = -2 * a.pf2 - a.pf1;
  const float* __restrict__ mbp = a.mel + (size_t)b * a.F * a.NM;
  const int RSL = 4 * RS.seg;                    // bins per patch row (4 (F2 + 1): the last window reaches bin 4 F2 + 3)
  const unsigned plane = (unsigned)rows * C1_ROWB;                                        // C1M: bytes of one fp16 plane
  const unsigned melp_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)melp;
  if constexpr (C1M) {
    // two fp16 planes, hi then lo, rows of 96 bins (two bins per dword), then C1_ZERO_ROWS rows of zeros
    unsigned* mph = reinterpret_cast<unsigned*>(melp);
    constexpr int RD = C1_ROWB / 4;                // dwords per row
    for (int i = threadIdx.x; i < rows * RD; i += SCT) {
      const int rr = i / RD, jp = i - rr * RD;
      const int tm = tm_base + rr, fm = fm_base + 2 * jp;
      const bool okr = tm >= 0 && tm < a.F;
      const float v0 = (okr && fm >= 0 && fm < a.NM) ? mbp[(size_t)tm * a.NM + fm] * sm : 0.f;
      const float v1 = (okr && fm + 1 >= 0 && fm + 1 < a.NM) ? mbp[(size_t)tm * a.NM + fm + 1] * sm : 0.f;
      const unsigned h = pk_f16(v0, v1);
      const f16x2 hh = __builtin_bit_cast(f16x2, h);
      mph[i] = h;
      mph[rows * RD + i] = pk_f16(v0 - (float)hh.x, v1 - (float)hh.y);
    }
    for (int i = threadIdx.x; i < C1_ZERO_ROWS * RD; i += SCT) mph[2 * rows * RD + i] = 0u;
  } else {
  for (int i = threadIdx.x; i < rows * RSL; i += SCT) {
    const int rr = i / RSL, jj = i - rr * RSL;
    const int tm = tm_base + rr, fm = fm_base + jj;
    melp[MI355ASR_CONV1_V2 ? rr * RS.row + jj : rr * RS.row + (jj & 3) * RS.seg + (jj >> 2)] =
        (tm >= 0 && tm < a.F && fm >= 0 && fm < a.NM) ? mbp[(size_t)tm * a.NM + fm] : 0.f;
  }
  }
  {
    const float ws1 = C1M ? a.c1_wscale : s1, bs1 = C1M ? sm * a.c1_wscale : s1;
    for (int i = threadIdx.x; i < 9 * D; i += SCT) p_w1[i] = a.w1[i] * ws1;
    for (int i = threadIdx.x; i < D; i += SCT) { p_b1[i] = a.b1[i] * bs1; p_b2[i] = a.b2[i] * s2; }
  }
  std::conditional_t<C1M, C1Lane<RTN>, SplitLane<RTN, DM == 144>> sl;
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) {
    const int r = min(r0 + 16 * RTN * wave + 16 * rt + c, PU - 1);
    const int t2 = r / a.F2, f2 = r - t2 * a.F2;
    if constexpr (C1M) {
      // lane group g reads window row g of a tap (g = 3: the zero rows); the tap's first row 2 kt is an immediate offset
      const int g = lane >> 4;
      const unsigned win = melp_addr + (unsigned)(4 * (t2 - t2a) + g) * C1_ROWB + 8u * (unsigned)f2;
      const unsigned zero = melp_addr + 2u * plane;
      sl.bh[rt] = g < 3 ? win : zero;
      sl.bl[rt] = g < 3 ? win + plane : zero;
    } else {
      sl.mb[rt] = 4 * (t2 - t2a) * RS.row + (MI355ASR_CONV1_V2 ? 4 * f2 : f2);
    }
    unsigned vm = 0;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int kf = 0; kf < 3; ++kf) {
        const int t1 = 2 * t2 + kt - a.pt2, f1 = 2 * f2 + kf - a.pf2;
        vm |= (unsigned)((t1 >= 0) & (t1 < a.T1) & (f1 >= 0) & (f1 < a.F1)) << (kt * 3 + kf);
      }
    sl.valid[rt] = vm;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int idx = threadIdx.x + SCT * q;
    if (idx < SLABF) wl[0][idx] = nw[q];
  }
  __syncthreads();

  f32x4 acc[RTN][NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const f32x4 bv = lds4(p_b2, c0 + n, g4);
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) acc[rt][n] = bv;
  }
  f32x4 w1r[C1M ? 1 : 9];
  C1Taps c1t{0u, 0u, 0u, 0u};
  auto load_taps = [&](int cb) {
    if constexpr (C1M) {
      // this lane's A-operand row: channel 16 cb + c, window row g (lane group 3: zeros): three weights, hi and lo
      const int g = lane >> 4;
      const float* wp = p_w1 + 3 * min(g, 2) * D + 16 * cb + c;
      const float w0 = g < 3 ? wp[0] : 0.f, w1 = g < 3 ? wp[D] : 0.f, w2 = g < 3 ? wp[2 * D] : 0.f;
      c1t.ph = pk_f16(w0, w1);
      c1t.qh = pk_f16(w2, 0.f);
      const f16x2 hp = __builtin_bit_cast(f16x2, c1t.ph), hq = __builtin_bit_cast(f16x2, c1t.qh);
      c1t.pl = pk_f16(w0 - (float)hp.x, w1 - (float)hp.y);
      c1t.ql = pk_f16(w2 - (float)hq.x, 0.f);
    } else {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) w1r[tp] = *reinterpret_cast<const f32x4*>(p_w1 + tp * D + 16 * cb + g4);
    }
  };
  auto sl_bh = [&](int rt) { if constexpr (C1M) return sl.bh[rt]; else return 0u; };
  auto sl_bl = [&](int rt) { if constexpr (C1M) return sl.bl[rt]; else return 0u; };
  constexpr bool C1I = C1M && MI355ASR_SUBCONV_C1I;    // interleaved operand production: two operand buffers, step s reads xbuf[s & 1]
  SplitFrag xbuf[C1I ? 2 : 1][RTN];
  SplitFrag (&xa)[RTN] = xbuf[0];
  // the operand of a step through either evaluation, into operand buffer DST
  auto frags_pair = [&](auto PI, int cb, auto DST) {
    constexpr int pair = decltype(PI)::value;
    SplitFrag (&dst)[RTN] = xbuf[C1I ? decltype(DST)::value : 0];
    if constexpr (C1M) frags_for_mm<pair>(dst, sl, c1t, lds4(p_b1, cb, g4), kmul);
    else frags_for<pair, DIAG, TM>(dst, melp, RS, sl, w1r, lds4(p_b1, cb, g4));
  };
  auto frags_9th = [&](int cbA, auto DST) {
    SplitFrag (&dst)[RTN] = xbuf[C1I ? decltype(DST)::value : 0];
    if constexpr (C1M) frags_ninth_mm(dst, sl, c1t, p_b1, g4, cbA, KB, kmul, load_taps);
    else frags_ninth<DIAG, TM>(dst, melp, RS, sl, w1r, p_b1, g4, cbA, KB, load_taps);
  };
  using B0 = std::integral_constant<int, 0>;
  // interleaved mode: the operand of a tap-pair step in pieces, called behind fragment group GI of the previous step's MFMAs.  By
  // the time piece GI + 1 runs, the counted lgkmcnt wait in front of group GI + 1 has covered the window reads piece GI issued
  // (LDS returns in order and they are older than the fragment reads that wait leaves outstanding).
  u32x4 c1w[4];                 // the window fragments of one row tile: taps A / B, hi / lo plane
  f32x4 c1v[2];                 // conv1 accumulators of that row tile
  auto c1_piece = [&](auto PI, auto GI, int cb, auto DST) {
    constexpr int pair = decltype(PI)::value, gi = decltype(GI)::value, QA = 2 * pair, QB = 2 * pair + 1;
    SplitFrag (&dst)[RTN] = xbuf[C1I ? decltype(DST)::value : 0];
    auto reads = [&](int rt) {
      c1w[0] = c1_read<QA / 3>(sl_bh(rt)); c1w[1] = c1_read<QA / 3>(sl_bl(rt));
      c1w[2] = c1_read<QB / 3>(sl_bh(rt)); c1w[3] = c1_read<QB / 3>(sl_bl(rt));
    };
    auto conv = [&]() {
      asm volatile("" : "+v"(c1w[0]), "+v"(c1w[1]), "+v"(c1w[2]), "+v"(c1w[3]));
      const f32x4 c0v = lds4(p_b1, cb, g4);
      const u32x4 ahA = c1_place<QA % 3>(c1t.ph, c1t.qh), alA = c1_place<QA % 3>(c1t.pl, c1t.ql);
      const u32x4 ahB = c1_place<QB % 3>(c1t.ph, c1t.qh), alB = c1_place<QB % 3>(c1t.pl, c1t.ql);
      c1v[0] = c1_mma(alA, c1w[0], c0v); c1v[1] = c1_mma(alB, c1w[2], c0v);
      c1v[0] = c1_mma(ahA, c1w[1], c1v[0]); c1v[1] = c1_mma(ahB, c1w[3], c1v[1]);
      c1v[0] = c1_mma(ahA, c1w[0], c1v[0]); c1v[1] = c1_mma(ahB, c1w[2], c1v[1]);
    };
    auto finish = [&](int rt) { dst[rt] = split8h(c1_finish<QA>(c1v[0], sl, rt, kmul), c1_finish<QB>(c1v[1], sl, rt, kmul)); };
    if constexpr (gi == 0) reads(0);
    else if constexpr (gi == 1) conv();
    else if constexpr (gi == 2) { finish(0); if constexpr (RTN > 1) reads(1); }
    else if constexpr (gi == 3) { if constexpr (RTN > 1) conv(); }
    else if constexpr (gi == 4) { if constexpr (RTN > 1) finish(1); }
  };
  if constexpr (DIAG == 1) {
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt)
#pragma unroll
      for (int t = 0; t < 3; ++t) xa[rt].t[t] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  }
  // which waves run MFMAs first: the SIMD partner of a wave must take the other order (see the step loop)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool late = wv >= SCW / 2;     // (other pairings of early / late waves measured the same or worse: round 2)
  // pieces of one slab this wave fetches (64 lanes x 16 bytes each): wave-uniform, 2 or 3 of the 18 at dmodel 144 / two terms
  int my_pieces = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) my_pieces += (SCT * q + 64 * wv < SLABF) ? 1 : 0;
  auto issue_slab = [&](int t) {       // slab t -> ring slot t % SRING
    if (t < NK32 && DIAG != 3) {
      const u32x4* src = wg + (size_t)t * SLABF;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int w0 = SCT * q + 64 * wv;                        // wave-uniform
        if (w0 < SLABF) dma16(src + w0 + lane, &wl[t & (SRING - 1)][w0]);
      }
    }
  };
  for (int t = 1; t <= SRING - 2; ++t) issue_slab(t);            // slab 0 came through registers; SRING - 2 more in flight
  // the whole step loop once per order (compile-time LATE): with a run-time order inside one loop hipcc keeps both
  // paths' temporaries alive and spills (256 VGPRs + 232 bytes of scratch instead of 194)
  auto run = [&](auto LATE_T) {
  constexpr bool LATE = decltype(LATE_T)::value || C1I;
  if constexpr (LATE) {
    load_taps(0);
    frags_pair(B0{}, 0, B0{});
  }
  // one MFMA step s: slab s is in wl[s & 1]; frags_this() = the operand of this step (early waves, before the MFMAs),
  // frags_next() = the operand of step s + 1 (late waves, after the MFMAs)
  auto step_body = [&](int s, auto&& frags_this, auto&& frags_next, auto PAR_T) {
      const int cur = s & (SRING - 1);
      SplitFrag (&xin)[RTN] = xbuf[(C1M && MI355ASR_SUBCONV_C1I) ? decltype(PAR_T)::value : 0];    // the operand this step's MFMAs read
      // slab s + SRING - 1: global -> LDS directly (global_load_lds_dwordx4: lane i of a wave lands at base + 16 i) into the slot
      // that was last read in step s - 1, whose barrier every wave has passed.
      auto slab_dma = [&]() { issue_slab(s + SRING - 1); };
      // fragments of G column tiles at a time (one with nine tiles: registers; two with eight), the next group requested
      // before the MFMAs of the current one; lgkmcnt(3 G) = "all but the newest group"
      auto mfma_cur = [&](auto&& between) {         // between(GI): called behind the MFMAs of every fragment group
        constexpr int G = NB % 2 == 0 ? 2 : 1, NG = NB / G;
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&wl[cur][lane]);
        u32x4 wa[G][3], wb[G][3];
        auto fetch = [&](u32x4 (&w)[G][3], auto GI) {
          constexpr int gi = decltype(GI)::value;
          static_for<0, G>([&](auto HI) {
            constexpr int h = decltype(HI)::value;
            if constexpr (DIAG == 9 && gi > 0) { w[h][0] = wa[0][0]; w[h][1] = wa[0][1]; if constexpr (TM == 3) w[h][2] = wa[0][2]; return; }
            w[h][0] = lds_read16<((gi * G + h) * TM + 0) * 1024>(base);
            w[h][1] = lds_read16<((gi * G + h) * TM + 1) * 1024>(base);
            if constexpr (TM == 3) w[h][2] = lds_read16<((gi * G + h) * TM + 2) * 1024>(base);
          });
        };
        auto wait = [&](u32x4 (&w)[G][3], auto N_T) {
          constexpr int N = decltype(N_T)::value;
          if constexpr (TM == 2 && G == 1)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0][0]), "+v"(w[0][1]) : "n"(N));
          else if constexpr (TM == 2)
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N));
          else if constexpr (G == 1)
            asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]) : "n"(N));
          else
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[1][0]), "+v"(w[1][1]),
                         "+v"(w[1][2]) : "n"(N));
        };
        auto mma = [&](const u32x4 (&w)[G][3], auto GI) {
          constexpr int gi = decltype(GI)::value;
#pragma unroll
          for (int ord = DIAG == 7 ? 1 : TM - 1; ord >= 0; --ord)   // DIAG 7: without the three products of order 2^-16
#pragma unroll
            for (int p = 0; p <= ord; ++p)
#pragma unroll
              for (int h = 0; h < G; ++h)
#pragma unroll
                for (int rt = 0; rt < RTN; ++rt) {
                  if constexpr (DIAG == 8)        // no matrix work: the operands are consumed by one VALU op per fragment pair
                    acc[rt][gi * G + h] += __builtin_bit_cast(f32x4, w[h][ord - p] ^ xin[rt].t[p]);
                  else
                    acc[rt][gi * G + h] = mma_terms<TM>(w[h][ord - p], xin[rt].t[p], acc[rt][gi * G + h]);
                }
        };
        fetch(wa, std::integral_constant<int, 0>{});
        static_for<0, NG>([&](auto GI) {
          constexpr int gi = decltype(GI)::value;
          if constexpr (gi % 2 == 0) {
            if constexpr (gi + 1 < NG) { fetch(wb, std::integral_constant<int, gi + 1>{}); wait(wa, std::integral_constant<int, TM * G>{}); }
            else wait(wa, std::integral_constant<int, 0>{});
            mma(wa, GI);
            between(GI);
          } else {
            if constexpr (gi + 1 < NG) { fetch(wa, std::integral_constant<int, gi + 1>{}); wait(wb, std::integral_constant<int, TM * G>{}); }
            else wait(wb, std::integral_constant<int, 0>{});
            mma(wb, GI);
            between(GI);
          }
        });
      };
      auto nothing = [](auto) {};
      if constexpr (DIAG == 1) {
        slab_dma();
        mfma_cur(nothing);
      } else if constexpr (C1M && MI355ASR_SUBCONV_C1I) {
        // round 5, conv1 on the matrix pipe: every wave runs the same order -- the operand of step s + 1 is produced BETWEEN the
        // fragment groups of step s's MFMAs (pieces(GI): window reads, conv1 MFMAs, finish), into the other operand buffer
        mfma_cur([&](auto GI) {
          if constexpr (decltype(GI)::value == 0) {
            __builtin_amdgcn_sched_barrier(0);
            slab_dma();
            __builtin_amdgcn_sched_barrier(0);
          }
          frags_this(GI);
        });
        // five pieces per operand (two row tiles); eight column tiles are four fragment groups: the rest behind the last group
        constexpr int NGR = NB / (NB % 2 == 0 ? 2 : 1);
        if constexpr (NGR < 5) static_for<NGR, 5>([&](auto GI) { frags_this(GI); });
        __builtin_amdgcn_sched_barrier(0);
        frags_next();
      } else if constexpr (LATE) {
        // the DMA pieces go out behind the first group of MFMAs: this wave's matrix work starts at once (its partner is in
        // its VALU phase), and by the end of the MFMAs the pieces have landed -- the vmcnt(0) hipcc puts in front of the
        // window reads of frags_next() costs nothing
        mfma_cur([&](auto GI) {
          if constexpr (decltype(GI)::value == 0) {
            __builtin_amdgcn_sched_barrier(0);
            slab_dma();
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
        frags_next();
      } else {
        frags_this();
        __builtin_amdgcn_sched_barrier(0);
        slab_dma();
        __builtin_amdgcn_sched_barrier(0);
        mfma_cur(nothing);
      }
      // this wave's pieces of slab s + 1 have landed when at most the pieces of the slabs behind it are outstanding: those are
      // slabs s + 2 .. min(s + SRING - 1, NK32 - 1); then the barrier: every wave's pieces have, and slot cur is free
      {
        const int behind = min(SRING - 2, NK32 - 2 - s);
        wait_vmcnt(DIAG == 3 ? 0 : max(behind, 0) * my_pieces);
      }
      if constexpr (DIAG != 4) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's LDS reads of the step are done
        __builtin_amdgcn_s_barrier();                          // bare barrier: a fence would wait for the DMAs in flight
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  // operand of the next step (VALU + LDS reads) and the MFMAs of this one are independent.  hipcc puts the ~270
  // VALU instructions in front of the 108 MFMAs, and an in-order wave cannot fill the matrix pipe's shadow that
  // way; since all waves meet at the barrier of every step the two waves of a SIMD would also be in the same phase.
  // So the waves of the upper half of the workgroup (the SIMD partners of the lower half) run the two parts in
  // the opposite order: one wave's conv1 / split VALU work always faces its partner's MFMAs.
#pragma unroll 1
  for (int cb = 0; cb < KB; ++cb) {
    static_for<0, NPAIR>([&](auto PI) {
      constexpr int pair = decltype(PI)::value;
      using PAR = std::integral_constant<int, pair & 1>;             // s = NPAIR cb + pair, NPAIR even
      using NXT = std::integral_constant<int, (pair & 1) ^ 1>;
      if constexpr (C1I) {
        // the next step's operand in pieces behind this step's fragment groups when it is a tap-pair step of the same channel
        // block or the first of the next block (its taps are loaded behind group 0 then)
        step_body(cb * NPAIR + pair,
                  [&](auto GI) {
                    if constexpr (pair + 1 < NPAIR) c1_piece(std::integral_constant<int, pair + 1>{}, GI, cb, NXT{});
                    else if (cb + 1 < KB) {
                      if constexpr (decltype(GI)::value == 0) load_taps(cb + 1);
                      c1_piece(B0{}, GI, cb + 1, NXT{});
                    }
                  },
                  [&]() { if constexpr (pair + 1 == NPAIR) { if (cb + 1 == KB) frags_9th(0, NXT{}); } },
                  PAR{});
      } else {
      step_body(cb * NPAIR + pair,
                [&]() {                      // early waves: the operand of this step, just before its MFMAs
                  if constexpr (pair == 0) load_taps(cb);
                  frags_pair(PI, cb, B0{});
                },
                [&]() {                      // late waves: the operand of the next step, after this step's MFMAs
                  if constexpr (pair + 1 < NPAIR) {
                    frags_pair(std::integral_constant<int, pair + 1>{}, cb, B0{});
                  } else if (cb + 1 < KB) {
                    load_taps(cb + 1);
                    frags_pair(B0{}, cb + 1, B0{});
                  } else {
                    frags_9th(0, B0{});
                  }
                }, B0{});
      }
    });
  }
  if constexpr (C1I) {
    static_assert((KB * NPAIR) % 2 == 0, "the first ninth-tap step reads operand buffer 0");
    static_for<0, NI>([&](auto II) {             // the ninth taps, two channel blocks per step: not interleaved (their taps change mid-step)
      constexpr int i = decltype(II)::value;
      step_body(KB * NPAIR + i, [](auto) {}, [&]() { if constexpr (i + 1 < NI) frags_9th(2 * i + 2, std::integral_constant<int, (i & 1) ^ 1>{}); },
                std::integral_constant<int, i & 1>{});
    });
  } else {
#pragma unroll 1
  for (int i = 0; i < NI; ++i) {             // the ninth taps, two channel blocks per step
    step_body(KB * NPAIR + i,
              [&]() { frags_9th(2 * i, B0{}); },
              [&]() { if (i + 1 < NI) frags_9th(2 * i + 2, B0{}); }, B0{});
  }
  }
  };
  if (late || C1I) run(std::integral_constant<bool, true>{});
  else run(std::integral_constant<bool, false>{});
  const float inv2 = 1.0f / s2;            // a power of two
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) {
    const int r = r0 + 16 * RTN * wave + 16 * rt + c;
    if (r < PU) {
      float* orow = a.out + ((size_t)b * PU + r) * D;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        f32x4 v = acc[rt][n];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if constexpr (TM == 2) v = v * splat4(inv2);
        stg4(orow + 16 * (c0 + n) + g4, v);
      }
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

// split-bf16 kernel for dmodel 144 / 256 / 512; returns -1 when the shape does not fit its LDS mel patch or the dmodel has
// no instantiation (the caller falls back)
template <int DIAG>
static int launch_split_d(int d, const dim3& g144, const SubConvArgs& a, PatchGeom RS, int rows, hipStream_t s, int rtn = SRT) {
  const dim3 g128(g144.x, g144.y, d / 128);
  note_scheme(SCHEME_BF16X3);
  if constexpr (DIAG == 0) {
    // conv1 on the matrix pipe (see C1M): dmodel 144, two-term weights, a mel scale (static or run-time), patch rows of <= 96 bins
    static const bool c1m_env = mi355_env("MI355ASR_SUBCONV_C1M", 1) != 0;
    const bool c1m = c1m_env && a.w2h && (d == 144 || d == 256) && a.c1_wscale > 0.f && (a.c1_mscale > 0.f || a.h_melmax) && 4 * a.F2 + 4 <= C1_ROWB / 2 &&
                     (size_t)(2 * rows + C1_ZERO_ROWS) * C1_ROWB <= sizeof(float) * MELP;
    if (c1m) {
      note_scheme(SCHEME_F16X2);
      if (d == 256) {      // (round 5, the streaming configuration: two 128-channel chunks on grid.z, conv1 evaluated by both)
        if (rtn == 1) hipLaunchKernelGGL((subconv_split_ring_kernel<0, 256, 8, 2, 1, true>), g128, dim3(SCT), 0, s, a, RS, rows);
        else hipLaunchKernelGGL((subconv_split_ring_kernel<0, 256, 8, 2, SRT, true>), g128, dim3(SCT), 0, s, a, RS, rows);
      } else if (rtn == 1) hipLaunchKernelGGL((subconv_split_ring_kernel<0, 144, 9, 2, 1, true>), g144, dim3(SCT), 0, s, a, RS, rows);
      else hipLaunchKernelGGL((subconv_split_ring_kernel<0, 144, 9, 2, SRT, true>), g144, dim3(SCT), 0, s, a, RS, rows);
      return 0;
    }
    if (a.w2h && rtn == 1) {
      note_scheme(SCHEME_F16X2);
      switch (d) {
        case 144: hipLaunchKernelGGL((subconv_split_ring_kernel<0, 144, 9, 2, 1>), g144, dim3(SCT), 0, s, a, RS, rows); return 0;
        case 256: hipLaunchKernelGGL((subconv_split_ring_kernel<0, 256, 8, 2, 1>), g128, dim3(SCT), 0, s, a, RS, rows); return 0;
        case 512: hipLaunchKernelGGL((subconv_split_ring_kernel<0, 512, 8, 2, 1>), g128, dim3(SCT), 0, s, a, RS, rows); return 0;
        default: return -1;
      }
    }
    if (a.w2h) {
      note_scheme(SCHEME_F16X2);
      switch (d) {
        case 144: hipLaunchKernelGGL((subconv_split_ring_kernel<0, 144, 9, 2>), g144, dim3(SCT), 0, s, a, RS, rows); return 0;
        case 256: hipLaunchKernelGGL((subconv_split_ring_kernel<0, 256, 8, 2>), g128, dim3(SCT), 0, s, a, RS, rows); return 0;
        case 512: hipLaunchKernelGGL((subconv_split_ring_kernel<0, 512, 8, 2>), g128, dim3(SCT), 0, s, a, RS, rows); return 0;
        default: return -1;
      }
    }
  }
  if constexpr (DIAG != 0) {           // timing variants of the two-term kernel (dmodel 144) when the two-term weights are there
    if (a.w2h && d == 144) {
      note_scheme(SCHEME_F16X2);
      hipLaunchKernelGGL((subconv_split_ring_kernel<DIAG, 144, 9, 2>), g144, dim3(SCT), 0, s, a, RS, rows);
      return 0;
    }
  }
  switch (d) {
    case 144: hipLaunchKernelGGL((subconv_split_ring_kernel<DIAG, 144, 9>), g144, dim3(SCT), 0, s, a, RS, rows); return 0;
    case 256: hipLaunchKernelGGL((subconv_split_ring_kernel<DIAG, 256, 8>), g128, dim3(SCT), 0, s, a, RS, rows); return 0;
    case 512: hipLaunchKernelGGL((subconv_split_ring_kernel<DIAG, 512, 8>), g128, dim3(SCT), 0, s, a, RS, rows); return 0;
    default: return -1;
  }
}

int launch_subconv_split(int d, const SubConvArgs& a, hipStream_t s) {
  const int PU = a.T2 * a.F2;
  // patch rows hold bins fm_base .. fm_base + 4 (F2 - 1) + 6, de-interleaved by four into segments of F2 + 1 floats; the row
  // stride is padded to 5 (mod 8) floats (see PatchGeom)
  PatchGeom RS;
  RS.seg = a.F2 + 1;
  RS.row = 4 * RS.seg;
#if MI355ASR_CONV1_V2
  // plain rows, windows read as 16-byte vectors at a 16-byte lane stride (conflict-free by construction); a row tile that wraps
  // into the next output row (f2: F2 - 1 -> 0, + 4 patch rows = RS.row 16-byte units) continues on the next banks when
  // RS.row = F2 (mod 8) -- 84 for F2 = 20
  while (RS.row % 4 != 0) ++RS.row;
  if (a.F2 % 4 == 0)                       // (otherwise no multiple of four does it: the wrap then costs a two-way conflict)
    while (RS.row % 8 != a.F2 % 8) RS.row += 4;
#else
  while (RS.row % 8 != 5) ++RS.row;
#endif
  // One row tile per wave (128 positions per workgroup) while that still gives no CU a second workgroup -- single utterances:
  // one to four 10 s utterances per call 1.226 / 1.233 / 1.243 -> 1.197 / 1.202 / 1.208 ms.  (The streaming shapes -- 64 chunks
  // x 260 positions, of whose 2 x 64 x 2 workgroups every second one holds four positions -- do NOT gain: 384 workgroups of 128
  // positions take 159 us where the 256 took 134; a workgroup's time is its 41 steps, not its row tiles.)  Two-term kernels
  // only.  MI355ASR_SUBCONV_RT=1 / 2 forces one.
  static const int rt_env = (int)mi355_env("MI355ASR_SUBCONV_RT", 0);
  const long wg1 = (long)((PU + SPOSG / 2 - 1) / (SPOSG / 2)) * a.B * (d == 144 ? 1 : d / 128);
  const int rtn = (a.w2h && (rt_env == 1 || (rt_env == 0 && wg1 <= 256))) ? 1 : SRT;
  const int posg = SCW * 16 * rtn;
  const int span = (posg - 1 + a.F2 - 1) / a.F2;            // t2 steps a tile can touch beyond its first
  const int rows = 4 * span + 7;
  if ((!a.w2s && !a.w2h) || a.st1 != 2 || rows * RS.row > MELP || PU <= 0) return -1;
  const dim3 grid((PU + posg - 1) / posg, a.B);
#ifdef MI355ASR_DIAG_KERNELS
  // timing-only variants (wrong results), compiled in with -DMI355ASR_DIAG_KERNELS: see the DIAG comment above
  static const int diag = [] {
    const int d = (int)mi355_env("MI355ASR_SUBCONV_DIAG", 0);
    if (d) fprintf(stderr, "libmi355asr: MI355ASR_SUBCONV_DIAG=%d -- timing experiment, the subsampling output is WRONG\n", d);
    return d;
  }();
  if (diag && rtn != SRT) return -1;       // the timing variants are instantiated for two row tiles per wave
  switch (diag) {
    case 1: return launch_split_d<1>(d, grid, a, RS, rows, s);
    case 2: return launch_split_d<2>(d, grid, a, RS, rows, s);
    case 3: return launch_split_d<3>(d, grid, a, RS, rows, s);
    case 4: return launch_split_d<4>(d, grid, a, RS, rows, s);
    case 5: return launch_split_d<5>(d, grid, a, RS, rows, s);
    case 6: return launch_split_d<6>(d, grid, a, RS, rows, s);
    case 7: return launch_split_d<7>(d, grid, a, RS, rows, s);
    case 8: return launch_split_d<8>(d, grid, a, RS, rows, s);
    case 9: return launch_split_d<9>(d, grid, a, RS, rows, s);
    default: break;
  }
#endif
  return launch_split_d<0>(d, grid, a, RS, rows, s, rtn);
}
