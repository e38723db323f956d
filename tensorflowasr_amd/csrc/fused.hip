// Block-level fused kernels (dmodel 144): whole runs of token-local layers of a ConformerBlock in one launch.
//
// Why: at the benchmark shape one layer is 1-5 GFLOP = 13-57 us, and a pure-MFMA kernel of that size already
// loses ~35 % to launch / ramp-up / drain (tools/ubench/mfma_stream.hip: 4000 waves x 650 MFMAs -> 101 TFLOP/s vs
// 150 for long waves).  Because the transposed-chain layout keeps a token tile's activations in registers across
// any number of GEMMs, the token-local layers between the two layers that mix tokens (attention, depthwise conv)
// can be one kernel each:
//   ff1_qkv_kernel    x0 -> x1 = x0 + fc*FFN1(LN(x0))  ;  qkv = LN(x1) Wqkv (+b), q scaled      (3 GEMMs)
//   out_glu_kernel    x2 = x1 + ctx Wo + bo            ;  u = GLU(LN(x2) Wpw1 + b)                (2 GEMMs)
//   tail_ff2_kernel   x3 = x2 + pw2(swish(BN(dw Wpc + b))) + b ; y = LN(x3 + fc*FFN2(LN(x3)))    (4 GEMMs)
// One wave = 16 tokens and (at 16 000 tokens) one wave per SIMD, so nothing hides a stall: every load a wave
// waits for is matrix-pipe idle time.  Hence
//   * weight fragments are fetched one k-block ahead, interleaved with the MFMAs at tile-pair granularity
//     (2 loads, 8 MFMAs, fenced) -- the micro-benchmark holds 124-138 TFLOP/s with that schedule at this shape;
//   * every small parameter vector of the kernel (biases, LayerNorm gamma/beta, folded BatchNorm) is copied to
//     LDS once per workgroup at kernel start and read from there (~100 cycles, issued a fence group ahead of its
//     use) instead of from L2 (~700 cycles, exposed at each of the 8-10 stage boundaries of a kernel);
//   * bias + activation of hidden tile n+1 runs inside the fenced MFMA region of tile n (v_exp / v_rcp in the
//     matrix pipe's shadow), the residual input stays in registers;
//   * the kernels fit 256 registers (__launch_bounds__(256, 2)), so hipcc keeps the accumulators in VGPRs; with
//     the 512-register budget it parks them in AGPRs and moves them back and forth around every MFMA group.
// Reference semantics: asr/models/conformer_blocks.py:126-134 (FFModule), :164-170 + multihead_attention.py:151-188
// (MHSA), :209-219 (ConvModule), :259-265 (block).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "env.h"
#include "launch.h"
#include "wstream.h"

namespace {

constexpr int D = 144;
constexpr int KB = D / 16;   // 9
constexpr int NB = KB;       // fragments per batch (dmodel 144: every batch on the path is 9 fragments)

// acc[i] += W[t]^T * x[t] for the 9 batches own(0..8) of one GEMM; next0 / next1 = the two batches that follow it
// in the kernel's stream.  Enters with CUR, leaves with CUR^1 (9 is odd).  hook(T, GI) as in batch_step.
template <int CUR, class OWN, class HOOK>
DEV void wave_gemm(f32x4 (&acc)[NB], const f32x4 (&x)[KB], WStream<NB>& s, OWN&& own, const f32x4* __restrict__ next0,
                   const f32x4* __restrict__ next1, HOOK&& hook) {
  const unsigned l16 = fresh_lane16(s.lane16);
  static_for<0, KB>([&](auto T) {
    constexpr int t = decltype(T)::value;
    const f32x4* p1 = (t + 1 < KB) ? own(t + 1) : next0;
    const f32x4* p2 = (t + 2 < KB) ? own(t + 2) : (t + 2 == KB ? next0 : next1);
    batch_step<(CUR + t) & 1>(acc, x[t], s, l16, p1, p2, [&](auto GI) { hook(T, GI); });
  });
}

// Bias / folded-BatchNorm / swish of one hidden tile, split in two so that the LDS reads are issued one fence
// group before the VALU work that consumes them.
template <bool AFF>
struct Act {
  const float *b1, *as, *at;   // LDS
  int g4;
  f32x4 pb, ps, pt;
  DEV void fetch(int tile) {
    pb = lds4(b1, tile, g4);
    if (AFF) { ps = lds4(as, tile, g4); pt = lds4(at, tile, g4); }
  }
  DEV f32x4 apply(f32x4 v) const {
    v = v + pb;
    if (AFF) v = v * ps + pt;
    return swish4(v);
  }
};

// y += W2 act( W1 xin + b1 )  with the hidden dimension swept in chunks of 9 tiles; b1 / aff_* are LDS pointers.
//   AFF: act = swish(s * (. + b1) + t) (folded BatchNorm), else act = swish(. + b1)
// Stream: per chunk 9 batches of W1 (k-block t, hidden tiles h0..h0+8) then 9 batches of W2 (hidden tile h0+t as
// the k-block); next0 / next1 = the two batches after the chain.  Enters and leaves with CUR = 0.
template <int HT, bool AFF>
DEV void wave_chain(f32x4 (&y)[KB], const f32x4 (&xin)[KB], const f32x4* __restrict__ w1, const float* b1,
                    const float* aff_s, const float* aff_t, const f32x4* __restrict__ w2, int g4, WStream<NB>& s,
                    const f32x4* __restrict__ next0, const f32x4* __restrict__ next1) {
  static_assert(HT % NB == 0, "hidden tiles must split evenly");
  Act<AFF> act{b1, aff_s, aff_t, g4, {}, {}, {}};
#pragma unroll 1
  for (int c = 0; c < HT / NB; ++c) {
    const int h0 = c * NB;
    f32x4 h[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) h[i] = splat4(0.f);
    auto own1 = [&](int t) { return w1 + (size_t)(t * HT + h0) * 64; };
    auto own2 = [&](int t) { return w2 + (size_t)((h0 + t) * KB) * 64; };
    // GEMM1: hidden tile 0 is complete after group 0 of the last batch and is activated two groups later
    wave_gemm<0>(h, xin, s, own1, own2(0), own2(1), [&](auto T, auto GI) {
      if constexpr (decltype(T)::value == KB - 1 && decltype(GI)::value == 1) act.fetch(h0);
      if constexpr (decltype(T)::value == KB - 1 && decltype(GI)::value == 2) h[0] = act.apply(h[0]);
    });
    // GEMM2: hidden tile t+1 is activated inside the fenced MFMA region of tile t
    const bool last = (c + 1 == HT / NB);
    const f32x4* a0 = last ? next0 : w1 + (size_t)(0 * HT + h0 + NB) * 64;
    const f32x4* a1 = last ? next1 : w1 + (size_t)(1 * HT + h0 + NB) * 64;
    wave_gemm<1>(y, h, s, own2, a0, a1, [&](auto T, auto GI) {
      constexpr int t = decltype(T)::value;
      if constexpr (t + 1 < NB && decltype(GI)::value == 0) act.fetch(h0 + t + 1);
      if constexpr (t + 1 < NB && decltype(GI)::value == 1) h[t + 1] = act.apply(h[t + 1]);
    });
  }
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

// the same, recomputed from an opaque copy of the thread index: long kernels call this again before their stores instead of
// keeping the 64-bit row offset alive across the whole stream (it was the value the register allocator spilled)
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

// LayerNorm with gamma / beta in LDS
DEV void ln_lds(f32x4 (&xs)[KB], const float* ga, const float* be, int g4, float eps) {
  float mean, rstd;
  ln_stats<KB>(xs, eps, mean, rstd);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = (xs[kb] - splat4(mean)) * splat4(rstd) * lds4(ga, kb, g4) + lds4(be, kb, g4);
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS, 2) void ff1_qkv_kernel(Ff1QkvArgs a) {
  __shared__ __attribute__((aligned(16))) float p_ln1g[D], p_ln1b[D], p_b1[4 * D], p_b2[D], p_ln2g[D], p_ln2b[D], p_qb[3 * D];
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.ff_w1p);
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.ff_w2p);
  const f32x4* wq = reinterpret_cast<const f32x4*>(a.qkv_wp);
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x0 + c.row + 16 * kb + c.g4);
  WStream<NB> ws;
  ws.lane16 = (unsigned)c.lane * 16u;
  stream_begin(ws, w1, w1 + (size_t)(4 * KB) * 64);    // first two batches ride under the stash + LayerNorm
  stash(p_ln1g, a.ff_ln_g, D); stash(p_ln1b, a.ff_ln_b, D); stash(p_b1, a.ff_b1, 4 * D); stash(p_b2, a.ff_b2, D);
  stash(p_ln2g, a.att_ln_g, D); stash(p_ln2b, a.att_ln_b, D); stash(p_qb, a.qkv_b, 3 * D);
  __syncthreads();
  // the residual rides in the accumulator: y = x0/fc + b2 + W2 h, x1 = fc*y (fc = 0.5 in every reference
  // config, so the scaling is exact); keeping x0 in registers across the chain would cost 36 VGPRs
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = lds4(p_b2, kb, c.g4) + splat4(inv_fc) * xs[kb];
  ln_lds(xs, p_ln1g, p_ln1b, c.g4, a.eps);
  wave_chain<4 * KB, false>(y, xs, w1, p_b1, nullptr, nullptr, w2, c.g4, ws, wq, wq + (size_t)(3 * KB) * 64);
  // x1 = x0 + fc * (ffn + b2)
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = splat4(a.fc) * y[i];
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x1 + c.row + 16 * i + c.g4, xs[i]);
  }
  // qkv = LN(x1) Wqkv + b, query tiles scaled
  ln_lds(xs, p_ln2g, p_ln2b, c.g4, a.eps);
  float* qrow = a.qkv + (size_t)min(c.tok, a.M - 1) * (3 * D);
  static_for<0, 3>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    constexpr int qn = q < 2 ? q + 1 : 2;                 // after the last GEMM the stream idles on valid addresses
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = lds4(p_qb, q * KB + i, c.g4);
    wave_gemm<q & 1>(acc, xs, ws, [&](int t) { return wq + (size_t)(t * 3 * KB + q * KB) * 64; },
                     wq + (size_t)(qn * KB) * 64, wq + (size_t)(3 * KB + qn * KB) * 64, NoHook());
    const float sc = (q == 0) ? a.qscale : 1.0f;
    if (c.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(qrow + 16 * (q * KB + i) + c.g4, acc[i] * splat4(sc));
    }
  });
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS, 2) void out_glu_kernel(OutGluArgs a) {
  __shared__ __attribute__((aligned(16))) float p_ob[D], p_lng[D], p_lnb[D], p_pb[2 * D];
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* wo = reinterpret_cast<const f32x4*>(a.out_wp);
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.pw1_wp);
  f32x4 xs[KB], acc[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.ctx + c.row + 16 * kb + c.g4);
  WStream<NB> ws;
  ws.lane16 = (unsigned)c.lane * 16u;
  stream_begin(ws, wo, wo + (size_t)KB * 64);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) acc[kb] = ldg4(a.x1 + c.row + 16 * kb + c.g4);  // residual rides in the accumulator
  stash(p_ob, a.out_b, D); stash(p_lng, a.cv_ln_g, D); stash(p_lnb, a.cv_ln_b, D); stash(p_pb, a.pw1_b, 2 * D);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] += lds4(p_ob, i, c.g4);
  wave_gemm<0>(acc, xs, ws, [&](int t) { return wo + (size_t)(t * KB) * 64; }, wg, wg + (size_t)(2 * KB) * 64, NoHook());
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = acc[i];                                      // x2 = x1 + attention
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x2 + c.row + 16 * i + c.g4, xs[i]);
  }
  ln_lds(xs, p_lng, p_lnb, c.g4, a.eps);
  f32x4 gate[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    acc[i] = lds4(p_pb, i, c.g4);
    gate[i] = lds4(p_pb, KB + i, c.g4);
  }
  wave_gemm<1>(acc, xs, ws, [&](int t) { return wg + (size_t)(t * 2 * KB) * 64; }, wg + (size_t)KB * 64,
               wg + (size_t)(3 * KB) * 64, NoHook());
  wave_gemm<0>(gate, xs, ws, [&](int t) { return wg + (size_t)(t * 2 * KB + KB) * 64; }, wg, wg, NoHook());
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const f32x4 va = acc[i], vb = gate[i];
      f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
      stg4(a.u + c.row + 16 * i + c.g4, o);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// out_glu on the bf16 matrix pipe (fp32 operands as three bf16 terms, see subconv.hip / leaf.hip) with the weights
// shared through LDS.  At 16 000 tokens there is one 16-token row tile per wave and one wave per SIMD, so the fp32
// kernel above is bound by each wave's own MFMA chain (4 x 32 cycles per fragment); six bf16 MFMAs per 32 k-slots are
// 2.7x faster, which a per-wave weight stream from L2 cannot feed -- the four waves of a workgroup read the fragments
// from a double-buffered LDS slab instead, filled by global_load_lds_dwordx4 (no register round trip) one step ahead.
//   step 0..4: out-projection, k-slots = (feature block 2 t | block 2 t + 1) of ctx, 9 column tiles   (27 KB slab)
//   step 5..9: pw_conv_1 on LN(x2), same k-slots, 18 column tiles (value | gate)                    (54 KB slab)
// A lane's B operand for step t is the split of its two accumulator-layout float4 (blocks 2 t, 2 t + 1): the transposed
// chain property of the fp32 kernels carries over.  Column tiles are processed in groups of three (fragments of the
// next group are read from LDS during the 18 MFMAs of the current one; MFMAs on one accumulator are three apart).
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

constexpr int KS32X = 5;                      // 32-wide steps over K = 144

// ---- slab ring: weights as a linear stream of 28 KB slabs (9 column tiles x 3 terms x 64 lanes, padded to 7 x 256
// fragments so that every thread issues exactly 7 DMAs per slab), fetched three slabs ahead into a ring of four LDS
// slots by global_load_lds_dwordx4.  "Slab s + 1 has landed" is the counted wait vmcnt(14) -- valid as long as the
// slab DMAs are the only vector-memory operations in flight (inputs are loaded before the stream starts, outputs are
// stored after it ends).  The fragment reads are inline asm: the compiler would otherwise put vmcnt(0) in front of
// every LDS read that may alias a pending DMA (it tracks LDS-DMA per LDS object; a ring indexed at run time is one
// object), and the fence of __syncthreads() waits for all DMAs too -- hence also the bare s_barrier.
// s_waitcnt vmcnt(PER * ahead) lgkmcnt(0) for a wave-uniform run-time `ahead` in [0, MAXA] (the count is an immediate)
template <int PER, int MAXA>
DEV void wait_dma_ahead(int ahead) {
  static_for<0, MAXA + 1>([&](auto K) {
    constexpr int k = decltype(K)::value, n = PER * k;
    if (ahead == k) __builtin_amdgcn_s_waitcnt(0x0070 | (n & 15) | ((n >> 4) << 14));
  });
}

constexpr int SLB = 7 * BLOCK_THREADS;
template <int RING>
struct SlabStream {
  u32x4_t* ring;              // LDS, RING x SLB fragments
  const u32x4_t* src;         // global slab stream
  int total, wv, lane;
  int s = 0, rd = 0;          // slabs consumed; ring slot of slab s (the slot before it is the one to refill)
  DEV void issue(int slab, int slot) const {
    const u32x4_t* g = src + (size_t)min(slab, total - 1) * SLB + 64 * wv + lane;
    u32x4_t* l = ring + slot * SLB + 64 * wv;
#pragma unroll
    for (int q = 0; q < 7; ++q) dma16(g + BLOCK_THREADS * q, l + BLOCK_THREADS * q);
  }
  DEV void begin() {
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue(i, i);
  }
  // slab s + RING - 1 into the slot read in step s - 1; nothing past the end of the stream (a DMA still in flight when
  // the workgroup ends could land in the LDS of the next workgroup on the CU)
  DEV void prefetch() const {
    if (s + RING - 1 < total) issue(s + RING - 1, rd == 0 ? RING - 1 : rd - 1);
  }
  DEV unsigned cur_addr() const {                        // LDS byte address of this lane's first fragment of slab s
    return (unsigned)(size_t)(__attribute__((address_space(3))) void*)(ring + rd * SLB + lane);
  }
  // After begin() (s = 0): slab 0 and every vector-memory operation issued before begin() are complete; the RING - 2
  // slabs issued after slab 0 may still be in flight.  This wave's LDS operations are done; then the barrier.
  DEV void sync() const {
    wait_dma_ahead<7, RING - 2>(min(RING - 2, max(total - 1, 0)));
    __builtin_amdgcn_s_barrier();
  }
  // end of step s: slab s + 1 has landed.  In flight may be only the slabs issued after it: s + 2 .. min(s + RING - 1,
  // total - 1); in the last steps that is fewer than RING - 2, down to none -- the stream ends with no DMA outstanding.
  DEV void advance() {
    wait_dma_ahead<7, RING - 2>(max(min(RING - 2, total - 2 - s), 0));
    __builtin_amdgcn_s_barrier();
    ++s;
    rd = rd + 1 == RING ? 0 : rd + 1;
  }
};

// parameter stash in two phases: all global loads of a kernel first, then the LDS writes (stash() per array makes one
// L2 round trip per array: load, wait, write)
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

// acc[0..9) += W(slab)^T x: column tiles in groups of three, the fragments of the next group are requested before the
// 18 MFMAs of the current one (LDS returns in order: lgkmcnt(9) = "everything but the nine newest reads")
DEV void slab_step(f32x4* acc, const Split8& xf, unsigned addr) {
  u32x4_t w0[3][3], w1[3][3];
#define SLAB_FETCH(W, GRP) \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) { \
    W[i][0] = lds_read16<((3 * GRP + 0) * 3 + 0) * 1024>(addr + i * 3 * 1024); \
    W[i][1] = lds_read16<((3 * GRP + 0) * 3 + 1) * 1024>(addr + i * 3 * 1024); \
    W[i][2] = lds_read16<((3 * GRP + 0) * 3 + 2) * 1024>(addr + i * 3 * 1024); }
#define SLAB_WAIT(W, N) \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(W[0][0]), "+v"(W[0][1]), "+v"(W[0][2]), "+v"(W[1][0]), "+v"(W[1][1]), \
               "+v"(W[1][2]), "+v"(W[2][0]), "+v"(W[2][1]), "+v"(W[2][2]))
#define SLAB_MMA(W, GRP) \
  _Pragma("unroll") for (int ord = 2; ord >= 0; --ord) \
    _Pragma("unroll") for (int p = 0; p <= ord; ++p) \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) \
        acc[3 * GRP + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, W[i][ord - p]), \
                                                                   __builtin_bit_cast(bf16x8_t, xf.t[p]), acc[3 * GRP + i], 0, 0, 0);
  SLAB_FETCH(w0, 0)
  SLAB_FETCH(w1, 1)
  SLAB_WAIT(w0, 9);
  SLAB_MMA(w0, 0)
  SLAB_FETCH(w0, 2)
  SLAB_WAIT(w1, 9);
  SLAB_MMA(w1, 1)
  SLAB_WAIT(w0, 0);
  SLAB_MMA(w0, 2)
#undef SLAB_FETCH
#undef SLAB_WAIT
#undef SLAB_MMA
}




// ---------------------------------------------------------------------------------------------------------
// Loader-wave versions of the three ring kernels (round 2).  The kernels above make every consumer wave issue its share
// of the slab DMAs: seven global_load_lds per slab and wave, each of which blocks the wave's instruction stream for
// 60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") -- with one wave per SIMD that is matrix-pipe idle
// time, about as much per slab as the 54 MFMAs themselves -- and the counted vmcnt waits forbid any other vector-memory
// operation while the stream runs.  Here a workgroup is eight waves: waves 0-3 are the consumers (16 tokens each, the
// same register-resident chain as before: ds_read_b128 + MFMA + one s_barrier per slab, nothing else), waves 4-7 --
// one per SIMD, beside a consumer -- issue the DMAs, wait for them (their own vmcnt) and meet the consumers at the
// barrier.  512 threads => 256 registers per wave, so the accumulators stay in architectural VGPRs.
// Fragment pipeline of the loader-wave kernels: one register set of nine fragments (3 column tiles x 3 terms) that is
// refilled in place -- the three fragments of a term are re-requested (for the next group) as soon as the MFMAs that
// read them have issued, so a group needs 36 fragment registers instead of the 72 of the double buffer in slab_step.
// Order inside a group: W2 x0 | fetch | W1 x1, W1 x0 | fetch | W0 x2, W0 x1, W0 x0 | fetch  (3 + 6 + 9 MFMAs; the three
// accumulators rotate, so MFMAs on one accumulator are three apart).  LDS returns in order and exactly nine reads are
// outstanding at each wait, hence lgkmcnt(6) = "the three oldest have landed".  The pipeline runs across slab
// boundaries: the last group of slab s prefetches group 0 of slab s + 1, which the loaders guarantee has landed one
// barrier earlier (RingLoader waits for slab s + 2 before the barrier that ends step s).
struct WG3 { u32x4_t w[3][3]; };   // [tile][term]
#define GRP_FETCH(G, GRP, TERM, ADDR) \
  G.w[0][TERM] = lds_read16<((3 * (GRP) + 0) * 3 + (TERM)) * 1024>(ADDR); \
  G.w[1][TERM] = lds_read16<((3 * (GRP) + 1) * 3 + (TERM)) * 1024>(ADDR); \
  G.w[2][TERM] = lds_read16<((3 * (GRP) + 2) * 3 + (TERM)) * 1024>(ADDR);
#define GRP_WAIT(G, TERM) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(G.w[0][TERM]), "+v"(G.w[1][TERM]), "+v"(G.w[2][TERM]))
#define MMA_ASM(ACC, W, X) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, W), __builtin_bit_cast(bf16x8_t, X), ACC, 0, 0, 0)
#define GRP_MMA(ACC, G, TERM, XT) \
  MMA_ASM(ACC[0], G.w[0][TERM], XT); MMA_ASM(ACC[1], G.w[1][TERM], XT); MMA_ASM(ACC[2], G.w[2][TERM], XT);
#define WAIT_MMA_ASM(ACC, G, TERM, XT) GRP_WAIT(G, TERM); MMA_ASM(ACC, G.w[0][TERM], XT)
#define GRP_WAIT_MMA(ACC, G, TERM, XT) \
  WAIT_MMA_ASM(ACC[0], G, TERM, XT); MMA_ASM(ACC[1], G.w[1][TERM], XT); MMA_ASM(ACC[2], G.w[2][TERM], XT);
template <int NGRP>
DEV void grp_step(f32x4* acc, const Split8& xf, WG3& g, unsigned naddr) {
  GRP_WAIT_MMA(acc, g, 2, xf.t[0])
  GRP_FETCH(g, NGRP, 2, naddr)
  GRP_WAIT_MMA(acc, g, 1, xf.t[1])
  GRP_MMA(acc, g, 1, xf.t[0])
  GRP_FETCH(g, NGRP, 1, naddr)
  GRP_WAIT_MMA(acc, g, 0, xf.t[2])
  GRP_MMA(acc, g, 0, xf.t[1])
  GRP_MMA(acc, g, 0, xf.t[0])
  GRP_FETCH(g, NGRP, 0, naddr)
}
DEV void grp_prime(WG3& g, unsigned addr) {
  GRP_FETCH(g, 0, 2, addr)
  GRP_FETCH(g, 0, 1, addr)
  GRP_FETCH(g, 0, 0, addr)
}
// acc[0..9) += W(slab at addr)^T x; leaves the fragments of group 0 of the slab at next_addr in flight
DEV void slab_step_p(f32x4* acc, const Split8& xf, WG3& g, unsigned addr, unsigned next_addr) {
  grp_step<1>(acc, xf, g, addr);
  grp_step<2>(acc + 3, xf, g, addr);
  grp_step<0>(acc + 6, xf, g, next_addr);
}

// ---- VALU under the MFMAs -------------------------------------------------------------------------------------------
// Measured (profiles/r02_ring_experiments.md): with swish and the operand splits removed, tail_ff1 drops from 85 to 68 us --
// hipcc emits that VALU work as clusters between the slabs, where the matrix pipe idles (one consumer wave per SIMD).
// tools/ubench/mfma_fill.hip: behind one v_mfma_f32_16x16x32_bf16 (16.5 cycles) one or two VALU instructions are free
// (16.9 / 17.6 cycles per MFMA), the third costs 4.5 cycles, a second v_exp 8, a ds_read_b128 3.3.  So the work is cut into
// slots of at most two instructions and one transcendental (prep_sched.inc, generated by tools/gen_prep_sched.py) and slot
// k is issued right behind MFMA k of the slab, fenced there with sched_barrier.
// What was tried on the way (profiles/r02_ring_experiments.md): fences per MFMA triple with 5-10 instructions per slot (no
// gain: only ~1-2 hide behind an MFMA); empty asm pins on the slot's registers and the accumulator (hipcc pads each pin
// with s_nop, which costs what the placement saves); the MFMAs themselves as volatile asm statements (clean ISA, but
// 4-6 % slower than the builtin: hipcc then also pads around every asm MFMA).  tools/ubench/slab_stream.hip bounds what
// is left: this stream without DMA / barrier runs 1031 cycles per slab, 1071 with one filler per MFMA, 1332 with two.
struct PrepCtx {
  f32x4 &lo, &hi;                       // the two hidden tiles being prepared (residuals of the split end up here)
  const f32x4 &slo, &tlo, &shi, &thi;   // folded-BatchNorm scale / shift (AFF)
  Split8& out;
  float ta, tb, m0, m1;
};
template <bool AFF, bool FULL> struct PrepSlots;
#include "prep_sched.inc"
#define PREP_PIN(C) __builtin_amdgcn_sched_barrier(0)

// MFMA k of the slab (k = SLOT0 + i), wrapped by the caller's pre(k, acc) / post(k, acc)
template <int TERM, int SLOT0, int I, bool WAIT, class FILL>
DEV void mma_f(f32x4* acc, WG3& g, const u32x4_t& xt, const FILL& f) {
  f.template pre<SLOT0 + I>();
  if constexpr (WAIT) { WAIT_MMA_ASM(acc[I], g, TERM, xt); } else { MMA_ASM(acc[I], g.w[I][TERM], xt); }
  f.template post<SLOT0 + I>();
}
// WAIT: the batch is the first use of its term's fragments (wait folded into its first MFMA)
template <int TERM, int SLOT0, bool WAIT, class FILL>
DEV void grp_mma_f(f32x4* acc, WG3& g, const u32x4_t& xt, const FILL& f) {
  mma_f<TERM, SLOT0, 0, WAIT>(acc, g, xt, f);
  mma_f<TERM, SLOT0, 1, false>(acc, g, xt, f);
  mma_f<TERM, SLOT0, 2, false>(acc, g, xt, f);
}
template <int NGRP, int SLOT0, class FILL>
DEV void grp_step_f(f32x4* acc, const Split8& xf, WG3& g, unsigned naddr, const FILL& pre) {

  grp_mma_f<2, SLOT0 + 0, true>(acc, g, xf.t[0], pre);
  GRP_FETCH(g, NGRP, 2, naddr)
  grp_mma_f<1, SLOT0 + 3, true>(acc, g, xf.t[1], pre);
  grp_mma_f<1, SLOT0 + 6, false>(acc, g, xf.t[0], pre);
  GRP_FETCH(g, NGRP, 1, naddr)
  grp_mma_f<0, SLOT0 + 9, true>(acc, g, xf.t[2], pre);
  grp_mma_f<0, SLOT0 + 12, false>(acc, g, xf.t[1], pre);
  grp_mma_f<0, SLOT0 + 15, false>(acc, g, xf.t[0], pre);
  GRP_FETCH(g, NGRP, 0, naddr)
}
// W(slab)^T x with prep work for (c.lo, c.hi) behind the MFMAs FIRST .. 53: the N slots of the schedule are spread over the
// 54 - FIRST MFMAs, the leading ones doubled when there are more slots than MFMAs
template <bool AFF, bool FULL, int FIRST>
struct PrepFill {
  static constexpr int N = PrepSlots<AFF, FULL>::N, ROOM = 54 - FIRST, DBL = N > ROOM ? N - ROOM : 0;
  static_assert(N <= 2 * ROOM, "prep schedule does not fit");
  PrepCtx& c;
  static constexpr int first_slot(int j) { return j < DBL ? 2 * j : j + DBL; }   // work slot(s) of MFMA FIRST + j
  template <int K>
  DEV void pre() const {                               // MFMA K - 1 had work: its results before this MFMA
    constexpr int j = K - FIRST;
    if constexpr (j >= 1) {
      if constexpr (first_slot(j - 1) < N) PREP_PIN(c);
    }
  }
  template <int K>
  DEV void post() const {
    constexpr int j = K - FIRST;
    if constexpr (j >= 0) {
      constexpr int w0 = first_slot(j);
      if constexpr (w0 < N) {
        PREP_PIN(c);
        prep_slot<w0, AFF, FULL>(c);
        if constexpr (j < DBL && w0 + 1 < N) prep_slot<w0 + 1, AFF, FULL>(c);
      }
    }
  }
};
template <bool AFF, bool FULL, int FIRST>
DEV void slab_step_f(f32x4* acc, const Split8& xf, WG3& g, unsigned addr, unsigned next_addr, PrepCtx& c) {
  const PrepFill<AFF, FULL, FIRST> f{c};
  grp_step_f<1, 0>(acc, xf, g, addr, f);
  grp_step_f<2, 18>(acc + 3, xf, g, addr, f);
  grp_step_f<0, 36>(acc + 6, xf, g, next_addr, f);
  PREP_PIN(c);                                          // the last slot before what follows
}

constexpr int LD_THREADS = 2 * BLOCK_THREADS;

// Measured alternatives (profiles/r02_ring_experiments.md): sending part or all of a slab through VGPRs (global_load ->
// ds_write_b128) instead of global_load_lds is slower (48 -> 50-51 us), i.e. the DMA path is not what limits the stream.
template <int RING>
struct RingLoader {             // waves 4..7
  u32x4_t* ring;
  const u32x4_t *src, *src2;    // slabs [0, n1) from src, [n1, total) from src2
  int n1, total, wv, lane;      // wv = 0..3
  DEV void issue(int slab, int slot) const {
    const u32x4_t* g = (slab < n1 ? src + (size_t)slab * SLB : src2 + (size_t)(slab - n1) * SLB) + 64 * wv + lane;
    u32x4_t* l = ring + slot * SLB + 64 * wv;
#pragma unroll
    for (int q = 0; q < 7; ++q) dma16(g + BLOCK_THREADS * q, l + BLOCK_THREADS * q);
  }
  DEV void run() const {
    static_assert(RING >= 4, "slab s + 1 is read ahead while slab s + RING - 1 is written");
    const int pre = min(RING - 1, total);
    for (int i = 0; i < pre; ++i) issue(i, i);
    wait_dma_ahead<7, RING - 3>(min(RING - 3, max(pre - 2, 0)));   // slabs 0 and 1 have landed
    __builtin_amdgcn_s_barrier();                                  // B0 (consumers: inputs + parameter stash)
    int rd = 0;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
      // consumers read slot rd and prefetch from slot rd + 1; the slot they read in step s - 1 is free since the barrier
      // that ended that step
      if (s + RING - 1 < total) issue(s + RING - 1, rd == 0 ? RING - 1 : rd - 1);
      wait_dma_ahead<7, RING - 3>(max(min(RING - 3, total - 3 - s), 0));     // slab s + 2 has landed
      __builtin_amdgcn_s_barrier();
      rd = rd + 1 == RING ? 0 : rd + 1;
    }
  }
};

template <int RING>
struct RingReader {             // waves 0..3
  u32x4_t* ring;
  int lane;
  int rd = 0;
  DEV void sync() const {       // this wave's LDS writes (parameter stash) are done; B0
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  }
  DEV unsigned slot_addr(int slot) const {
    return (unsigned)(size_t)(__attribute__((address_space(3))) void*)(ring + slot * SLB + lane);
  }
  DEV unsigned cur_addr() const { return slot_addr(rd); }
  DEV unsigned next_addr() const { return slot_addr(rd + 1 == RING ? 0 : rd + 1); }
  DEV void advance() {          // every read of the slot has landed (the last wait of slab_step_p covers them)
    __builtin_amdgcn_s_barrier();
    rd = rd + 1 == RING ? 0 : rd + 1;
    __builtin_amdgcn_sched_barrier(0);   // keeps the operand split of later steps from being hoisted (register pressure)
  }
};

// LDS parameter blocks of the two long kernels (so that the fused tail + ff1 kernel can hold both)
struct Ff1Lds { float ln1g[D], ln1b[D], b1[4 * D], b2[D], ln2g[D], ln2b[D], qb[3 * D]; };
struct TailLds { float pcb[2 * D], bns[2 * D], bnt[2 * D], pw2b[D], lng[D], lnb[D], b1[4 * D], b2[D], fg[D], fb[D]; };

DEV void ff1_stash(Ff1Lds& p, const Ff1QkvArgs& a) {
  const auto r0 = stash_load<D>(a.ff_ln_g), r1 = stash_load<D>(a.ff_ln_b), r3 = stash_load<D>(a.ff_b2),
             r4 = stash_load<D>(a.att_ln_g), r5 = stash_load<D>(a.att_ln_b);
  const auto r2 = stash_load<4 * D>(a.ff_b1);
  const auto r6 = stash_load<3 * D>(a.qkv_b);
  stash_store<D>(p.ln1g, r0); stash_store<D>(p.ln1b, r1); stash_store<4 * D>(p.b1, r2); stash_store<D>(p.b2, r3);
  stash_store<D>(p.ln2g, r4); stash_store<D>(p.ln2b, r5); stash_store<3 * D>(p.qb, r6);
}
DEV void tail_stash(TailLds& p, const TailFf2Args& a) {
  const auto r0 = stash_load<2 * D>(a.pc_b1), r1 = stash_load<2 * D>(a.bn_s), r2 = stash_load<2 * D>(a.bn_t);
  const auto r3 = stash_load<D>(a.pw2_b), r4 = stash_load<D>(a.ff_ln_g), r5 = stash_load<D>(a.ff_ln_b),
             r7 = stash_load<D>(a.ff_b2), r8 = stash_load<D>(a.ln_g), r9 = stash_load<D>(a.ln_b);
  const auto r6 = stash_load<4 * D>(a.ff_b1);
  stash_store<2 * D>(p.pcb, r0); stash_store<2 * D>(p.bns, r1); stash_store<2 * D>(p.bnt, r2); stash_store<D>(p.pw2b, r3);
  stash_store<D>(p.lng, r4); stash_store<D>(p.lnb, r5); stash_store<4 * D>(p.b1, r6); stash_store<D>(p.b2, r7);
  stash_store<D>(p.fg, r8); stash_store<D>(p.fb, r9);
}

// y += W2 act(W1 x + b1) over nch hidden chunks of 9 tiles (10 slabs each); AFF: act = swish(s (.) + t) (folded BatchNorm).
// Activation and operand split run behind the MFMAs (slab_step_f), one pair of hidden tiles per slab:
//   W1 slab 4, MFMAs 18..53 (tiles 0..2 are complete since MFMA 17): prepare (h0, h1) -> operand of W2 slab 0
//   W2 slab t = 0..3: prepare (h[2t+2], h[2t+3]) -> operand of slab t + 1 (the last pair is (h8, zero padding))
template <bool AFF, class ST>
DEV void ring_chain(f32x4 (&y)[KB], const Split8 (&xf)[KS32X], ST& st, WG3& wg, int g4, int nch, const float* b1,
                    const float* as, const float* at) {
#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    f32x4 h[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) h[i] = lds4(b1, ch * KB + i, g4);
    static_for<0, KS32X - 1>([&](auto T) {
      constexpr int t = decltype(T)::value;
      slab_step_p(h, xf[t], wg, st.cur_addr(), st.next_addr());
      st.advance();
    });
    Split8 hf[KS32X];
    f32x4 zero = splat4(0.f);
    // folded-BatchNorm scale / shift of the pair being prepared (fetched right before its slab)
    f32x4 slo = splat4(1.f), tlo = splat4(0.f), shi = splat4(1.f), thi = splat4(0.f);
    auto aff_fetch = [&](int i) {
      if (AFF) {
        slo = lds4(as, ch * KB + i, g4); tlo = lds4(at, ch * KB + i, g4);
        if (i + 1 < KB) { shi = lds4(as, ch * KB + i + 1, g4); thi = lds4(at, ch * KB + i + 1, g4); }
      }
    };
    aff_fetch(0);
    {
      PrepCtx c{h[0], h[1], slo, tlo, shi, thi, hf[0], 0.f, 0.f, 0.f, 0.f};
      slab_step_f<AFF, true, 18>(h, xf[KS32X - 1], wg, st.cur_addr(), st.next_addr(), c);
    }
    st.advance();
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      if constexpr (t == KS32X - 1) {
        slab_step_p(y, hf[t], wg, st.cur_addr(), st.next_addr());
      } else {
        aff_fetch(2 * t + 2);
        if constexpr (2 * t + 3 < KB) {
          PrepCtx c{h[2 * t + 2], h[2 * t + 3], slo, tlo, shi, thi, hf[t + 1], 0.f, 0.f, 0.f, 0.f};
          slab_step_f<AFF, true, 0>(y, hf[t], wg, st.cur_addr(), st.next_addr(), c);
        } else {
          PrepCtx c{h[2 * t + 2], zero, slo, tlo, shi, thi, hf[t + 1], 0.f, 0.f, 0.f, 0.f};
          slab_step_f<AFF, false, 0>(y, hf[t], wg, st.cur_addr(), st.next_addr(), c);
        }
      }
      st.advance();
    });
  }
}

// FFModule 1 + q/k/v projections of the 16 tokens in xs (x0 rows, xs[KB] = 0): 55 slabs; stores x1 and qkv
template <class ST>
DEV void ff1_consume(const Ff1QkvArgs& a, const Ff1Lds& p, const WaveCtx& c, ST& st, WG3& wg, f32x4 (&xs)[KB]) {
  f32x4 y[KB];
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = lds4(p.b2, kb, c.g4) + splat4(inv_fc) * xs[kb];
  ln_lds(xs, p.ln1g, p.ln1b, c.g4, a.eps);
  Split8 xf[KS32X];
#pragma unroll
  for (int t = 0; t < KS32X; ++t) xf[t] = split8(xs[2 * t], 2 * t + 1 < KB ? xs[2 * t + 1 < KB ? 2 * t + 1 : 0] : splat4(0.f));
  ring_chain<false>(y, xf, st, wg, c.g4, 4, p.b1, nullptr, nullptr);
#pragma unroll
  for (int i = 0; i < KB; ++i) { y[i] = splat4(a.fc) * y[i]; xs[i] = y[i]; }        // x1 = x0 + fc * (ffn + b2)
  {
    const WaveCtx e = wave_ctx_fresh(a.M);
    if (e.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(a.x1 + e.row + 16 * i + e.g4, y[i]);
    }
  }
  ln_lds(xs, p.ln2g, p.ln2b, c.g4, a.eps);
#pragma unroll
  for (int t = 0; t < KS32X; ++t) xf[t] = split8(xs[2 * t], 2 * t + 1 < KB ? xs[2 * t + 1 < KB ? 2 * t + 1 : 0] : splat4(0.f));
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {                            // q, k, v: one accumulator set, stored as soon as it is done
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = lds4(p.qb, q * KB + i, c.g4);
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      slab_step_p(acc, xf[t], wg, st.cur_addr(), st.next_addr());
      st.advance();
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

// conv-module tail + FFModule 2 + block-final LayerNorm: xs = dw rows (xs[KB] = 0), y = x2 rows on entry; y = the block's
// output rows on exit.  60 slabs.
template <class ST>
DEV void tail_consume(const TailFf2Args& a, const TailLds& p, const WaveCtx& c, ST& st, WG3& wg, f32x4 (&xs)[KB],
                      f32x4 (&y)[KB]) {
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] += lds4(p.pw2b, i, c.g4);
  Split8 xf[KS32X];
#pragma unroll
  for (int t = 0; t < KS32X; ++t) xf[t] = split8(xs[2 * t], 2 * t + 1 < KB ? xs[2 * t + 1 < KB ? 2 * t + 1 : 0] : splat4(0.f));
  ring_chain<true>(y, xf, st, wg, c.g4, 2, p.pcb, p.bns, p.bnt);
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    xs[i] = y[i];                                                                    // x3 = x2 + conv module
    y[i] = lds4(p.b2, i, c.g4) + splat4(inv_fc) * y[i];                              // x3/fc + b2 (+ W2 h)
  }
  ln_lds(xs, p.lng, p.lnb, c.g4, a.eps);
#pragma unroll
  for (int t = 0; t < KS32X; ++t) xf[t] = split8(xs[2 * t], 2 * t + 1 < KB ? xs[2 * t + 1 < KB ? 2 * t + 1 : 0] : splat4(0.f));
  ring_chain<false>(y, xf, st, wg, c.g4, 4, p.b1, nullptr, nullptr);
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = splat4(a.fc) * y[i];
  ln_lds(y, p.fg, p.fb, c.g4, a.eps);                                                // block-final LayerNorm
}

__global__ __launch_bounds__(LD_THREADS) void ff1_qkv_ld_kernel(Ff1QkvArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[5 * SLB];
  __shared__ __attribute__((aligned(16))) Ff1Lds p;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= WAVES_PER_BLOCK) {
    RingLoader<5>{ring, reinterpret_cast<const u32x4_t*>(a.slabs), nullptr, 55, 55, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  const WaveCtx c = wave_ctx(a.M);
  RingReader<5> st{ring, c.lane};
  f32x4 xs[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x0 + c.row + 16 * kb + c.g4);
  ff1_stash(p, a);
  st.sync();
  WG3 wg;
  grp_prime(wg, st.cur_addr());
  ff1_consume(a, p, c, st, wg, xs);
}

__global__ __launch_bounds__(LD_THREADS) void tail_ff2_ld_kernel(TailFf2Args a) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[5 * SLB];
  __shared__ __attribute__((aligned(16))) TailLds p;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= WAVES_PER_BLOCK) {
    RingLoader<5>{ring, reinterpret_cast<const u32x4_t*>(a.slabs), nullptr, 60, 60, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  const WaveCtx c = wave_ctx(a.M);
  RingReader<5> st{ring, c.lane};
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + c.row + 16 * kb + c.g4);     // residuals ride in the accumulators
  tail_stash(p, a);
  st.sync();
  WG3 wg;
  grp_prime(wg, st.cur_addr());
  tail_consume(a, p, c, st, wg, xs, y);
  const WaveCtx e = wave_ctx_fresh(a.M);
  if (e.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.y + e.row + 16 * i + e.g4, y[i]);
  }
}

// tail of block i + ff_module_1 / qkv of block i + 1 in one launch: 115 slabs from two streams; the block output stays in
// registers (a.y may be null: nothing else reads it), one launch / cold ring fill / input round trip less per block
// (the fixed part of these kernels is ~12 us of ~46)
__global__ __launch_bounds__(LD_THREADS) void tail_ff1_ld_kernel(TailFf2Args a, Ff1QkvArgs b) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[5 * SLB];
  __shared__ __attribute__((aligned(16))) TailLds pt;
  __shared__ __attribute__((aligned(16))) Ff1Lds pf;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= WAVES_PER_BLOCK) {
    RingLoader<5>{ring, reinterpret_cast<const u32x4_t*>(a.slabs), reinterpret_cast<const u32x4_t*>(b.slabs), 60, 115,
                  wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  const WaveCtx c = wave_ctx(a.M);
  RingReader<5> st{ring, c.lane};
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + c.row + 16 * kb + c.g4);
  tail_stash(pt, a);
  ff1_stash(pf, b);
  st.sync();
  WG3 wg;
  grp_prime(wg, st.cur_addr());
  tail_consume(a, pt, c, st, wg, xs, y);
  if (a.y) {
    const WaveCtx e = wave_ctx_fresh(a.M);
    if (e.live) {
#pragma unroll
      for (int i = 0; i < KB; ++i) stg4(a.y + e.row + 16 * i + e.g4, y[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = y[i];
  ff1_consume(b, pf, c, st, wg, xs);
}

__global__ __launch_bounds__(LD_THREADS) void out_glu_ld_kernel(OutGluArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[5 * SLB];
  __shared__ __attribute__((aligned(16))) float p_ob[D], p_lng[D], p_lnb[D], p_pb[2 * D];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= WAVES_PER_BLOCK) {
    RingLoader<5>{ring, reinterpret_cast<const u32x4_t*>(a.og_slabs), nullptr, 15, 15, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  const WaveCtx c = wave_ctx(a.M);
  RingReader<5> st{ring, c.lane};
  f32x4 xs[KB + 1], acc[2 * KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.ctx + c.row + 16 * kb + c.g4);
  xs[KB] = splat4(0.f);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) acc[kb] = ldg4(a.x1 + c.row + 16 * kb + c.g4);    // residual rides in the accumulator
  {
    const auto r0 = stash_load<D>(a.out_b), r1 = stash_load<D>(a.cv_ln_g), r2 = stash_load<D>(a.cv_ln_b);
    const auto r3 = stash_load<2 * D>(a.pw1_b);
    stash_store<D>(p_ob, r0); stash_store<D>(p_lng, r1); stash_store<D>(p_lnb, r2); stash_store<2 * D>(p_pb, r3);
  }
  st.sync();
  WG3 wg;
  grp_prime(wg, st.cur_addr());
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] += lds4(p_ob, i, c.g4);
  static_for<0, KS32X>([&](auto T) {
    constexpr int t = decltype(T)::value;
    const Split8 xf = split8(xs[2 * t], xs[2 * t + 1]);
    slab_step_p(acc, xf, wg, st.cur_addr(), st.next_addr());
    st.advance();
  });
#pragma unroll
  for (int i = 0; i < KB; ++i) xs[i] = acc[i];                                      // x2 = x1 + attention
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.x2 + c.row + 16 * i + c.g4, acc[i]);
  }
  {
    f32x4 (&xr)[KB] = reinterpret_cast<f32x4 (&)[KB]>(xs);
    ln_lds(xr, p_lng, p_lnb, c.g4, a.eps);
  }
#pragma unroll
  for (int i = 0; i < 2 * KB; ++i) acc[i] = lds4(p_pb, i, c.g4);                   // value tiles 0..8, gate tiles 9..17
  static_for<0, KS32X>([&](auto T) {
    constexpr int t = decltype(T)::value;
    const Split8 xf = split8(xs[2 * t], xs[2 * t + 1]);
    slab_step_p(acc, xf, wg, st.cur_addr(), st.next_addr());
    st.advance();
    slab_step_p(acc + KB, xf, wg, st.cur_addr(), st.next_addr());
    st.advance();
  });
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const f32x4 va = acc[i], vb = acc[KB + i];
      f32x4 o = {va.x * fast_sigmoid(vb.x), va.y * fast_sigmoid(vb.y), va.z * fast_sigmoid(vb.z), va.w * fast_sigmoid(vb.w)};
      stg4(a.u + c.row + 16 * i + c.g4, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// CTC class head for dmodel 144 on the slab ring (round 2): logits = x W + b over V classes (1332: 84 column tiles, swept
// in groups of nine like q / k / v above, the last group padded with zero columns), per-frame arg-max fused (first
// maximum wins, as test_asr.py's argmax over the softmax), logits stored only when asked for.  The fp32-MFMA
// gemm_rows_kernel<HEAD> it replaces ran 66-72 us (91 TFLOP/s); this is G x 5 slabs of the same stream design.
// slabs: append_slabs(W padded to 144 G columns, group_major) -- group g = column tiles 9 g .. 9 g + 8, five steps each.
// ---------------------------------------------------------------------------------------------------------
// The bias of the next group is requested BEFORE the current group's logits are stored: loads and stores retire through
// one in-order counter, and a bias request behind nine stores waited for all of them (415 -> 330 us at 12 000 frames x
// 9 160 classes; what remains is the slab stream, 1.66 GB from L2 into LDS -- splitting the classes over workgroups to
// fill the 68 idle CUs of a 188-workgroup launch changed nothing).
__global__ __launch_bounds__(LD_THREADS) void head_ld_kernel(GemmArgs a, const u32x4_t* __restrict__ slabs, int groups) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring[5 * SLB];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g_begin = 0, g_end = groups;
  if (wv >= WAVES_PER_BLOCK) {
    RingLoader<5>{ring, slabs, nullptr, KS32X * groups, KS32X * groups, wv - WAVES_PER_BLOCK, (int)(threadIdx.x & 63)}.run();
    return;
  }
  const WaveCtx c = wave_ctx(a.M);
  RingReader<5> st{ring, c.lane};
  f32x4 xs[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.x + c.row + 16 * kb + c.g4);
  Split8 xf[KS32X];
#pragma unroll
  for (int t = 0; t < KS32X; ++t) xf[t] = split8(xs[2 * t], 2 * t + 1 < KB ? xs[2 * t + 1 < KB ? 2 * t + 1 : 0] : splat4(0.f));
  st.sync();
  WG3 wg;
  grp_prime(wg, st.cur_addr());
  float best_v = -INFINITY;
  int best_i = 0;
  const bool want_max = a.argmax_out != nullptr || a.maxval_out != nullptr;
  auto bias_of = [&](int g, int i) { return KB * g + i < a.NT ? ldg4(a.bias + 16 * (KB * g + i) + c.g4) : splat4(0.f); };
  f32x4 nb[KB];                               // the next group's bias
#pragma unroll
  for (int i = 0; i < KB; ++i) nb[i] = bias_of(g_begin, i);
#pragma unroll 1
  for (int g = g_begin; g < g_end; ++g) {
    // the group's bias rides in the accumulators
    f32x4 acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = nb[i];
    static_for<0, KS32X>([&](auto T) {
      constexpr int t = decltype(T)::value;
      slab_step_p(acc, xf[t], wg, st.cur_addr(), st.next_addr());
      st.advance();
    });
    if (g + 1 < g_end) {
#pragma unroll
      for (int i = 0; i < KB; ++i) nb[i] = bias_of(g + 1, i);
    }
    const WaveCtx e = wave_ctx_fresh(a.M);
    float* yrow = a.y ? a.y + (size_t)e.tok * a.ldy : nullptr;
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int tile = KB * g + i, f0 = 16 * tile + e.g4;
      if (tile < a.NT) {                         // wave-uniform: the bias holds NT tiles; the rest are padding columns
        const f32x4 v = acc[i];
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
  if (!want_max) return;
  // the four lane groups of a token hold disjoint classes: max over the groups, lowest class on ties
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float ov = __shfl_xor(best_v, off);
    const int oi = __shfl_xor(best_i, off);
    if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
  }
  const WaveCtx e = wave_ctx_fresh(a.M);
  if (a.argmax_out && e.live && c.lane < 16) a.argmax_out[e.tok] = best_i;
  if (a.maxval_out && e.live && c.lane < 16) a.maxval_out[e.tok] = best_v;
}



// acc[0..NT) += W(step)^T x  for one 32-wide step: slab = [NT tiles][3 terms][64 lanes] in LDS
template <int NT>
DEV void split_step(f32x4* acc, const Split8& xf, const u32x4_t* slab, int lane) {
  static_assert(NT % 3 == 0, "tiles in groups of three");
  bf16x8_t wf[2][3][3];
  auto fetch = [&](int grp, int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[buf][i][t] = __builtin_bit_cast(bf16x8_t, slab[((3 * grp + i) * 3 + t) * 64 + lane]);
  };
  fetch(0, 0);
#pragma unroll
  for (int grp = 0; grp < NT / 3; ++grp) {
    if (grp + 1 < NT / 3) fetch(grp + 1, (grp + 1) & 1);
#pragma unroll
    for (int ord = 2; ord >= 0; --ord)
#pragma unroll
      for (int p = 0; p <= ord; ++p)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          acc[3 * grp + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[grp & 1][i][ord - p], __builtin_bit_cast(bf16x8_t, xf.t[p]),
                                                                     acc[3 * grp + i], 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Dense(F2 d -> d) of the subsampling (K = 2880, N = 144) on the bf16 pipe with split operands: the long-K layer is where
// a deep DMA ring works -- per 32-wide step a workgroup (4 waves x 16 rows) needs a 27 KB weight slab and its own
// 64 x 32 tile of x; both are fetched by global_load_lds_dwordx4 into a ring of RING buffers, RING - 1 steps ahead, so
// that the ~1-2 us of DMA latency is covered by several steps of MFMAs (one step = 54 MFMAs = 0.4 us; the out_glu
// experiment above had a single step of cover).  Every thread issues exactly SL_DMA DMAs per step (the slab is padded to
// 7 x 256 fragments), so "slab s + 1 has landed" is the counted wait vmcnt((RING - 2) * SL_DMA).
//   x tile in LDS: 16-byte slot (chunk, row) at chunk * 64 + row -- the 16 lanes of a row tile read consecutive slots.
constexpr int SL_RING = 4;
constexpr int SL_WFR = 7 * BLOCK_THREADS;            // weight fragments per slab (1728 used)
constexpr int SL_XFR = 2 * BLOCK_THREADS;            // x slots per slab: 64 rows x 8 chunks of 16 bytes
constexpr int SL_DMA = 9;                            // DMA instructions per thread and step
constexpr int SL_SLOT = SL_WFR + SL_XFR;             // fragments per ring slot (36 KB)


// The same with loader waves (round 2, as the block ring kernels): nine LDS-DMA instructions per wave and step hold the
// issuing wave for about as long as its 54 MFMAs take, and with one wave per SIMD nothing overlaps them.  Waves 4..7 only
// issue the DMAs (the pieces waves 0..3 issued before), wait for "slab s + 1 has landed" and meet the consumers at the
// barrier; waves 0..3 read, split and multiply.
__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void sublinear_split_ld_kernel(StreamGemmArgs a, const u32x4_t* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) u32x4_t ring0[SL_SLOT], ring1[SL_SLOT], ring2[SL_SLOT], ring3[SL_SLOT];
  __shared__ __attribute__((aligned(16))) float p_b[D];
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool loader = wave >= WAVES_PER_BLOCK;
  const int wv = loader ? wave - WAVES_PER_BLOCK : wave;          // piece owner / row tile
  const int ltid = 64 * wv + lane;
  const int row0 = blockIdx.x * 64;
  const int steps = a.K / 32;
  const float* xsrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = ltid + BLOCK_THREADS * q;
    const int row = min(row0 + (p & 63), a.M - 1);
    xsrc[q] = a.x + (size_t)row * a.K + 4 * (p >> 6);
  }
  auto fill = [&](int s, u32x4_t* slot) {             // slab of step s -> ring slot (loader waves)
    const u32x4_t* wsrc = ws + (size_t)s * SL_WFR;
#pragma unroll
    for (int q = 0; q < 7; ++q) dma16(wsrc + BLOCK_THREADS * q + 64 * wv + lane, slot + BLOCK_THREADS * q + 64 * wv);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      dma16(reinterpret_cast<const u32x4_t*>(xsrc[q] + 32 * s), slot + SL_WFR + BLOCK_THREADS * q + 64 * wv);
  };
  if (loader) {
    fill(0, ring0);
    fill(min(1, steps - 1), ring1);
    fill(min(2, steps - 1), ring2);
    constexpr int kWait = 0x0f70 | (((SL_RING - 2) * SL_DMA) & 15) | ((((SL_RING - 2) * SL_DMA) >> 4) << 14);
    __builtin_amdgcn_s_waitcnt(kWait);
  } else {
    for (int i = ltid; i < D; i += BLOCK_THREADS) p_b[i] = a.bias[i];
  }
  __syncthreads();
  if (loader) {
    auto lstep = [&](int s, u32x4_t* refill) {
      if (s + SL_RING - 1 < steps) fill(s + SL_RING - 1, refill);
      wait_dma_ahead<SL_DMA, SL_RING - 2>(max(min(SL_RING - 2, steps - 2 - s), 0));
      __builtin_amdgcn_s_barrier();
    };
    int s = 0;
#pragma unroll 1
    for (; s + 4 <= steps; s += 4) { lstep(s, ring3); lstep(s + 1, ring0); lstep(s + 2, ring1); lstep(s + 3, ring2); }
    if (s < steps) lstep(s++, ring3);
    if (s < steps) lstep(s++, ring0);
    if (s < steps) lstep(s++, ring1);
    return;
  }
  f32x4 acc[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] = lds4(p_b, i, 4 * g);
  auto step = [&](const u32x4_t* cur) {
    const f32x4 lo = __builtin_bit_cast(f32x4, cur[SL_WFR + g * 64 + 16 * wv + c]);
    const f32x4 hi = __builtin_bit_cast(f32x4, cur[SL_WFR + (4 + g) * 64 + 16 * wv + c]);
    const Split8 xf = split8(lo, hi);
    split_step<KB>(acc, xf, cur, lane);
    __builtin_amdgcn_s_barrier();          // the loaders arrive once slab s + 1 has landed; this wave's reads of the slot are done
  };
  int s = 0;
#pragma unroll 1
  for (; s + 4 <= steps; s += 4) { step(ring0); step(ring1); step(ring2); step(ring3); }
  if (s < steps) { step(ring0); ++s; }
  if (s < steps) { step(ring1); ++s; }
  if (s < steps) { step(ring2); ++s; }
  const int tok = row0 + 16 * wv + c;
  if (tok < a.M) {
    float* yrow = a.y + (size_t)tok * a.ldy;
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(yrow + 16 * i + 4 * g, acc[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK_THREADS, 2) void tail_ff2_kernel(TailFf2Args a) {
  __shared__ __attribute__((aligned(16))) float p_pcb[2 * D], p_bns[2 * D], p_bnt[2 * D], p_pw2b[D], p_lng[D], p_lnb[D], p_b1[4 * D],
      p_b2[D], p_fg[D], p_fb[D];
  const WaveCtx c = wave_ctx(a.M);
  const f32x4* wpc = reinterpret_cast<const f32x4*>(a.pc_w1p);
  const f32x4* wp2 = reinterpret_cast<const f32x4*>(a.pw2_wp);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.ff_w1p);
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.ff_w2p);
  f32x4 xs[KB], y[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xs[kb] = ldg4(a.dw + c.row + 16 * kb + c.g4);
  WStream<NB> ws;
  ws.lane16 = (unsigned)c.lane * 16u;
  stream_begin(ws, wpc, wpc + (size_t)(2 * KB) * 64);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) y[kb] = ldg4(a.x2 + c.row + 16 * kb + c.g4);     // residuals ride in the accumulators
  stash(p_pcb, a.pc_b1, 2 * D); stash(p_bns, a.bn_s, 2 * D); stash(p_bnt, a.bn_t, 2 * D); stash(p_pw2b, a.pw2_b, D);
  stash(p_lng, a.ff_ln_g, D); stash(p_lnb, a.ff_ln_b, D); stash(p_b1, a.ff_b1, 4 * D); stash(p_b2, a.ff_b2, D);
  stash(p_fg, a.ln_g, D); stash(p_fb, a.ln_b, D);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] += lds4(p_pw2b, i, c.g4);
  wave_chain<2 * KB, true>(y, xs, wpc, p_pcb, p_bns, p_bnt, wp2, c.g4, ws, w1, w1 + (size_t)(4 * KB) * 64);
  const float inv_fc = 1.0f / a.fc;
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    xs[i] = y[i];                                                                    // x3 = x2 + conv module
    y[i] = lds4(p_b2, i, c.g4) + splat4(inv_fc) * y[i];                              // x3/fc + b2 (+ W2 h): see ff1_qkv
  }
  ln_lds(xs, p_lng, p_lnb, c.g4, a.eps);
  wave_chain<4 * KB, false>(y, xs, w1, p_b1, nullptr, nullptr, w2, c.g4, ws, w1, w1);
#pragma unroll
  for (int i = 0; i < KB; ++i) y[i] = splat4(a.fc) * y[i];
  ln_lds(y, p_fg, p_fb, c.g4, a.eps);                                                // block-final LayerNorm
  if (c.live) {
#pragma unroll
    for (int i = 0; i < KB; ++i) stg4(a.y + c.row + 16 * i + c.g4, y[i]);
  }
}

}  // namespace

// MI355ASR_FF1QKV_RING / MI355ASR_TAILFF2_RING / MI355ASR_OUTGLU_SPLIT = 0: the fp32-MFMA register-stream kernels (exact fp32
// products on v_mfma_f32_16x16x4_f32; what a handle without slab streams runs anyway).  Otherwise, in order: the pair-pipelined
// two-term fp16 kernels (fused_pp.hip; MI355ASR_PP=0 switches them off), then the round-2 loader-wave kernels on three bf16 terms.
// (Rounds 1-2 also had per-wave-DMA ring kernels and a double-buffered out_glu: slower than both, deleted in round 4.)
static bool env_on(const char* name) { return mi355_env(name, 1) != 0; }
int launch_ff1_qkv(const Ff1QkvArgs& a, hipStream_t s) {
  static const bool ring = env_on("MI355ASR_FF1QKV_RING");
  const int tiles = (a.M + 15) / 16;
  if (ring && a.slabs) {
    if (launch_pp_ff1_qkv(a, s) == 0) return 0;
    if (a.pre_pp || a.qkv_T > 0 || a.xq_pe) return -1;   // only the pair-pipelined kernel computes x0 itself (callers ask ff1_pre_selected first) / stores q, k, v head-major / projects an RBlock's query
    note_scheme(SCHEME_BF16X3);
    hipLaunchKernelGGL(ff1_qkv_ld_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a);
    return 0;
  }
  if (a.pre_pp || a.qkv_T > 0 || a.xq_pe) return -1;
  note_scheme(SCHEME_F32);
  hipLaunchKernelGGL(ff1_qkv_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
int launch_out_glu(const OutGluArgs& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  static const bool split = env_on("MI355ASR_OUTGLU_SPLIT");
  if (a.og_slabs && split) {
    if (launch_pp_out_glu(a, s) == 0) return 0;   // the two-term fp16 stream (fused_pp.hip)
    note_scheme(SCHEME_BF16X3);
    hipLaunchKernelGGL(out_glu_ld_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a);
    return 0;
  }
  note_scheme(SCHEME_F32);
  hipLaunchKernelGGL(out_glu_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
// split-bf16 ring-DMA kernel for the subsampling Dense; ws = pack_split32 fragments padded to 1792 per step
int launch_head_ld(const GemmArgs& a, const float* slabs, int groups, hipStream_t s) {
  // MI355ASR_HEAD_RING=0: the fp32-MFMA gemm_rows_kernel<HEAD>
  static const bool on = mi355_env("MI355ASR_HEAD_RING", 1) != 0;
  if (!on || !slabs || groups < 1 || a.NT > KB * groups || a.M <= 0) return -1;
  const int tiles = (a.M + 15) / 16;
#ifdef MI355ASR_DIAG_KERNELS
  static const bool nostore = mi355_env("MI355ASR_HEAD_NOSTORE", 0) != 0;
  if (nostore) {                            // timing only: what the logit stores cost
    GemmArgs t = a;
    t.y = nullptr;
    hipLaunchKernelGGL(head_ld_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, t, reinterpret_cast<const u32x4_t*>(slabs), groups);
    return 0;
  }
#endif
  note_scheme(SCHEME_BF16X3);
  hipLaunchKernelGGL(head_ld_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, reinterpret_cast<const u32x4_t*>(slabs), groups);
  return 0;
}

int launch_sublinear_split(const StreamGemmArgs& a, const float* ws, hipStream_t s) {
  if (a.NT != KB || a.K % 32 != 0 || a.M <= 0 || !ws) return -1;
  note_scheme(SCHEME_BF16X3);
  hipLaunchKernelGGL(sublinear_split_ld_kernel, dim3((a.M + 63) / 64), dim3(2 * BLOCK_THREADS), 0, s, a,
                     reinterpret_cast<const u32x4_t*>(ws));
  return 0;
}
// the conv tail will run on the pair-pipelined kernels (fused_pp.hip): api.hip then folds the depthwise conv into them
bool tail_pp_selected() {
  static const bool ring = env_on("MI355ASR_TAILFF2_RING");
  return ring && pp_enabled();
}
// launch_ff1_qkv will take the pair-pipelined kernel, which can compute x0 from the layer in front (Ff1QkvArgs::pre_*)
bool ff1_pre_selected() {
  static const bool ring = env_on("MI355ASR_FF1QKV_RING");
  return ring && pp_pre_fold_ok();
}
// launch_ff1_qkv / launch_tail_ff1 will produce q, k, v with the pair-pipelined kernel (the only producer of the head-major layout)
bool ff1_qkv_pp_selected(bool has_slabs, bool has_pp) {
  static const bool ring = env_on("MI355ASR_FF1QKV_RING");
  return ring && has_slabs && has_pp && pp_enabled();
}
// tail of one block + ff1_qkv of the next in one launch; -1 when the loader-wave kernels are switched off
bool tail_ff1_available() {
  // MI355ASR_TAIL_FF1=0: separate tail_ff2 / ff1_qkv launches (also whenever one of the two is switched to the fp32 kernels)
  static const bool on = env_on("MI355ASR_TAIL_FF1") && env_on("MI355ASR_TAILFF2_RING") && env_on("MI355ASR_FF1QKV_RING");
  return on;
}
int launch_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, hipStream_t s) {
  if (!tail_ff1_available() || !a.slabs || !b.slabs || a.M != b.M) return -1;
  if (launch_pp_tail_ff1(a, b, s) == 0) return 0;
  if (b.qkv_T > 0) return -1;         // head-major q / k / v: the pair-pipelined kernels only
  const int tiles = (a.M + 15) / 16;
  note_scheme(SCHEME_BF16X3);
  hipLaunchKernelGGL(tail_ff1_ld_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a, b);
  return 0;
}
int launch_tail_ff2(const TailFf2Args& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  static const bool ring = env_on("MI355ASR_TAILFF2_RING");
  if (ring && a.slabs) {
    if (launch_pp_tail_ff2(a, s) == 0) return 0;
    note_scheme(SCHEME_BF16X3);
    hipLaunchKernelGGL(tail_ff2_ld_kernel, dim3((tiles + 3) / 4), dim3(LD_THREADS), 0, s, a);
    return 0;
  }
  note_scheme(SCHEME_F32);
  hipLaunchKernelGGL(tail_ff2_kernel, dim3((tiles + 3) / 4), dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}
