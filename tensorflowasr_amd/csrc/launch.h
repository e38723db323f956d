// Host-visible launch interface between the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- which arithmetic a launch used (reported to callers through mi355asr_profile_schemes) ----
// A launch_* function that picks between kernel variants records the operand scheme of the one it launched; the profiling
// scope around the launch (model.h ProfScope) files it under the kernel category.  bench.py prices each kernel against the
// peak of the pipe the LIBRARY says it ran on instead of mirroring the environment switches.
enum OperandScheme {
  SCHEME_F32 = 0,      // v_mfma_f32_16x16x4_f32 / fp32 VALU: exact fp32 products
  SCHEME_BF16X3 = 1,   // fp32 operands as three bf16 terms, six v_mfma_f32_16x16x16/32_bf16 per fragment pair
  SCHEME_F16X2 = 2,    // fp32 operands as two fp16 terms, three v_mfma_f32_16x16x32_f16 per fragment pair
  SCHEME_BF16 = 3,     // operands rounded to bf16 (gemm_dtype = bfloat16): one MFMA per fragment pair
};
extern thread_local int mi355asr_last_scheme;   // api.hip
static inline void note_scheme(int s) { mi355asr_last_scheme = s; }

enum { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_QKV = 2, EPI_GLU = 3, EPI_HEAD = 4 };

// column-chunk width (in 16-wide tiles) a wave accumulates at once, per (dmodel, epilogue).
// Packed weights are padded to a multiple of this (for GLU: each half).
static inline int gemm_ct(int D, int epi) {
  if (epi == EPI_HEAD) return 12;
  if (epi == EPI_GLU) return D == 144 ? 3 : 4;
  return D == 144 ? 9 : 8;
}

struct Chain2Args {
  const float* x;      // [M, D] GEMM1 input rows
  const float* res;    // [M, D] residual rows
  float* y;            // [M, D]
  const float* ln_g;   // prologue LayerNorm (MODE 0)
  const float* ln_b;
  const float* w1p;    // packed [D/16][HT][64][4]
  const float* b1;     // [HT*16]
  const float* aff_s;  // hidden affine (MODE 1: folded BatchNorm scale / shift)
  const float* aff_t;
  const float* w2p;    // packed [HT][D/16][64][4]
  const float* b2;     // [D]
  const float* fln_g;  // optional trailing LayerNorm (block-final LN), nullptr = none
  const float* fln_b;
  float scale;
  float eps;
  int M;
};

struct GemmArgs {
  const float* x;    // [M, D]
  float* y;          // [M, ldy] (may be nullptr for EPI_HEAD)
  const float* res;  // [M, ldy] (EPI_RESIDUAL)
  const float* ln_g;
  const float* ln_b;
  const float* wp;   // packed [D/16][NT][64][4]
  const float* bias; // [NT*16]
  int M, NT, ldy, n_valid;
  float eps, qscale;
  int qtiles;
  int32_t* argmax_out;  // [M] (EPI_HEAD)
  float* maxval_out;    // [M] or nullptr
};

struct AttnArgs {
  // q [B*Tq, ldq], k / v [B*Tk, ldk]: head h occupies columns [h*HS, (h+1)*HS) of each.  Self-attention passes the
  // three column blocks of one [B*T, 3D] qkv buffer (Tq == Tk); the Translator's cross-attention passes q from the
  // token stream and k / v from the encoder output (conformer_blocks.py:459).
  const float *q, *k, *v;
  float* ctx;        // [B*Tq, D]
  int B, Tq, Tk, H, D, ldq, ldk;
  int q_off = 0;     // band mask only: query row tq is sequence position q_off + tq of the Tk keys (streaming: the
                     // queries are the last Tq rows of [cache ; new], chunk_conformer_blocks.py:209-214)
  // band attention of the ChunkConformer (chunk_conformer_blocks.py:158-176): query i sees keys
  // [min(max(i-win_front,0), T-win_back), max(min(i+win_back,T), win_back)]; win_front < 0 = full attention
  // (band attention is self-attention only: Tq == Tk == T)
  int win_front, win_back;
  // two-term fp16 scheme of attention_split_kernel: powers of two with bound(|q| log2 e) * h2_sq, bound(|k|) * h2_sk,
  // bound(|v|) * h2_sv <= 2^15 (api.hip derives the bounds from the q / k / v weights and the LayerNorm in front of them);
  // 0 = unknown bounds (stage calls on caller-supplied q / k / v): the three-term bf16 kernel
  float h2_sq = 0.f, h2_sk = 0.f, h2_sv = 0.f;
  // round 5, attention_split_kernel only: q / k / v stored HEAD-MAJOR by the producer (pp_ff1_consume): three planes of
  // [B, H, T, 36] floats, so that the 36 floats of a (token, head) are one contiguous 144-byte row and a head's T rows one
  // contiguous block -- the token-major [B T, 3 D] rows put a head's 144 bytes at a 1728-byte stride, two 128-byte lines per
  // row, 1.79x the fetch the kernel needs (round-4 counters).  q / k / v then point at the planes, ldq / ldk are 36.
  int head_major = 0;
};

struct DwArgs {
  const float* u;   // [B, T, D]
  float* y;         // [B, T, D]
  const float* wd;  // [K, D]
  int B, T, D, pad_left;
};

// frontend / subsampling (frontend.hip)
struct StftArgs {
  const float* wav;   // [B, L]
  float* logp;        // [B, F, LP] 10*log10(max(power,1e-10)) (or log10 for 'valid')
  float* pmax;        // [B, FT * NCH] per-(frame tile, column chunk) maxima
  const float* wp;    // packed DFT kernels [n_dft/16][NT][64][4], columns interleaved re/im per bin
  int B, L, F, hop, pad_left, n_dft, NT, LP, nbins, FT, NCH;
  int db10;           // 1: 10*ln(p)/ln10 ; 0: ln(p)/ln10
};
// 32 x 32 Cooley-Tukey STFT (fft_stft.hip); n_dft = 1024 only
struct FftStftArgs {
  const float* wav;     // [B, L]
  float* logp;          // [B, F, LP]
  float* pmax;          // [B, F] per-frame maxima
  const float* w1p;     // stage-1 DFT-32 [2][4][64][4]
  const float* w2p;     // stage-2 DFT-32 on (re | im) [4][2][64][4]
  const float* tw_c;    // cos(2 pi k1 n2 / 1024) [32 k1][32 n2]
  const float* tw_s;    // sin(...)
  const float* window;  // [1024]
  int B, L, F, hop, pad_left, LP;
  int db10;
  // the two DFT-32 stage matrices as split-bf16 fragments (pack_split32: [steps][tiles][3 terms][64 lanes][8]) for the bf16
  // matrix pipe (fft_stft_split_kernel, round 3), or null
  const float* w1s = nullptr;   // stage 1: K = 32 (n1), 64 columns -> [1][4][3][64][8]
  const float* w2s = nullptr;   // stage 2: K = 64 (re | im of n2), 32 columns -> [2][2][3][64][8]
  const float *w1h = nullptr, *w2h = nullptr;   // the same two matrices times 2^14 as hi + lo fp16 terms ([..][2][64][8]; two-term scheme)
};
int launch_fft_stft(const FftStftArgs& a, hipStream_t s);
struct UttMaxArgs { const float* pmax; float* umax; int n; };
struct MelArgs {
  const float* logp;  // [B, F, LP]
  const float* umax;  // [B] or nullptr (no max-normalisation / floor: 'valid' chunk frontend)
  float* mel;         // [B, F, NM]
  const float* wp;    // packed freq2mel [KBm][NTm][64][4]
  int B, F, LP, nbins, KBm, NTm, NM, FT;
  float floor_db;
  // banded form of freq2mel (mel_band_kernel), or null: band[m] = (first bin, number of bins) of mel filter m, bw[m][j] =
  // its weight at bin first + j, rows of BW floats
  const int* band = nullptr;
  const float* bw = nullptr;
  int BW = 0;
  // mel_band_kernel only: when set ([B] words), the largest |mel| of every utterance is left here as a float bit pattern (atomicMax
  // on the bits of non-negative floats; the caller zeroes the words first).  The 'valid' chunk frontend has no dB normalisation, hence no static
  // bound on its features: the two-term subsampling conv takes its operand scale from this run-time maximum.
  unsigned* absmax = nullptr;
};
struct SubConvArgs {
  const float* mel;   // [B, F, NM]
  float* out;         // [B*T2*F2, D]
  const float* w1;    // conv1 kernel [3][3][D]
  const float* b1;    // [D]
  const float* w2p;   // packed conv2 kernel, K order = (cblock, kt, kf, 16) -> [9*D/16][D/16][64][4]
  const float* b2;    // [D]
  const float* w2s;   // conv2 kernel as split-bf16 fragments [column chunk][4 d/16 + ceil(d/32) steps][9 or 8 tiles][3 terms][64 lanes][8] (subconv.hip), or null
  // the same kernel as TWO fp16 terms (round-to-nearest hi + lo of the kernel times h_wscale, a power of two), or null: the
  // conv1 values are then scaled by h_scale (a power of two with bound(conv1) * h_scale <= 2^15), three products per
  // fragment pair instead of six, the accumulators carry h_scale * h_wscale (subconv.hip, "two-term scheme")
  const float* w2h = nullptr;
  float h_scale = 1.f, h_wscale = 1.f;
  // run-time operand scale (features without a static bound): when h_melmax is set ([B] words, one per utterance: results do not
  // depend on what else is in the batch), h_scale is derived in the kernel as the power
  // of two that puts h_l1 * max|mel of the utterance| + h_bmax (h_l1 = the largest L1 norm of a conv1 filter, h_bmax = the largest |bias|) at 2^15
  const unsigned* h_melmax = nullptr;
  float h_l1 = 0.f, h_bmax = 0.f;
  // conv1 on the matrix pipe (subconv.hip, C1M): the mel patch is staged as fp16 hi + lo of mel * c1_mscale (a power of two with
  // bound(mel) * c1_mscale <= 2^15; 0 with h_melmax set: derived per utterance from the run-time maximum), the conv1 kernel as
  // hi + lo of w * c1_wscale; both 0: conv1 is evaluated on the VALU in fp32
  float c1_mscale = 0.f, c1_wscale = 0.f;
  int B, F, NM, T1, F1, T2, F2;
  int st1;            // conv1 time stride (reduction_factor/2)
  int pt1, pf1, pt2, pf2;  // pad-before of conv1 (time,freq) and conv2 (time,freq)
};
struct StreamGemmArgs {
  const float* x;     // [M, K]
  float* y;           // [M, ldy]
  const float* wp;    // packed [K/16][NT][64][4]
  const float* bias;
  int M, K, NT, ldy, n_valid;
};
struct CollapseArgs {
  const int32_t* frame_ids;  // [B, T]
  const int32_t* in_len;     // [B] or nullptr (= T)
  int32_t* ids;              // [B, T] padded with -1
  int32_t* out_len;          // [B]
  int B, T, blank;
};

struct PickArgs {
  const int32_t* frame_ids;  // [B, T] phone argmax
  int32_t* idx;              // [B, T] indices of kept frames (first cnt[b] entries valid); may alias frame_ids (in place)
  int32_t* cnt;              // [B]
  int B, T, blank;
};
struct GatherArgs {
  const float* src;          // [B, T, D]
  const int32_t* idx;        // [B, T]
  const int32_t* cnt;        // [B]
  float* dst;                // [B, Tp, D] zero padded
  int B, T, Tp, D;
};
struct EmbedArgs {
  const int32_t* ids;        // [M]
  const float* table;        // [V, D]
  float* dst;                // [M, D]
  int M, V, D;
};
struct AddPeArgs {
  const float* src;          // [B, U, D]
  const float* pe;           // [>= U, D]
  float* dst;                // [B, U, D]
  int B, U, D;
};
int launch_embed(const EmbedArgs& a, hipStream_t s);
int launch_add_pe(const AddPeArgs& a, hipStream_t s);
// LEAF frontend (leaf.hip)
struct LeafConvArgs {
  const float* wav;     // [B, L]
  const float* wp;      // Gabor filters, P16 packed [26][10][64][4]: rows = taps (401, zero padded), cols = (re, im) per filter
  const float* gcoef;   // [80] -0.5 log2(e) / (sigma_c * 200)^2 of the Gaussian pooling windows
  float* part;          // [B, NH, 4, 80] partial pooled sums per tile of 128 positions (NH = ceil(L / 128))
  float p0, p1;         // pre-emphasis taps: xp[n] = p0 x[n] + p1 x[n+1]
  int B, L, F, NH, hop, pl;   // pl = left padding of the SAME pooling
};
struct LeafPcenArgs {
  const float* part;
  const float *alpha, *delta, *root, *smooth, *gamma, *beta;   // [80] each
  float* out;           // [B, F, 80]
  int B, F, NH, hop, pl;
  int tile, nrel;       // positions per partial tile and frame slots per tile: 128 / 4 (fp32 kernel), 512 / 7 (split kernel)
};
int launch_leaf_conv_pool(const LeafConvArgs& a, hipStream_t s);
// split-bf16 variant (ns = 2 or 3 bf16 terms per fp32 operand): wp = bf16 fragments [13][10][ns][64][8]
constexpr int kLeafSplitTile = 512, kLeafSplitSlots = 7;   // positions per workgroup, frames a tile can touch
int launch_leaf_conv_pool_split(int ns, const LeafConvArgs& a, hipStream_t s);
int launch_leaf_pcen_norm(const LeafPcenArgs& a, hipStream_t s);
// add_wav_info branch (wavpick.hip)
struct WpSepConvArgs {
  const float* wav;     // [B, L]
  const float *dw, *pw, *bias;   // depthwise [7], pointwise [32], bias [32]
  float* out;           // [B, T0, 32] = LeakyReLU(SeparableConv1D(32, 7, stride))
  int B, L, T0, stride, pad_left;
  float slope;
};
struct WpPadActArgs {
  const float* src;     // [B, T, C]
  const float* src2;    // optional second addend [B, T, C]
  float* dst;           // [B, T + lo + hi, C]
  int B, T, C, lo, hi;
  int reflect;          // 1: tf.pad REFLECT, 0: zeros
  float slope;          // LeakyReLU slope applied to (src + src2); 1 = identity
};
int launch_wp_sepconv(const WpSepConvArgs& a, hipStream_t s);
int launch_wp_pad_act(const WpPadActArgs& a, hipStream_t s);
// bf16-MFMA GEMM family (bf16.hip)
enum { E16_BIAS = 0, E16_SWISH = 1, E16_RES = 2, E16_QKV = 3, E16_GLU = 4, E16_AFFSWISH = 5, E16_HEAD = 6 };
struct Gemm16Args {
  const float* x;       // [M, ldx] fp32, the first K columns are the operand
  int ldx;
  const void* wp;       // bf16 weights in P16 fragment order [K/16][NT][64 lanes][4]
  const float* bias;    // [NT*16]
  float* y;             // [M, ldy] (E16_HEAD: may be null)
  int ldy;
  const float* res;     // [M, ldy] (E16_RES)
  const float *ln_g, *ln_b;     // prologue LayerNorm over K
  const float *fln_g, *fln_b;   // E16_RES: optional LayerNorm over the output row (N = 16*NT)
  const float *aff_s, *aff_t;   // E16_AFFSWISH
  int M, K, NT, n_valid;
  float scale, eps, qscale;
  int qtiles;
  int32_t* argmax_out;  // [M] (E16_HEAD)
  // optional batched / overlapping rows (conv1d as a GEMM on channels-last activations): row r reads from
  // x + (r / rpb) * bstride + (r % rpb) * ldx when rpb > 0
  int rpb = 0;
  long long bstride = 0;
  // E16_HEAD on the ring kernel (round 6): with scratch for [ranges][M] (maximum, class) pairs the class chunks of a row tile are
  // split over up to part_max workgroups (rows alone: 16 640 / 128 = 130 workgroups on 256 CUs), launch_head_combine picks the winner
  float* part_v = nullptr;
  int32_t* part_i = nullptr;
  int part_max = 0;
  int head_chunks = 0;  // (set by the launcher: all chunks of the head, the last range may hold fewer than cpw)
};
int launch_head_combine(const float* part_v, const int32_t* part_i, int ranges, int M, int32_t* argmax_out, float* maxval_out, hipStream_t s);   // fused_pp.hip
int launch_gemm16_bf16(int epi, bool ln, const Gemm16Args& a, hipStream_t s);
int launch_gemm256_bf16(int epi, bool ln, const Gemm16Args& a, const void* ring, hipStream_t s);   // bf16.hip (round 6): K = 256, many rows, rows resident in LDS; -1: not taken
int launch_chain256_bf16(int mode, const Chain2Args& a, hipStream_t s);   // bf16.hip: dmodel 256, bf16 mode, FFModule / conv tail in one launch
int launch_gemm16_f32(int epi, bool ln, const Gemm16Args& a, hipStream_t s);   // wp = fp32 P16 weights
// stream256.hip (round 5): the whole block stack of the streaming encoder (bf16 mode, dmodel 256 = 4 heads x 64, chunks of <= 16
// rows) in ONE launch, one workgroup per chunk.  Matrices: the one-term slab-ring packs (api.hip: put_ring); vectors: f32
struct S256Block {
  const float *ff_ln_g[2], *ff_ln_b[2], *ff_b1[2], *ff_b2[2];
  const void *ff_w1[2], *ff_w2[2];
  const float *att_ln_g, *att_ln_b, *qkv_b, *out_b;
  const void *qkv_w, *out_w;
  const float *cv_ln_g, *cv_ln_b, *pw1_b, *dw_w, *pc_b1, *bn_s, *bn_t, *pw2_b;
  const void *pw1_w, *pc_w1, *pw2_w;
  const float *ln_g, *ln_b;
};
constexpr int S256_MAXB = 8;
struct S256Args {
  const float* x;      // [B * T, 256] block-stack input rows (chunk b = rows b T .. b T + T - 1)
  float* y;            // [B * T, 256]
  int B, T, nblocks, ksz, pad_left;
  float fc, qscale, eps;
  S256Block blk[S256_MAXB];
};
// the ONE shape predicate of stream256_kernel, shared by the launcher and by api.hip's stream256_args (round-5 advice: two copies of
// the same checks).  The kernel is written for dmodel 256, 4 heads of 64, a 4 d FFN hidden, a 2 d conv hidden, 'same' padding and
// kernel_size 5 (depthwise taps in registers: Streaming_ConformerS.yml); chunks of 1 .. 16 rows, one workgroup per chunk
inline bool stream256_shape_ok(int B, int T, int nblocks, int ksz) {
  return B > 0 && T >= 1 && T <= 16 && nblocks >= 1 && nblocks <= S256_MAXB && ksz == 5;
}
int launch_stream256(const S256Args& a, hipStream_t s);   // -1: not this kernel's shape (the caller falls back to the per-layer path)
int launch_to_bf16(const float* src, void* dst, size_t n, hipStream_t s);
// gemm_ring.hip: the same contract on the bf16 pipe with exactly split fp32 operands, weights as a slab ring
// [N / 128 chunks][K / 32 steps][8 tiles][3 terms (1 in bf16 mode)][64 lanes][8 bf16] (api.hip: put_ring); -1: shape not taken
bool gemm_ring_applicable(int epi, bool ln, const Gemm16Args& a);
int launch_gemm_ring(int epi, bool ln, const Gemm16Args& a, const void* ring, int terms, hipStream_t s);   // terms: 3 (fp32) or 1 (bf16 mode)
// block-level fused kernels (fused.hip, dmodel 144)
// host-side constants of one pair-pipelined chain y += W2 act(W1aug [x ; 1]) in the two-term fp16 scheme (fused_pp.hip): the
// powers of two the streamed matrices were multiplied by, the largest column L1 norm of W1 and the largest |bias| (row 144)
struct PpChainSc { float sw1 = 1.f, sw2 = 1.f, l1 = 0.f, bmax = 0.f; };
struct Ff1QkvArgs {
  const float* x0; float* x1; float* qkv;
  const float *ff_ln_g, *ff_ln_b, *ff_w1p, *ff_b1, *ff_w2p, *ff_b2;
  const float *att_ln_g, *att_ln_b, *qkv_wp, *qkv_b;
  float fc, qscale, eps;
  int M;
  const float* slabs = nullptr;   // slab stream of ff1_qkv_ring_kernel (55 slabs of 1792 fragments), or null
  const float* pp_slabs = nullptr;   // pair-pipelined stream (fused_pp.hip: 51 ring slots of 20 fragments; biases in row 144), or null
  PpChainSc pp_sc;                   // ... the scales its ff_module_1 chain was packed with (api.hip: append_pp_chain)
  float pp_sw_qkv = 1.f;             // ... and its q / k / v matrix
  // round 5: qkv_T > 0 = store q / k / v head-major for attention_split_kernel (AttnArgs::head_major): planes of M * 144 floats,
  // row ((b H + h) T + t) of 36 floats; qkv_T = frames per utterance (M = B qkv_T), qkv_H heads of 36.  Pair-pipelined kernels only:
  // every other ff_module_1 kernel refuses such a launch.
  int qkv_T = 0, qkv_H = 0;
  // round 4: the plain layer in front of the block (the subsampling Dense, the CTC decoder's projection) in the same launch:
  // x0 = pre_x [M, 144 * pre_chunks] W + b from its two-term stream (api.hip: append_pp_plain) -- x0 above is not read then
  const float* pre_x = nullptr; const float* pre_pp = nullptr;
  float pre_sw = 1.f;
  int pre_chunks = 0;
  // round 6 (fused_ns.hip): the same two-term fragments in plain order for the N-split kernels -- ff_module_1's W1aug [5][36][2][64]
  // and W2 [18][9][2][64], q / k / v [5][27][2][64] (u32x4 per lane; api.hip: pack_half32); packed with pp_sc / pp_sw_qkv
  const float *ns_w1 = nullptr, *ns_w2 = nullptr, *ns_qkv = nullptr;
  // round 6: the Translator's RBlock (conformer_blocks.py:447-503): only the QUERY is projected from this stream -- of LayerNorm(x1 +
  // positional encoding) -- keys and values come from the encoder output through their own projection.  xq_pe = the encoding table
  // [>= xq_U, 144], row = the token's position in its sequence of xq_U tokens; q is stored token-major at columns 0..143 of the
  // [M, 432] qkv rows.  Pair-pipelined kernel only (the stream's q group is read, its k / v groups are not).
  const float* xq_pe = nullptr;
  int xq_U = 0;
};
struct OutGluArgs {
  const float* ctx; const float* x1; float* x2; float* u;
  const float *out_wp, *out_b, *cv_ln_g, *cv_ln_b, *pw1_wp, *pw1_b;
  float eps;
  int M;
  const float* og_slabs = nullptr;                     // the same kernels as the three-term slab stream of out_glu_ld_kernel (15 slabs of 1792 fragments)
  // ... and as the two-term fp16 stream of pp_out_glu_kernel (fused_pp.hip: 15 ring slots -- out projection, pw_conv_1 value
  // tiles, gate tiles; biases in row 144), packed with these powers of two
  const float* pp_slabs = nullptr;
  float pp_sw_out = 1.f, pp_sw_pw1 = 1.f;
  // round 6 (fused_ns.hip): the same fragments in plain order -- out projection [5][9][2][64], pw_conv_1 [5][18][2][64] (u32x4 per lane)
  const float *ns_out = nullptr, *ns_pw1 = nullptr;
  // ... and, for the one-tile-per-workgroup kernels, the attention in front of the out projection in the same launch (ctx is then
  // not read): q / k / v of the block (AttnArgs' layout fields), T <= 256 frames per utterance, the two-term operand scales
  int attn = 0;
  const float *aq = nullptr, *ak = nullptr, *av = nullptr;
  int a_T = 0, a_H = 0, a_ldq = 0, a_ldk = 0, a_head_major = 0;
  float a_sq = 0.f, a_sk = 0.f, a_sv = 0.f;
};
struct TailFf2Args {
  const float* dw; const float* x2; float* y;
  const float *pc_w1p, *pc_b1, *bn_s, *bn_t, *pw2_wp, *pw2_b;
  const float *ff_ln_g, *ff_ln_b, *ff_w1p, *ff_b1, *ff_w2p, *ff_b2, *ln_g, *ln_b;
  float fc, eps;
  int M;
  const float* slabs = nullptr;   // slab stream of tail_ff2_ring_kernel (60 slabs of 1792 fragments), or null
  const float* pp_slabs = nullptr;   // pair-pipelined stream (fused_pp.hip: 54 ring slots; BatchNorm and biases folded into W1), or null
  PpChainSc pp_sc[2];                // ... the scales of its conv-tail chain and of its ff_module_2 chain
  // depthwise conv folded into the pair-pipelined kernel's prologue (fused_pp.hip): when dw_u is set, `dw` is not read -- the
  // kernel tiles the tokens per utterance (64-token chunks of the dw_T frames of each of M / dw_T utterances) and computes
  // dw = depthwise_k32(u) (taps dw_wd [32, 144], dw_pad zeros in front) from a 95-row window in LDS
  const float* dw_u = nullptr;
  const float* dw_wd = nullptr;
  int dw_T = 0, dw_pad = 0;
  // round 4: the class head behind the block (the CTC decoder's last block; out-projection + GLU kernels without a next block):
  // logits = y W + b over head_groups column groups of nine tiles from the two-term stream head_pp (pp_head_kernel's loop),
  // per-frame arg-max / maximum and / or the logits; y itself is stored only if `y` is set
  const float* head_pp = nullptr;
  float head_sw = 1.f;
  int head_groups = 0, head_ldy = 0, head_nvalid = 0;
  float* head_y = nullptr;
  int32_t* head_argmax = nullptr;
  float* head_maxval = nullptr;
  // round 6 (fused_ns.hip): the chains' fragments in plain order -- conv tail W1aug [5][18][2][64] / W2 [9][9][2][64], ff_module_2
  // W1aug [5][36][2][64] / W2 [18][9][2][64]; packed with pp_sc[0] / pp_sc[1]
  const float *ns_cv_w1 = nullptr, *ns_cv_w2 = nullptr, *ns_ff_w1 = nullptr, *ns_ff_w2 = nullptr;
};
int launch_ff1_qkv(const Ff1QkvArgs& a, hipStream_t s);
int launch_out_glu(const OutGluArgs& a, hipStream_t s);
int launch_tail_ff2(const TailFf2Args& a, hipStream_t s);
bool tail_ff1_available();
bool tail_pp_selected();   // launch_tail_ff1 / launch_tail_ff2 will take the pair-pipelined kernels when the block has their streams
int launch_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, hipStream_t s);   // -1: not available, nothing launched
// pair-pipelined versions (fused_pp.hip); -1: switched off (MI355ASR_PP=0) or no pp_slabs, nothing launched
bool pp_enabled();
int launch_pp_out_glu(const OutGluArgs& a, hipStream_t s);
// class head of dmodel 144 on the two-term fp16 stream (pp: append_pp_plain of [W ; b] over `groups` column groups, packed with pp_sw)
int launch_pp_head(const GemmArgs& a, const float* pp, float pp_sw, int groups, hipStream_t s);
// ... with the column groups split over `ranges` workgroups per row tile (round 6: few rows, many classes); scratch: 2 * ranges * M words
int pp_head_ranges(int M, int groups);
int launch_pp_head_split(const GemmArgs& a, const float* pp, float pp_sw, int groups, int ranges, float* scratch, hipStream_t s);
bool pp_dw_fold_ok(int T, int ksz);   // the tail kernels can take the depthwise conv (kernel size ksz, T frames per utterance) in their prologue
int launch_pp_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, hipStream_t s);
int launch_pp_tail_ff2(const TailFf2Args& a, hipStream_t s);
int launch_pp_ff1_qkv(const Ff1QkvArgs& b, hipStream_t s);
// N-split kernels (fused_ns.hip, round 6): one 16-token tile per workgroup, for small batches (up to MI355ASR_NS1_MAX_M rows): ff_module_1 + qkv; and what the folded tail
// launches do, as out-projection + GLU (writes g.x2, g.u) followed by depthwise conv + tail [+ next ff_module_1 + qkv when b is set]
bool ns1_rows_ok(int M);
bool ns1_block_ok(const TailFf2Args& a, const Ff1QkvArgs* b, const OutGluArgs& g);      // launch_ns1_og_tail will take this block
bool ns1_attn_ok(int hs, const AttnArgs& at);                                             // ... with its attention in the first launch
int launch_ns1_ff1_qkv(const Ff1QkvArgs& b, hipStream_t s);
int launch_ns1_og_tail(const TailFf2Args& a, const Ff1QkvArgs* b, const OutGluArgs& g, hipStream_t s);
int launch_ns1_head(const GemmArgs& a, const float* ns, float sw, int groups, hipStream_t s);
int launch_ns1_sublinear(const StreamGemmArgs& a, const float* ns, float sw, hipStream_t s);   // ns: [W chunk ; b] fragments in plain order [5][9 chunks][2][64]   // ns: [W ; b] fragments in plain order [5][9 groups][2][64]
bool ff1_qkv_pp_selected(bool has_slabs, bool has_pp);   // fused.hip: q, k, v will come from the pair-pipelined producer (head-major layout possible)
bool ff1_pre_selected();             // ... and launch_ff1_qkv will take that kernel (fused.hip) when the block has its streams
bool pp_pre_fold_ok();
bool pp_head_fold_ok(int M, int n_valid, int groups);   // the class head rides in the last block's tail launch (MI355ASR_PP_HEADF=0: own launch)               // the layer in front of a block rides in its ff_module_1 + qkv launch (MI355ASR_PP_PRE=0: own launch)
// round 4: out-projection + residual + LayerNorm + pw_conv_1 + GLU in the prologue of the pair-pipelined tail kernels (the
// block = attention + ONE launch); -1: not applicable (switched off, no streams, depthwise fold impossible), nothing launched
bool pp_og_fold_ok(const TailFf2Args& a, const OutGluArgs& g);
int launch_pp_og_tail_ff1(const TailFf2Args& a, const Ff1QkvArgs& b, const OutGluArgs& g, hipStream_t s);
int launch_pp_og_tail_ff2(const TailFf2Args& a, const OutGluArgs& g, hipStream_t s);
int launch_sublinear_split(const StreamGemmArgs& a, const float* ws, hipStream_t s);   // -1: shape not supported
// the same layer on the two-term fp16 stream (fused_pp.hip; pp = append_pp_plain of K / 144 chunks [W_f ; bias or 0], packed with pp_sw)
int launch_pp_sublinear(const StreamGemmArgs& a, const float* pp, float pp_sw, hipStream_t s);
bool pp_sublinear_ok(const StreamGemmArgs& a, const float* pp);   // ... would take it
int launch_pick(const PickArgs& a, hipStream_t s);
int launch_row_argmax(const float* x, int32_t* out, int M, int V, hipStream_t s);
int launch_gather(const GatherArgs& a, hipStream_t s);
int launch_chain2(int D, int mode, const Chain2Args& a, hipStream_t s);
int launch_gemm_rows(int D, int epi, bool ln, const GemmArgs& a, hipStream_t s);
int launch_attention(int HS, const AttnArgs& a, hipStream_t s);
bool attention_head_size_ok(int HS);      // 36 / 64 (tuned kernels) and 12, 16, 24, 32, 48, 72, 128 (online-softmax kernel only)
bool attention_lds_applicable(int HS, const AttnArgs& a);
int launch_attention_lds(int HS, const AttnArgs& a, hipStream_t s);
bool attention_split_applicable(int HS, const AttnArgs& a);
bool attention_split_two_term(int HS, const AttnArgs& a);
bool attention_takes_head_major(int HS, const AttnArgs& a);   // blocks.hip: launch_attention's own switches included   // the two-term kernel would take this launch (head-major operands allowed)
int launch_attention_split(int HS, const AttnArgs& a, hipStream_t s);
bool attention_split64_applicable(int HS, const AttnArgs& a);   // attention_split64.hip: head size 64, two fp16 terms, <= 288 keys
int launch_attention_split64(int HS, const AttnArgs& a, hipStream_t s);
int launch_dwconv(int K, const DwArgs& a, hipStream_t s);
int launch_stft(const StftArgs& a, hipStream_t s);
int launch_utt_max(const UttMaxArgs& a, int B, hipStream_t s);
int launch_head_ld(const GemmArgs& a, const float* slabs, int groups, hipStream_t s);   // dmodel-144 class head on the slab ring; -1: not taken
int launch_mel(const MelArgs& a, hipStream_t s);
int launch_mel_band(const MelArgs& a, hipStream_t s);   // freq2mel as the banded matrix it is (triangular filters): HBM-bound
int launch_db_norm(const MelArgs& a, hipStream_t s);   // mel_layer_type 'Spectrogram': the dB normalisation without the mel matrix
int launch_subconv(int D, const SubConvArgs& a, hipStream_t s);
int launch_subconv144(const SubConvArgs& a, hipStream_t s);
int launch_subconv_split(int d, const SubConvArgs& a, hipStream_t s);   // split-bf16 ring kernel (dmodel 144 / 256 / 512); -1: not supported, nothing launched
int launch_stream_gemm(int D, const StreamGemmArgs& a, hipStream_t s);
int launch_collapse(const CollapseArgs& a, hipStream_t s);
