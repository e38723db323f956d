// Internal host-side model definitions shared by api.hip, api_chunk.hip and api_translator.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mi355asr.h"
#include "beam.h"
#include "launch.h"
#include "env.h"

namespace mi355 {

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) return fail(MI355ASR_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

#define LAUNCH_TRY(expr, what)                                                             \
  do {                                                                                     \
    if ((expr) != 0) return fail(MI355ASR_EINVAL, "no kernel instantiation for %s", what); \
    hipError_t e__ = hipGetLastError();                                                    \
    if (e__ != hipSuccess) return fail(MI355ASR_EHIP, "launch %s: %s", what, hipGetErrorString(e__)); \
  } while (0)

constexpr float kLnEps = 1e-3f;  // Keras LayerNormalization default
constexpr float kBnEps = 1e-3f;  // Keras BatchNormalization default

struct HostTensor {
  std::vector<float> data;
  bool set = false;
};
struct Expected {
  std::string name;
  std::vector<int64_t> dims;  // Keras layout
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : dims) n *= d;
    return n;
  }
};

struct BlockDev {
  // ff_module_1 / ff_module_2
  const float *ff_ln_g[2], *ff_ln_b[2], *ff_w1p[2], *ff_b1[2], *ff_w2p[2], *ff_b2[2];
  // mhsa_module
  const float *att_ln_g, *att_ln_b, *qkv_wp, *qkv_b, *out_wp, *out_b;
  // RBlock cross-attention (Translator): query kernel [d,d] and [key | value] kernels [d,2d], packed separately
  const float *xq_wp = nullptr, *xkv_wp = nullptr;
  // conv_module
  const float *cv_ln_g, *cv_ln_b, *pw1_wp, *pw1_b, *dw_w, *pc_w1p, *pc_b1, *bn_s, *bn_t, *pw2_wp, *pw2_b;
  // block-final LayerNorm
  const float *ln_g, *ln_b;
  // out-projection / pw_conv_1 kernels as split-bf16 fragments (dmodel 144, Keras-layout MHA; fused.hip), or null
  const float *og_slabs = nullptr, *ff1_slabs = nullptr, *tail_slabs = nullptr;
  // the pair-pipelined streams of fused_pp.hip (ff_module_1 + qkv ; conv tail + ff_module_2), or null
  const float *pp_ff1 = nullptr, *pp_tail = nullptr, *pp_og = nullptr;
  float att_h2[3] = {0.f, 0.f, 0.f};       // operand scales of the two-term attention kernel (0: no bound)
  float pp_sw_out = 1.f, pp_sw_pw1 = 1.f;
  PpChainSc pp_ff1_sc, pp_tail_sc[2];      // the scales those streams were packed with (two-term fp16 scheme)
  float pp_sw_qkv = 1.f;
  // the same two-term fragments in plain [step][tile][term] order for the N-split kernels of fused_ns.hip (round 6), or null:
  // ff_module_1 (W1aug, W2), q / k / v, out projection, pw_conv_1, conv tail (W1aug, W2), ff_module_2 (W1aug, W2)
  const float *ns_ff1_w1 = nullptr, *ns_ff1_w2 = nullptr, *ns_qkv = nullptr, *ns_out = nullptr, *ns_pw1 = nullptr,
              *ns_cv_w1 = nullptr, *ns_cv_w2 = nullptr, *ns_ff2_w1 = nullptr, *ns_ff2_w2 = nullptr;
};

struct Dims {
  int hop, nbins, NT_dft, NCH_dft, LP, KBm, NTm, F1, F2, st1, pf1, pf2;
};

// per-stack block options: ConformerBlock (full attention, 'same' depthwise padding) or ChunkConformerBlock
// (band attention [win_front, win_back], 'causal' depthwise padding; chunk_conformer_blocks.py:327-398)
struct BlockOpts {
  int ksz = 32;
  float fc = 0.5f;
  int win_front = -1;   // < 0: full attention
  int win_back = 0;
  bool causal = false;
  // round 4: the plain layer in front of the block riding in the block's first launch (block_takes_pre says when):
  // x0 = pre_x [B * T, 144 * pre_chunks] W + b from the two-term stream pre_pp; the block's input buffer (sc.xa) is not read
  const float* pre_x = nullptr; const float* pre_pp = nullptr;
  float pre_sw = 1.f;
  int pre_chunks = 0;
  // ... and the class head BEHIND the block (ctc_impl's last block), when the block's tail launch can take it: the head's
  // arguments (x is not read), its two-term stream; *head_done is set when the launch computed it
  const GemmArgs* head = nullptr;
  const float* head_pp = nullptr;
  float head_sw = 1.f;
  int head_groups = 0;
  bool* head_done = nullptr;
};

struct StackDev {
  std::vector<BlockDev> blocks;
  const float *proj_wp = nullptr, *proj_b = nullptr, *fc_wp = nullptr, *fc_b = nullptr;
  int NT_fc = 0, num_classes = 0;
  const float* proj_pp = nullptr;      // the projection as a two-term stream: rides in the first block's ff_module_1 + qkv launch
  float proj_pp_sw = 1.f;
  BlockOpts opts;
};

}  // namespace mi355
using namespace mi355;

struct mi355asr_model {
  mi355asr_config cfg;
  Dims dm;
  std::vector<Expected> expected;
  std::map<std::string, HostTensor> host;
  bool finalized = false;
  float* arena = nullptr;
  size_t arena_floats = 0;
  // gemm_dtype 1: the whole arena again in bf16 (same element offsets; packed matrices keep their fragment order)
  unsigned short* arena16 = nullptr;
  const void* w16(const float* p) const { return arena16 + (p - arena); }
  // fp32 P16 pack -> split-bf16 slab ring of the same matrix (gemm_ring.hip; dmodel 256 / 512 dense layers)
  std::unordered_map<const float*, const float*> ring_of;
  // mi355asr_set_expected_rows: the most rows (batch x frames) a call will bring, -1 = unknown.  Below the ring
  // kernels' crossover the rings are not packed at all (they cost 1.5 x the dense weights' bytes and their packing time).
  long expected_rows = -1;
  // dmodel 144: class-head P16 pack -> (slab stream of head_ld_kernel, column groups)
  struct HeadStreams { const float* slabs; int groups; const float* pp; float pp_sw; const float* ns; };   // pp: two-term fp16 stream (fused_pp.hip); ns: the same fragments in plain order (fused_ns.hip)
  std::unordered_map<const float*, HeadStreams> head_of;
  const float *dft_wp = nullptr, *mel_wp = nullptr, *c1_w = nullptr, *c1_b = nullptr, *c2_wp = nullptr,
              *c2_b = nullptr, *lin_wp = nullptr, *lin_b = nullptr, *proj_wp = nullptr, *proj_b = nullptr,
              *fc_wp = nullptr, *fc_b = nullptr;
  int NT_fc = 0;
  // freq2mel as a banded matrix (mel_band_kernel) when every filter's support is narrow, else null (pack_mel_band)
  const int* mel_band = nullptr;
  const float* mel_bw = nullptr;
  int mel_BW = 0;
  // FFT-as-GEMM STFT operands; fft_ok only when the loaded DFT kernels are window * exp(-2 pi i k n / N) (pack_fft)
  bool fft_ok = false;
  const float *fft_w1p = nullptr, *fft_w2p = nullptr, *fft_twc = nullptr, *fft_tws = nullptr, *fft_win = nullptr;
  const float *fft_w1s = nullptr, *fft_w2s = nullptr;   // the stage matrices as split-bf16 fragments (fft_stft_split_kernel)
  const float *fft_w1h = nullptr, *fft_w2h = nullptr;   // ... and times 2^14 as fp16 pairs
  // LEAF frontend (mel_layer_type 1): packed Gabor filters, pooling coefficients, PCEN / instance-norm vectors
  const float *leaf_wp = nullptr, *leaf_gcoef = nullptr, *leaf_alpha = nullptr, *leaf_delta = nullptr, *leaf_root = nullptr,
              *leaf_smooth = nullptr, *leaf_gamma = nullptr, *leaf_beta = nullptr;
  float leaf_p0 = 0.f, leaf_p1 = 1.f;
  const float* lin_wsplit = nullptr;    // subsampling Dense kernel as split-bf16 fragments, 1792 per 32-wide step (fused.hip)
  const float* lin_pp = nullptr;        // ... and as the two-term fp16 stream of pp_sublinear_kernel (F2 chunks of five ring slots), packed with
  float lin_pp_sw = 1.f;                // ... this power of two
  const float* lin_ns = nullptr;        // ... and the same fragments in plain order (fused_ns.hip: small batches)
  const float* proj_pp = nullptr;       // the CTC decoder's projection [W ; b] as such a stream (one chunk), packed with
  float proj_pp_sw = 1.f;               // ... this power of two
  const float* c2_wsplit = nullptr;     // conv2 kernel as split-bf16 fragments (subconv.hip; dmodel 144 / 256 / 512)
  const float* c2_whalf = nullptr;      // ... as two fp16 terms of kernel * c2_wscale (two-term scheme), conv1 values times c2_hscale
  float c2_hscale = 0.f, c2_wscale = 0.f;
  float c1_mscale = 0.f, c1_wscale = 0.f;  // conv1 on the matrix pipe (subconv.hip, C1M): power-of-two scales of the mel planes (static bound) and of the conv1 kernel
  float c1_l1 = 0.f, c1_bmax = 0.f;     // largest L1 norm of a conv1 filter / largest |bias|: the run-time operand bound of the chunk front
  const float* leaf_wsplit = nullptr;   // Gabor filters as split-bf16 MFMA fragments (leaf.hip)
  int leaf_terms = 3;                   // bf16 terms per fp32 operand in the Gabor conv (0: fp32 MFMA kernel)
  // add_wav_info: WavePickModel weights (conv kernels P16-packed with K = k * Cin)
  struct WavStage { const float *cw, *cb, *w5, *b5, *w1, *b1, *ws, *bs; int cin, c, stride; };
  const float *wp_dw = nullptr, *wp_pw = nullptr, *wp_b = nullptr, *wp_fw = nullptr, *wp_fb = nullptr;
  int wp_stride0 = 0;
  std::vector<WavStage> wp_stages;
  std::vector<BlockDev> enc_blocks, ctc_blocks;
  // ChunkConformer (mi355asr_chunk_create): front + encoder / phone picker / context helper / text decoder stacks
  bool is_chunk = false;
  mi355asr_chunk_config ccfg;
  // Translator (mi355asr_translator_create): Embedding + RBlock stack + Dense head
  bool is_translator = false;
  mi355asr_translator_config tcfg;
  StackDev t_stack;
  const float *t_emb = nullptr, *t_pe = nullptr;   // [inp_classes, d], [kMaxTokens, d]
  StackDev c_enc, c_picker, c_helper, c_decoder;
  // optional per-kernel timing with HIP events on the launch stream (mi355asr_profile_*)
  mutable bool prof = false;
  mutable std::vector<hipEvent_t> ev_free;
  struct Pending { int cat; hipEvent_t e0, e1; };
  mutable std::vector<Pending> ev_pending;
  mutable double prof_ms[MI355ASR_NUM_KERNELS] = {0};
  mutable int64_t prof_cnt[MI355ASR_NUM_KERNELS] = {0};
  mutable int prof_scheme[MI355ASR_NUM_KERNELS] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // OperandScheme of the last launch
  static_assert(MI355ASR_NUM_KERNELS == 20, "one initialiser per kernel category");
  hipEvent_t get_event() const {
    if (!ev_free.empty()) { hipEvent_t e = ev_free.back(); ev_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
};
namespace mi355 {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// brackets one kernel launch with events on its stream when profiling is on
struct ProfScope {
  const mi355asr_model* m;
  int cat;
  hipStream_t s;
  hipEvent_t e0 = nullptr;
  ProfScope(const mi355asr_model* m_, int cat_, hipStream_t s_) : m(m_), cat(cat_), s(s_) {
    mi355asr_last_scheme = SCHEME_F32;      // launchers with a choice of pipes overwrite it (common.h: note_scheme)
    if (m->prof) { e0 = m->get_event(); (void)hipEventRecord(e0, s); }
  }
  ~ProfScope() {
    m->prof_scheme[cat] = mi355asr_last_scheme;
    if (m->prof && e0) {
      hipEvent_t e1 = m->get_event();
      (void)hipEventRecord(e1, s);
      m->ev_pending.push_back({cat, e0, e1});
    }
  }
};
#define PROF(cat) ProfScope prof_scope_##cat(m, cat, s)

struct ArenaBuilder {
  std::vector<float> buf;
  // (offset of a P16 weight pack, offset of the same matrix as a gemm_ring.hip slab ring): resolved into
  // mi355asr_model::ring_of once the arena is on the device
  std::vector<std::pair<size_t, size_t>> ring_pairs;
  int ring_terms = 3;   // 3: fp32 weights as three bf16 terms; 1: bf16 mode (round-to-nearest-even bf16)
  // (offset of a dmodel-144 class head's P16 pack, offset of its slab stream for head_ld_kernel, column groups of nine tiles)
  struct HeadPair { size_t p16, slabs; int groups; size_t pp; float pp_sw; size_t ns; };
  std::vector<HeadPair> head_pairs;
  size_t put(const std::vector<float>& v) {
    size_t off = (buf.size() + 63) & ~(size_t)63;  // 256-byte alignment
    buf.resize(off + v.size());
    std::memcpy(buf.data() + off, v.data(), v.size() * sizeof(float));
    return off;
  }
  size_t put_padded(const float* p, size_t n, size_t padded) {
    std::vector<float> v(padded, 0.f);
    std::memcpy(v.data(), p, n * sizeof(float));
    return put(v);
  }
};

struct FftOff { bool ok = false; size_t w1 = 0, w2 = 0, twc = 0, tws = 0, win = 0, w1s = 0, w2s = 0, w1h = 0, w2h = 0; };
struct MelBandOff { bool ok = false; size_t band = 0, bw = 0; int BW = 0; };
// band form of freq2mel [nb, n_mels] for mel_band_kernel; ok = false when a filter spans more than 64 bins (a trained, dense matrix)
MelBandOff pack_mel_band(ArenaBuilder& ab, const std::vector<float>& f2m, int nb, int n_mels);
void use_mel_band(mi355asr_model* m, const MelBandOff& o, const float* base);
int launch_mel_auto(const mi355asr_model* m, MelArgs& me, hipStream_t s);   // banded kernel when available, else the GEMM

struct BlockOff {
  size_t ff_ln_g[2], ff_ln_b[2], ff_w1p[2], ff_b1[2], ff_w2p[2], ff_b2[2];
  size_t att_ln_g, att_ln_b, qkv_wp, qkv_b, out_wp, out_b;
  size_t xq_wp = 0, xkv_wp = 0;
  bool cross = false;
  size_t cv_ln_g, cv_ln_b, pw1_wp, pw1_b, dw_w, pc_w1p, pc_b1, bn_s, bn_t, pw2_wp, pw2_b;
  size_t ln_g, ln_b;
  size_t og_slabs = 0, ff1_slabs = 0, tail_slabs = 0, pp_ff1 = 0, pp_tail = 0, pp_og = 0;
  float att_h2[3] = {0.f, 0.f, 0.f};
  float pp_sw_out = 1.f, pp_sw_pw1 = 1.f;
  PpChainSc pp_ff1_sc, pp_tail_sc[2];
  float pp_sw_qkv = 1.f;
  size_t ns_ff1_w1 = 0, ns_ff1_w2 = 0, ns_qkv = 0, ns_out = 0, ns_pw1 = 0, ns_cv_w1 = 0, ns_cv_w2 = 0, ns_ff2_w1 = 0, ns_ff2_w2 = 0;
  bool ns = false;
  bool split = false;
};

struct Plan {
  size_t xa, xb, qkv, ctx, u, dw, enc, amax, logp, pmax, umax, mel, sub, h4, wv, wv_floats, total;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Geometry {
  int Bp, Lb, F, T1, T, nblk;
};

struct Scratch {
  float *xa, *xb, *qkv, *ctx, *u, *dw;
  float* h4 = nullptr;   // [M, 4d] FFN hidden (bf16 GEMM path only: its layers are separate launches)
};

struct CrossAttn {
  const float* enc;   // [B, T_enc, d]
  int T_enc;
  float* kv;          // scratch [B * T_enc, 2d]
  const float* pe;    // [>= T, d]
};

struct StackOff {
  std::vector<BlockOff> blocks;
  size_t proj_w = 0, proj_b = 0, fc_w = 0, fc_b = 0;
  int NT_fc = 0;
  size_t proj_pp = 0;      // dmodel 144: [W ; b] of the projection as a two-term stream (append_pp_plain), 0 = none
  float proj_pp_sw = 1.f;
};

// ---- shared host functions (defined in api.hip) -----------------------------------------------------------
int fail(int code, const char* fmt, ...);
void same_pad(int n, int k, int s, int* out, int* before);
void add_block_expected(std::vector<Expected>& ex, const std::string& p, int d, int H, int hs, int k, bool keras_mha = false);
std::vector<float> pack_p16(const std::function<float(int, int)>& f, int K, int N, int NTpad);
std::vector<float> pack_split32(const std::function<float(int, int)>& f, int K, int N);
std::vector<float> pack_conv2_split(const std::vector<float>& c2, int d);          // conv2 kernel -> subconv_split_ring_kernel fragments
std::vector<float> pack_linear_split(const std::vector<float>& lin, int K, int d);  // subsampling Dense -> sublinear_split_kernel slabs
float half_scale_for(double bound, int max_shift = 100);                            // largest power of two s with bound * s <= 2^15 (0: no usable bound)
std::vector<float> pack_conv2_half(const std::vector<float>& c2, int d, float wscale);   // conv2 kernel * wscale as hi + lo fp16 fragments (two-term scheme)
void append_slabs(std::vector<float>& stream, const std::function<float(int, int)>& f, int K, int N, bool group_major);
// pair-pipelined stream of one chain y += W2 act(W1 x + b1) (fused_pp.hip, tools/gen_pp.py): w1(k, n) with k <= K1 (row K1 = the
// bias), H hidden features, w2(k, n) [H, 144]; and of a plain layer [145, 144 G] in column groups of nine tiles
PpChainSc append_pp_chain(std::vector<float>& stream, const std::function<float(int, int)>& w1aug, int H, const std::function<float(int, int)>& w2,
                          std::vector<float>* plain1 = nullptr, std::vector<float>* plain2 = nullptr);
float append_pp_plain(std::vector<float>& stream, const std::function<float(int, int)>& waug, int groups, std::vector<float>* plain = nullptr);
// W[K, N] as the slab ring of gemm_ring.hip, registered in ab.ring_pairs against the P16 pack at p16_off
void put_ring(ArenaBuilder& ab, size_t p16_off, const std::function<float(int, int)>& f, int K, int N, bool glu);
void put_ring_head(ArenaBuilder& ab, size_t p16_off, const std::function<float(int, int)>& f, int K, int V);
bool ring_packs_wanted(const mi355asr_model* m);
void register_rings(mi355asr_model* m, const ArenaBuilder& ab, const float* base);
// W[144, V] of a class head as the slab stream of head_ld_kernel (fused.hip), registered against its P16 pack
void put_head_slabs(ArenaBuilder& ab, size_t p16_off, const std::function<float(int, int)>& f, int d, int V, const float* bias);
// the head on the slab ring when the handle has a stream for hd.wp (fp32 mode, dmodel 144); -1: not taken
int try_head_ld(const mi355asr_model* m, const GemmArgs& hd, hipStream_t s, float* split_scratch = nullptr);
FftOff pack_fft(ArenaBuilder& ab, const std::vector<float>& re, const std::vector<float>& im, int n_dft, int nb);
BlockOff pack_block(mi355asr_model* m, ArenaBuilder& ab, const std::string& p, int d, int H, int hs, int k, bool keras_mha = false);
BlockDev resolve(const BlockOff& o, const float* base);
bool use_gemm16(const mi355asr_model* m);
bool gemm16_for(const mi355asr_model* m, size_t M);
int launch_gemm16(const mi355asr_model* m, int epi, bool ln, Gemm16Args& g, const float* wp, hipStream_t s);
// next / ff1_done (dmodel-144 fused path): when `next` is given and the output stays in the scratch buffers, the tail kernel
// of this block also runs ff_module_1 + qkv of `next` (one launch) and sets *ff1_done, which the caller passes back in as
// `skip_ff1` for the next block
int run_block(const mi355asr_model* m, const BlockDev& w, const BlockOpts& bo, Scratch& sc, int B, int T, float* out,
              hipStream_t s, const CrossAttn* cross = nullptr, const BlockDev* next = nullptr, bool* ff1_done = nullptr,
              bool skip_ff1 = false);
bool block_takes_pre(const mi355asr_model* m, const BlockDev& w, size_t M);   // run_block(w, M rows) can take BlockOpts::pre_*
void resolve_stack(StackDev& sd, const StackOff& so, const float* base, bool project, int V);   // api_chunk.hip
int finalize_chunk(mi355asr_model* m, hipStream_t s);        // api_chunk.hip
int finalize_translator(mi355asr_model* m, hipStream_t s);   // api_translator.hip

}  // namespace mi355
