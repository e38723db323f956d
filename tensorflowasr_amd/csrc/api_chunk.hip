// ChunkConformer host side (chunk_conformer_blocks.py): offline predict and the streaming entry points with explicit
// caches, behind the mi355asr_chunk_* functions of include/mi355asr.h.
#include "model.h"

namespace {


struct ChunkGeom { int F, T1, T; };

int chunk_geometry(const mi355asr_model* m, int B, int L, ChunkGeom* g) {
  if (B <= 0 || L <= 0) return fail(MI355ASR_EINVAL, "B and L must be positive (B=%d, L=%d)", B, L);
  // 'valid' Spectrogram: the reference left-pads n_dft-1 zeros, then a VALID strided conv (time_frequency.py:106-107)
  g->F = (L - 1) / m->dm.hop + 1;
  // ConvSubsampling(padding='valid'): pad 4 frames in front, VALID 3x3 stride 2, twice (chunk_conformer_blocks.py:60-66)
  g->T1 = (g->F + 4 - 3) / 2 + 1;
  g->T = (g->T1 - 3) / 2 + 1;
  if (g->T < 1) return fail(MI355ASR_EINVAL, "L=%d is too short for the chunk front end", L);
  return 0;
}

struct ChunkPlan {
  size_t xa, xb, qkv, ctx, u, dw, hid, amax, idx, cnt, logp, pmax, mel, sub, h4, total;
};

ChunkPlan make_chunk_plan(const mi355asr_model* m, int B, int F, int T) {
  const int d = m->cfg.dmodel;
  const size_t M = (size_t)B * T;
  ChunkPlan p;
  size_t o = 0;
  auto take = [&](size_t floats) { size_t at = o; o = align256(o + floats * 4); return at; };
  p.xa = take(M * d); p.xb = take(M * d); p.qkv = take(M * 3 * d); p.ctx = take(M * d);
  p.u = take(M * d); p.dw = take(M * d); p.hid = take(M * d);
  p.amax = take(M); p.idx = take(M); p.cnt = take(B);
  const int FT = ceil_div(F, 16);
  p.logp = take((size_t)B * F * m->dm.LP);
  p.pmax = take((size_t)B * std::max(FT * m->dm.NCH_dft, F));
  p.mel = take((size_t)B * F * m->cfg.n_mels);
  p.sub = take(M * m->dm.F2 * d);
  p.h4 = gemm16_for(m, M) ? take(M * 4 * d) : 0;
  p.total = o;
  return p;
}

void add_stack_expected(std::vector<Expected>& ex, const std::string& prefix, const std::string& blk, int nblocks,
                        int d, int H, int hs, int k, bool project, int num_classes) {
  if (project) {
    ex.push_back({prefix + "/project/kernel", {d, d}});
    ex.push_back({prefix + "/project/bias", {d}});
  }
  for (int i = 0; i < nblocks; ++i) add_block_expected(ex, prefix + "/" + blk + std::to_string(i), d, H, hs, k, true);
  if (num_classes > 0) {
    ex.push_back({prefix + "/fully_connected/kernel", {d, num_classes}});
    ex.push_back({prefix + "/fully_connected/bias", {num_classes}});
  }
}


StackOff pack_stack(mi355asr_model* m, ArenaBuilder& ab, const std::string& prefix, const std::string& blk, int nblocks,
                    bool project, int V) {
  const auto& c = m->cfg;
  const int d = c.dmodel;
  StackOff so;
  if (project) {
    const auto& pj = m->host[prefix + "/project/kernel"].data;
    so.proj_w = ab.put(pack_p16([&](int k, int n) { return pj[(size_t)k * d + n]; }, d, d, d / 16));
    so.proj_b = ab.put(m->host[prefix + "/project/bias"].data);
    if (d == 144) {
      const auto& pb = m->host[prefix + "/project/bias"].data;
      std::vector<float> pp;
      so.proj_pp_sw = append_pp_plain(pp, [&](int k, int n) { return k < d ? pj[(size_t)k * d + n] : pb[n]; }, 1);
      so.proj_pp = ab.put(pp);
    }
  }
  for (int i = 0; i < nblocks; ++i)
    so.blocks.push_back(pack_block(m, ab, prefix + "/" + blk + std::to_string(i), d, c.num_heads, c.head_size,
                                   c.kernel_size, true));
  if (V > 0) {
    const auto& fc = m->host[prefix + "/fully_connected/kernel"].data;
    const int ct = gemm_ct(d, EPI_HEAD);
    so.NT_fc = ceil_div(ceil_div(V, 16), ct) * ct;
    so.fc_w = ab.put(pack_p16([&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V, so.NT_fc));
    if (ring_packs_wanted(m)) put_ring_head(ab, so.fc_w, [&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V);
    put_head_slabs(ab, so.fc_w, [&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V, m->host[prefix + "/fully_connected/bias"].data.data());
    so.fc_b = ab.put_padded(m->host[prefix + "/fully_connected/bias"].data.data(), V, (size_t)so.NT_fc * 16);
  }
  return so;
}



// Dense(d->d) [+ blocks] [+ Dense(d->V) with argmax]; input rows at `in`, blocks run in sc.xa
// On return the stack's hidden output is in sc.xa (sc is updated: the blocks ping-pong xa/xb).
// `front` (round 4): a plain layer the caller left to this stack's first block (the chunk front's subsampling Dense: BlockOpts::pre_*
// of the stack's options filled in; `in` is not read then).  A stack's own projection takes the same route when it can.
int run_stack(const mi355asr_model* m, const StackDev& st, const float* in, int B, int T, Scratch& sc,
              float* logits, int32_t* amax, hipStream_t s, const BlockOpts* front = nullptr) {
  const int d = m->cfg.dmodel;
  const int M = B * T;
  BlockOpts first = front ? *front : st.opts;
  if (!front && st.proj_wp && st.proj_pp && !st.blocks.empty() && block_takes_pre(m, st.blocks[0], (size_t)M)) {
    first.pre_x = in; first.pre_pp = st.proj_pp; first.pre_sw = st.proj_pp_sw; first.pre_chunks = 1;
  }
  if (first.pre_pp) {
    // (nothing here: the first block below computes the layer in front of it)
  } else if (st.proj_wp) {
    // on its own: the same two-term stream through pp_sublinear_kernel (bit-identical to the folded form), else fp32 MFMA
    StreamGemmArgs sp{};
    sp.x = in; sp.y = sc.xa; sp.M = M; sp.K = d; sp.NT = d / 16; sp.ldy = d; sp.n_valid = d;
    PROF(MI355ASR_K_CTC_PROJECT);
    if (!(st.proj_pp && m->cfg.gemm_dtype == 0 && !gemm16_for(m, (size_t)M) && launch_pp_sublinear(sp, st.proj_pp, st.proj_pp_sw, s) == 0)) {
      GemmArgs pr{};
      pr.x = in; pr.y = sc.xa; pr.wp = st.proj_wp; pr.bias = st.proj_b;
      pr.M = M; pr.NT = d / 16; pr.ldy = d; pr.n_valid = d; pr.eps = kLnEps;
      LAUNCH_TRY(launch_gemm_rows(d, EPI_BIAS, false, pr, s), "project");
    }
  } else if (in != sc.xa) {
    HIP_TRY(hipMemcpyAsync(sc.xa, in, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
  }
  bool ff1_done = false;                    // the tail of block i also runs ff_module_1 + qkv of block i + 1 (model.h)
  for (size_t i = 0; i < st.blocks.size(); ++i) {
    const bool skip = ff1_done;
    int rc = run_block(m, st.blocks[i], i == 0 ? first : st.opts, sc, B, T, nullptr, s, nullptr,
                       i + 1 < st.blocks.size() ? &st.blocks[i + 1] : nullptr, &ff1_done, skip);
    if (rc) return rc;
  }
  if (st.fc_wp && (logits || amax)) {
    GemmArgs hd{};
    hd.x = sc.xa; hd.y = logits; hd.wp = st.fc_wp; hd.bias = st.fc_b;
    hd.M = M; hd.NT = st.NT_fc; hd.ldy = st.num_classes; hd.n_valid = st.num_classes; hd.eps = kLnEps;
    hd.argmax_out = amax;
    {
      PROF(MI355ASR_K_CTC_HEAD);
      // (the q / k / v buffer is dead behind the last block: the per-range arg-max pairs of a head split over class ranges)
      if (try_head_ld(m, hd, s, sc.qkv) != 0) LAUNCH_TRY(launch_gemm_rows(d, EPI_HEAD, false, hd, s), "fully_connected");
    }
  }
  return 0;
}

}  // namespace

namespace mi355 {

void resolve_stack(StackDev& sd, const StackOff& so, const float* base, bool project, int V) {
  sd.blocks.clear();
  for (const auto& o : so.blocks) sd.blocks.push_back(resolve(o, base));
  if (project) { sd.proj_wp = base + so.proj_w; sd.proj_b = base + so.proj_b; }
  sd.proj_pp = (project && so.proj_pp) ? base + so.proj_pp : nullptr;
  sd.proj_pp_sw = so.proj_pp_sw;
  if (V > 0) { sd.fc_wp = base + so.fc_w; sd.fc_b = base + so.fc_b; sd.NT_fc = so.NT_fc; sd.num_classes = V; }
}

int finalize_chunk(mi355asr_model* m, hipStream_t s) {
  const auto& c = m->cfg;
  const auto& cc = m->ccfg;
  const Dims& dm = m->dm;
  const int d = c.dmodel, nb = dm.nbins;
  ArenaBuilder ab;
  ab.ring_terms = m->cfg.gemm_dtype == 1 ? 1 : 3;
  const auto& re = m->host["front/mel_layer/real_kernels"].data;
  const auto& im = m->host["front/mel_layer/imag_kernels"].data;
  const size_t o_dft = ab.put(pack_p16(
      [&](int k, int n) { const int bin = n >> 1; return (n & 1) ? im[(size_t)k * nb + bin] : re[(size_t)k * nb + bin]; },
      c.n_dft, 2 * nb, dm.NT_dft));
  const FftOff fo = pack_fft(ab, re, im, c.n_dft, nb);
  const auto& f2m = m->host["front/mel_layer/freq2mel"].data;
  const size_t o_mel = ab.put(pack_p16([&](int k, int n) { return k < nb ? f2m[(size_t)k * c.n_mels + n] : 0.f; },
                                       dm.KBm * 16, c.n_mels, dm.NTm));
  const MelBandOff mbo = pack_mel_band(ab, f2m, nb, c.n_mels);
  const size_t o_c1w = ab.put(m->host["front/conv_subsampling/conv1/kernel"].data);
  const size_t o_c1b = ab.put(m->host["front/conv_subsampling/conv1/bias"].data);
  const auto& c2 = m->host["front/conv_subsampling/conv2/kernel"].data;
  const size_t o_c2w = ab.put(pack_p16(
      [&](int kp, int n) {
        const int kb = kp / 16, r = kp % 16, cb = kb / 9, q = kb % 9;
        return c2[((size_t)q * d + (16 * cb + r)) * d + n];
      },
      9 * d, d, d / 16));
  const size_t o_c2b = ab.put(m->host["front/conv_subsampling/conv2/bias"].data);
  const auto& lin = m->host["front/conv_subsampling/linear/kernel"].data;
  const size_t o_lw = ab.put(pack_p16([&](int k, int n) { return lin[(size_t)k * d + n]; }, dm.F2 * d, d, d / 16));
  const size_t o_lb = ab.put(m->host["front/conv_subsampling/linear/bias"].data);
  // round 3: the front runs the split-bf16 subsampling kernels of the offline encoder (the explicit front padding and the
  // VALID geometry are index offsets of the same kernels: subconv.hip reads pt1 / pf1 / pt2 / pf2 / T1 / F1 from its arguments)
  const bool split_front = d == 144 && c.gemm_dtype == 0;
  const size_t o_c2s = split_front ? ab.put(pack_conv2_split(c2, d)) : 0;
  const size_t o_lws = split_front && (dm.F2 * d) % 32 == 0 ? ab.put(pack_linear_split(lin, dm.F2 * d, d)) : 0;
  // round 4: the two-term fp16 forms -- conv2 as hi + lo of kernel * 2^k (its operand scale comes from the batch's largest |mel|
  // at run time: the valid frontend's log10 features have no static bound), the Dense as the stream of pp_sublinear_kernel
  size_t o_c2h = 0, o_lpp = 0;
  float c2_ws = 0.f, c1_l1 = 0.f, c1_bmax = 0.f, c1_ws = 0.f, lin_pp_sw = 1.f;
  if (split_front) {
    const auto& w1 = m->host["front/conv_subsampling/conv1/kernel"].data;
    const auto& b1 = m->host["front/conv_subsampling/conv1/bias"].data;
    double wmax = 0.0, l1max = 0.0, bmax = 0.0;
    for (int ch = 0; ch < d; ++ch) {
      double sum = 0.0;
      for (int t = 0; t < 9; ++t) sum += std::fabs((double)w1[(size_t)t * d + ch]);
      l1max = std::max(l1max, sum);
      bmax = std::max(bmax, std::fabs((double)b1[ch]));
    }
    for (float v : c2) wmax = std::max(wmax, std::fabs((double)v));
    c2_ws = half_scale_for(wmax);
    if (c2_ws > 0.f && l1max > 0.0) {
      o_c2h = ab.put(pack_conv2_half(c2, d, c2_ws));
      c1_l1 = (float)(l1max * (1.0 + 1e-6));
      c1_bmax = (float)(bmax * (1.0 + 1e-6));
      double w1max = 0.0;                          // conv1 on the matrix pipe (subconv.hip, C1M): the mel scale is taken at run time
      for (float v : w1) w1max = std::max(w1max, std::fabs((double)v));
      c1_ws = half_scale_for(w1max);
    }
    if (o_lws) {
      const auto& lb = m->host["front/conv_subsampling/linear/bias"].data;
      std::vector<float> pp;
      lin_pp_sw = append_pp_plain(pp, [&](int k, int n) {
        const int f = n / d, col = n - f * d;
        return k < d ? lin[((size_t)f * d + k) * d + col] : (f == 0 ? lb[col] : 0.f);
      }, dm.F2);
      o_lpp = ab.put(pp);
    }
  }
  StackOff e = pack_stack(m, ab, "encoder", "chunk_conformer_block_", cc.enc_num_blocks, false, 0);
  StackOff pk = pack_stack(m, ab, "picker", "block_", cc.picker_num_blocks, true, cc.picker_num_classes);
  StackOff hp = pack_stack(m, ab, "helper", "block_", cc.helper_num_blocks, false, 0);
  StackOff dc = pack_stack(m, ab, "decoder", "block_", cc.decoder_num_blocks, true, cc.decoder_num_classes);
  if (m->arena) { (void)hipFree(m->arena); m->arena = nullptr; }
  HIP_TRY(hipMalloc((void**)&m->arena, ab.buf.size() * sizeof(float)));
  m->arena_floats = ab.buf.size();
  HIP_TRY(hipMemcpyAsync(m->arena, ab.buf.data(), ab.buf.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  const float* base = m->arena;
  m->ring_of.clear();
  register_rings(m, ab, base);
  m->dft_wp = base + o_dft; m->mel_wp = base + o_mel;
  m->fft_ok = fo.ok;
  use_mel_band(m, mbo, base);
  m->fft_w1p = base + fo.w1; m->fft_w2p = base + fo.w2; m->fft_twc = base + fo.twc; m->fft_tws = base + fo.tws;
  m->fft_w1s = base + fo.w1s; m->fft_w2s = base + fo.w2s; m->fft_w1h = base + fo.w1h; m->fft_w2h = base + fo.w2h;
  m->fft_win = base + fo.win;
  m->c1_w = base + o_c1w; m->c1_b = base + o_c1b; m->c2_wp = base + o_c2w; m->c2_b = base + o_c2b;
  m->lin_wp = base + o_lw; m->lin_b = base + o_lb;
  m->c2_wsplit = o_c2s ? base + o_c2s : nullptr;
  m->lin_wsplit = o_lws ? base + o_lws : nullptr;
  m->c2_whalf = o_c2h ? base + o_c2h : nullptr; m->c2_wscale = c2_ws; m->c1_l1 = c1_l1; m->c1_bmax = c1_bmax; m->c1_wscale = c1_ws;
  m->lin_pp = o_lpp ? base + o_lpp : nullptr; m->lin_pp_sw = lin_pp_sw;
  resolve_stack(m->c_enc, e, base, false, 0);
  resolve_stack(m->c_picker, pk, base, true, cc.picker_num_classes);
  resolve_stack(m->c_helper, hp, base, false, 0);
  resolve_stack(m->c_decoder, dc, base, true, cc.decoder_num_classes);
  m->finalized = true;
  return 0;
}

}  // namespace mi355

namespace {
// =======================================================================================================
// ChunkConformer streaming (single stream, explicit caches; chunk_conformer_blocks.py:72-91, 206-220, 294-310,
// 381-388, 447-458, 530-560, 641-672, 750-770).  The caller owns the caches and trims them (valid / window
// slicing is plain indexing, done in tensorflowasr_amd/models.py as the reference does it in Python).
// =======================================================================================================
struct StreamPlan {
  size_t xa, xb, qkv, ctx, u, dw, amax, logp, pmax, mel, sub, total;
};
// N = longest [cache ; new] row count of any module, F = mel frames of the wav buffer, Fs = rows of [sub cache ; mel]
StreamPlan make_stream_plan(const mi355asr_model* m, int N, int F, int Fs) {
  const size_t d = m->cfg.dmodel;
  StreamPlan p;
  size_t o = 0;
  auto take = [&](size_t floats) { size_t at = o; o = align256(o + floats * 4); return at; };
  p.xa = take(N * d); p.xb = take(N * d); p.qkv = take((size_t)N * 3 * d); p.ctx = take(N * d);
  p.u = take(N * d); p.dw = take(N * d); p.amax = take(N);
  const int FT = ceil_div(std::max(F, 1), 16);
  p.logp = take((size_t)std::max(F, 1) * m->dm.LP);
  p.pmax = take((size_t)std::max(FT * m->dm.NCH_dft, F));
  p.mel = take((size_t)std::max(F, 1) * m->cfg.n_mels);
  p.sub = take((size_t)std::max(Fs, 1) * m->dm.F2 * d);
  p.total = o;
  return p;
}

// ChunkConformerBlock.stream_call (:381-388) for one stream: x (T rows) in sc.xa -> result in sc.xa.
// new_mha [Cm+T, d] / new_cnn [Cc+T, d] receive [cache ; module input] (untrimmed).
int run_block_stream(const mi355asr_model* m, const BlockDev& w, const BlockOpts& bo, Scratch& sc, int T,
                     const float* mha_cache, int Cm, const float* cnn_cache, int Cc, float* new_mha, float* new_cnn,
                     hipStream_t s) {
  const int d = m->cfg.dmodel, H = m->cfg.num_heads, hs = m->cfg.head_size;
  const int N = Cm + T, Nc = Cc + T;
  const size_t row = (size_t)d * 4;
  // ff_module_1: xb = xa + fc * FFN(LN(xa))
  Chain2Args f1{};
  f1.x = sc.xa; f1.res = sc.xa; f1.y = sc.xb;
  f1.ln_g = w.ff_ln_g[0]; f1.ln_b = w.ff_ln_b[0];
  f1.w1p = w.ff_w1p[0]; f1.b1 = w.ff_b1[0]; f1.w2p = w.ff_w2p[0]; f1.b2 = w.ff_b2[0];
  f1.scale = bo.fc; f1.eps = kLnEps; f1.M = T;
  { PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_chain2(d, 0, f1, s), "ff_module_1"); }
  // new_mha = [cache ; xb]; q/k/v of every row of it (the queries are its last T rows, :209-214)
  if (Cm > 0) HIP_TRY(hipMemcpyAsync(new_mha, mha_cache, Cm * row, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(new_mha + (size_t)Cm * d, sc.xb, T * row, hipMemcpyDeviceToDevice, s));
  GemmArgs q{};
  q.x = new_mha; q.y = sc.qkv; q.ln_g = w.att_ln_g; q.ln_b = w.att_ln_b; q.wp = w.qkv_wp; q.bias = w.qkv_b;
  q.M = N; q.NT = 3 * d / 16; q.ldy = 3 * d; q.n_valid = 3 * d; q.eps = kLnEps;
  q.qscale = 1.0f / std::sqrt((float)hs); q.qtiles = d / 16;
  { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm_rows(d, EPI_QKV, true, q, s), "qkv projection"); }
  AttnArgs at{};
  at.q = sc.qkv + (size_t)Cm * 3 * d; at.k = sc.qkv + d; at.v = sc.qkv + 2 * d; at.ctx = sc.ctx;
  at.B = 1; at.Tq = T; at.Tk = N; at.H = H; at.D = d; at.ldq = 3 * d; at.ldk = 3 * d;
  at.win_front = bo.win_front; at.win_back = bo.win_back; at.q_off = Cm;
  { PROF(MI355ASR_K_ATTN); LAUNCH_TRY(launch_attention(hs, at, s), "attention"); }
  GemmArgs op{};
  op.x = sc.ctx; op.y = sc.xa; op.res = sc.xb; op.wp = w.out_wp; op.bias = w.out_b;
  op.M = T; op.NT = d / 16; op.ldy = d; op.n_valid = d; op.eps = kLnEps;
  { PROF(MI355ASR_K_ATTN_OUT); LAUNCH_TRY(launch_gemm_rows(d, EPI_RESIDUAL, false, op, s), "attention out-projection"); }
  // new_cnn = [cache ; xa]; the conv module runs over all of it, its last T rows are kept (:297-306)
  if (Cc > 0) HIP_TRY(hipMemcpyAsync(new_cnn, cnn_cache, Cc * row, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(new_cnn + (size_t)Cc * d, sc.xa, T * row, hipMemcpyDeviceToDevice, s));
  GemmArgs g{};
  g.x = new_cnn; g.y = sc.u; g.ln_g = w.cv_ln_g; g.ln_b = w.cv_ln_b; g.wp = w.pw1_wp; g.bias = w.pw1_b;
  g.M = Nc; g.NT = 2 * d / 16; g.ldy = d; g.n_valid = d; g.eps = kLnEps;
  { PROF(MI355ASR_K_PW1_GLU); LAUNCH_TRY(launch_gemm_rows(d, EPI_GLU, true, g, s), "pw_conv_1 + GLU"); }
  DwArgs dwa{};
  dwa.u = sc.u; dwa.y = sc.dw; dwa.wd = w.dw_w; dwa.B = 1; dwa.T = Nc; dwa.D = d;
  dwa.pad_left = bo.causal ? bo.ksz - 1 : (bo.ksz - 1) / 2;
  { PROF(MI355ASR_K_DWCONV); LAUNCH_TRY(launch_dwconv(bo.ksz, dwa, s), "depthwise conv"); }
  Chain2Args cv{};
  cv.x = sc.dw + (size_t)Cc * d; cv.res = sc.xa; cv.y = sc.xb;
  cv.w1p = w.pc_w1p; cv.b1 = w.pc_b1; cv.aff_s = w.bn_s; cv.aff_t = w.bn_t; cv.w2p = w.pw2_wp; cv.b2 = w.pw2_b;
  cv.scale = 1.0f; cv.eps = kLnEps; cv.M = T;
  { PROF(MI355ASR_K_CONV_TAIL); LAUNCH_TRY(launch_chain2(d, 1, cv, s), "conv module tail"); }
  Chain2Args f2{};
  f2.x = sc.xb; f2.res = sc.xb; f2.y = sc.xa;
  f2.ln_g = w.ff_ln_g[1]; f2.ln_b = w.ff_ln_b[1];
  f2.w1p = w.ff_w1p[1]; f2.b1 = w.ff_b1[1]; f2.w2p = w.ff_w2p[1]; f2.b2 = w.ff_b2[1];
  f2.fln_g = w.ln_g; f2.fln_b = w.ln_b;
  f2.scale = bo.fc; f2.eps = kLnEps; f2.M = T;
  { PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_chain2(d, 0, f2, s), "ff_module_2 + LayerNorm"); }
  return 0;
}

// mel frames of a wav buffer / rows after the two VALID stride-2 convs over [sub cache ; last chunk_num mel frames]
void front_stream_shape(const mi355asr_model* m, int Lw, int S, int chunk_num, int* F, int* nf, int* T1, int* T2, int* Tout) {
  *F = (Lw - 1) / m->dm.hop + 1;
  *nf = std::min(*F, chunk_num);
  const int rows = S + *nf;
  *T1 = rows >= 3 ? (rows - 3) / 2 + 1 : 0;
  *T2 = *T1 >= 3 ? (*T1 - 3) / 2 + 1 : 0;
  *Tout = std::min(*T2, chunk_num / m->cfg.reduction_factor);
}

const StackDev* chunk_stack_by_id(const mi355asr_model* m, int id) {
  switch (id) {
    case 0: return &m->c_enc;
    case 1: return &m->c_picker;
    case 2: return &m->c_helper;
    case 3: return &m->c_decoder;
  }
  return nullptr;
}

}  // namespace

extern "C" {

int mi355asr_chunk_create(const mi355asr_chunk_config* cfg, mi355asr_model** out) {
  if (!cfg || !out) return fail(MI355ASR_EINVAL, "null argument");
  const auto& c = *cfg;
  if (c.dmodel != 144) return fail(MI355ASR_EINVAL, "ChunkConformer: dmodel=%d, kernels instantiated for 144", c.dmodel);
  if (c.num_heads * c.head_size != c.dmodel || c.head_size != 36)
    return fail(MI355ASR_EINVAL, "ChunkConformer: need num_heads*head_size == dmodel and head_size 36");
  if (c.kernel_size != 32 && c.kernel_size != 5) return fail(MI355ASR_EINVAL, "kernel_size=%d unsupported", c.kernel_size);
  if (c.reduction_factor != 4 || c.n_dft != 1024 || (c.n_mels != 80 && c.n_mels != 128))
    return fail(MI355ASR_EINVAL, "ChunkConformer front: reduction_factor 4, n_dft 1024, n_mels 80|128 only");
  if (c.picker_num_classes < 2 || c.decoder_num_classes < 2) return fail(MI355ASR_EINVAL, "num_classes must be >= 2");
  if (c.enc_win_front < 0 || c.picker_win_front < 0 || c.helper_win_front < 0 || c.decoder_win_front < 0 ||
      c.enc_win_back < 0 || c.picker_win_back < 0 || c.helper_win_back < 0 || c.decoder_win_back < 0)
    return fail(MI355ASR_EINVAL, "window sizes must be non-negative");
  auto* m = new mi355asr_model();
  m->is_chunk = true;
  m->ccfg = c;
  std::memset(&m->cfg, 0, sizeof(m->cfg));
  m->cfg.dmodel = c.dmodel; m->cfg.head_size = c.head_size; m->cfg.num_heads = c.num_heads;
  m->cfg.kernel_size = c.kernel_size; m->cfg.fc_factor = c.fc_factor; m->cfg.reduction_factor = 4;
  m->cfg.n_mels = c.n_mels; m->cfg.sample_rate = c.sample_rate; m->cfg.stride_ms = c.stride_ms; m->cfg.n_dft = c.n_dft;
  Dims& dm = m->dm;
  dm.hop = c.stride_ms * c.sample_rate / 1000;
  if (dm.hop <= 0) { delete m; return fail(MI355ASR_EINVAL, "stride_ms*sample_rate/1000 must be positive"); }
  dm.nbins = c.n_dft / 2 + 1;
  dm.NT_dft = ceil_div(ceil_div(2 * dm.nbins, 16), 13) * 13;
  dm.NCH_dft = dm.NT_dft / 13;
  dm.LP = ceil_div(8 * dm.NT_dft, 16) * 16;
  dm.KBm = ceil_div(ceil_div(dm.nbins, 16), 2) * 2;
  dm.NTm = c.n_mels / 16;
  dm.st1 = 2;
  dm.pf1 = 2; dm.pf2 = 0;                          // tf.pad(..., [2,2]) on the mel axis, then VALID convs
  dm.F1 = (c.n_mels + 4 - 3) / 2 + 1;
  dm.F2 = (dm.F1 - 3) / 2 + 1;
  const int d = c.dmodel;
  auto& ex = m->expected;
  ex.push_back({"front/mel_layer/real_kernels", {c.n_dft, 1, 1, dm.nbins}});
  ex.push_back({"front/mel_layer/imag_kernels", {c.n_dft, 1, 1, dm.nbins}});
  ex.push_back({"front/mel_layer/freq2mel", {dm.nbins, c.n_mels}});
  ex.push_back({"front/conv_subsampling/conv1/kernel", {3, 3, 1, d}});
  ex.push_back({"front/conv_subsampling/conv1/bias", {d}});
  ex.push_back({"front/conv_subsampling/conv2/kernel", {3, 3, d, d}});
  ex.push_back({"front/conv_subsampling/conv2/bias", {d}});
  ex.push_back({"front/conv_subsampling/linear/kernel", {dm.F2 * d, d}});
  ex.push_back({"front/conv_subsampling/linear/bias", {d}});
  add_stack_expected(ex, "encoder", "chunk_conformer_block_", c.enc_num_blocks, d, c.num_heads, c.head_size, c.kernel_size, false, 0);
  add_stack_expected(ex, "picker", "block_", c.picker_num_blocks, d, c.num_heads, c.head_size, c.kernel_size, true, c.picker_num_classes);
  add_stack_expected(ex, "helper", "block_", c.helper_num_blocks, d, c.num_heads, c.head_size, c.kernel_size, false, 0);
  add_stack_expected(ex, "decoder", "block_", c.decoder_num_blocks, d, c.num_heads, c.head_size, c.kernel_size, true, c.decoder_num_classes);
  auto opts = [&](int wf, int wb) { BlockOpts o; o.ksz = c.kernel_size; o.fc = c.fc_factor; o.win_front = wf; o.win_back = wb; o.causal = true; return o; };
  m->c_enc.opts = opts(c.enc_win_front, c.enc_win_back);
  m->c_picker.opts = opts(c.picker_win_front, c.picker_win_back);
  m->c_helper.opts = opts(c.helper_win_front, c.helper_win_back);
  m->c_decoder.opts = opts(c.decoder_win_front, c.decoder_win_back);
  *out = m;
  return 0;
}

int mi355asr_chunk_out_frames(const mi355asr_model* m, int32_t L, int32_t* mel_frames, int32_t* enc_frames) {
  if (!m || !m->is_chunk) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle");
  ChunkGeom g;
  int rc = chunk_geometry(m, 1, L, &g);
  if (rc) return rc;
  if (mel_frames) *mel_frames = g.F;
  if (enc_frames) *enc_frames = g.T;
  return 0;
}

int mi355asr_chunk_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t L, size_t* bytes) {
  if (!m || !m->is_chunk || !bytes) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle / null argument");
  ChunkGeom g;
  int rc = chunk_geometry(m, B, L, &g);
  if (rc) return rc;
  *bytes = make_chunk_plan(m, B, g.F, g.T).total;
  return 0;
}

int mi355asr_chunk_predict(mi355asr_model* m, const float* wav, int32_t B, int32_t L, const mi355asr_chunk_outputs* outs,
                           int32_t* n_picked, int32_t* t_pick, void* ws_, size_t ws_bytes, void* stream) {
  if (!m || !m->is_chunk) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle");
  if (!m->finalized) return fail(MI355ASR_ESTATE, "weights not finalised: call mi355asr_finalize_weights first");
  if (!wav || !outs || !n_picked || !t_pick || !ws_) return fail(MI355ASR_EINVAL, "null argument");
  ChunkGeom g;
  int rc = chunk_geometry(m, B, L, &g);
  if (rc) return rc;
  const ChunkPlan p = make_chunk_plan(m, B, g.F, g.T);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  char* ws = (char*)ws_;
  hipStream_t s = (hipStream_t)stream;
  const auto& c = m->cfg;
  const int d = c.dmodel, T = g.T;
  const size_t act = (size_t)B * T * d * 4;
  Scratch sc{(float*)(ws + p.xa), (float*)(ws + p.xb), (float*)(ws + p.qkv),
             (float*)(ws + p.ctx), (float*)(ws + p.u), (float*)(ws + p.dw)};
  sc.h4 = (float*)(ws + p.h4);
  BlockOpts enc_front;
  bool dense_deferred = false;
  // ---- front: valid Melspectrogram (log10, no max-normalisation) + left-padded VALID ConvSubsampling
  {
    const int FT = ceil_div(g.F, 16);
    StftArgs st{};
    st.wav = wav; st.logp = (float*)(ws + p.logp); st.pmax = (float*)(ws + p.pmax); st.wp = m->dft_wp;
    st.B = B; st.L = L; st.F = g.F; st.hop = m->dm.hop; st.pad_left = c.n_dft - 1; st.n_dft = c.n_dft;
    st.NT = m->dm.NT_dft; st.LP = m->dm.LP; st.nbins = m->dm.nbins; st.FT = FT; st.NCH = m->dm.NCH_dft;
    st.db10 = 0;                                    // chunk_amplitude_to_decibel: log10 only (backend_keras.py:25-37)
    if (m->fft_ok) {
      FftStftArgs fa{wav, st.logp, st.pmax, m->fft_w1p, m->fft_w2p, m->fft_twc, m->fft_tws, m->fft_win,
                     B, L, g.F, m->dm.hop, c.n_dft - 1, m->dm.LP, 0};
    fa.w1s = m->fft_w1s; fa.w2s = m->fft_w2s; fa.w1h = m->fft_w1h; fa.w2h = m->fft_w2h;
      PROF(MI355ASR_K_STFT);
      LAUNCH_TRY(launch_fft_stft(fa, s), "stft (valid, fft)");
    } else {
      PROF(MI355ASR_K_STFT);
      LAUNCH_TRY(launch_stft(st, s), "stft (valid)");
    }
    MelArgs me{};
    me.logp = st.logp; me.umax = nullptr; me.mel = (float*)(ws + p.mel); me.wp = m->mel_wp;
    me.B = B; me.F = g.F; me.LP = m->dm.LP; me.nbins = m->dm.nbins; me.KBm = m->dm.KBm; me.NTm = m->dm.NTm;
    me.NM = c.n_mels; me.FT = FT; me.floor_db = 0.f;
    // round 4: the valid frontend's log10 features have no static bound; the banded mel kernel leaves each utterance's largest
    // |mel| in the first B words of the (by now consumed) per-frame maxima, and the two-term subsampling conv scales by it
    // (round 5: per utterance, not per batch -- a quiet utterance next to a loud one keeps its own 2^-22, and B = 1 == B = N)
    unsigned* melmax = (m->mel_band && m->c2_whalf && m->c1_l1 > 0.f) ? (unsigned*)(ws + p.pmax) : nullptr;
    if (melmax) { HIP_TRY(hipMemsetAsync(melmax, 0, sizeof(unsigned) * (size_t)B, s)); me.absmax = melmax; }
    { PROF(MI355ASR_K_MEL); LAUNCH_TRY(launch_mel_auto(m, me, s), "mel (valid)"); }
    SubConvArgs sa{};
    sa.mel = me.mel; sa.out = (float*)(ws + p.sub); sa.w1 = m->c1_w; sa.b1 = m->c1_b; sa.w2p = m->c2_wp; sa.b2 = m->c2_b;
    sa.w2s = m->c2_wsplit;
    static const bool three = mi355_env("MI355ASR_SUBCONV_TERMS", -1) == 3;
    if (melmax && me.absmax && !three) { sa.w2h = m->c2_whalf; sa.h_wscale = m->c2_wscale; sa.h_melmax = melmax; sa.h_l1 = m->c1_l1; sa.h_bmax = m->c1_bmax; sa.c1_wscale = m->c1_wscale; }
    sa.B = B; sa.F = g.F; sa.NM = c.n_mels; sa.T1 = g.T1; sa.F1 = m->dm.F1; sa.T2 = T; sa.F2 = m->dm.F2;
    sa.st1 = 2; sa.pt1 = 4; sa.pf1 = 2; sa.pt2 = 0; sa.pf2 = 0;
    { PROF(MI355ASR_K_SUBCONV); LAUNCH_TRY(launch_subconv(d, sa, s), "conv subsampling (valid)"); }
    StreamGemmArgs lg{};
    lg.x = sa.out; lg.y = sc.xa; lg.wp = m->lin_wp; lg.bias = m->lin_b;
    lg.M = B * T; lg.K = m->dm.F2 * d; lg.NT = d / 16; lg.ldy = d; lg.n_valid = d;
    // round 4: the Dense rides in the encoder's first ff_module_1 + qkv launch when both run on the two-term stream and
    // nobody asked for the front's output (as encoder_impl in api.hip)
    if (!outs->front_out && !m->c_enc.proj_wp && !m->c_enc.blocks.empty() && m->lin_wsplit && lg.M >= 4096 && m->lin_pp &&
        pp_sublinear_ok(lg, m->lin_pp) && block_takes_pre(m, m->c_enc.blocks[0], (size_t)lg.M)) {
      enc_front = m->c_enc.opts;
      enc_front.pre_x = sa.out; enc_front.pre_pp = m->lin_pp; enc_front.pre_sw = m->lin_pp_sw; enc_front.pre_chunks = m->dm.F2;
      dense_deferred = true;
    } else {
      PROF(MI355ASR_K_SUBLINEAR);
      // the split-bf16 ring-DMA Dense from 4096 rows on (as run_subsampling in api.hip), else the fp32-MFMA stream kernel
      // (round 4: the two-term stream with a scale per token and 144-wide chunk first, as run_subsampling in api.hip)
      if (!(m->lin_wsplit && lg.M >= 4096 && ((m->lin_pp && launch_pp_sublinear(lg, m->lin_pp, m->lin_pp_sw, s) == 0) ||
                                              launch_sublinear_split(lg, m->lin_wsplit, s) == 0)))
        LAUNCH_TRY(launch_stream_gemm(d, lg, s), "subsampling linear");
    }
  }
  if (outs->front_out) HIP_TRY(hipMemcpyAsync(outs->front_out, sc.xa, act, hipMemcpyDeviceToDevice, s));
  // ---- encoder
  rc = run_stack(m, m->c_enc, sc.xa, B, T, sc, nullptr, nullptr, s, dense_deferred ? &enc_front : nullptr);
  if (rc) return rc;
  if (outs->enc_out) HIP_TRY(hipMemcpyAsync(outs->enc_out, sc.xa, act, hipMemcpyDeviceToDevice, s));
  // ---- phone picker: logits (optional) + per-frame argmax, hidden = block output
  int32_t* amax = (int32_t*)(ws + p.amax);
  rc = run_stack(m, m->c_picker, sc.xa, B, T, sc, outs->picker_logits, amax, s);
  if (rc) return rc;
  float* hid = (float*)(ws + p.hid);
  HIP_TRY(hipMemcpyAsync(hid, sc.xa, act, hipMemcpyDeviceToDevice, s));
  if (outs->picker_hidden) HIP_TRY(hipMemcpyAsync(outs->picker_hidden, sc.xa, act, hipMemcpyDeviceToDevice, s));
  // ---- feature_pick
  int32_t* idx = (int32_t*)(ws + p.idx);
  int32_t* cnt = (int32_t*)(ws + p.cnt);
  PickArgs pa{amax, idx, cnt, B, T, m->ccfg.picker_num_classes - 1};
  LAUNCH_TRY(launch_pick(pa, s), "feature_pick compaction");
  HIP_TRY(hipMemcpyAsync(n_picked, cnt, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));    // the batch maximum sizes everything downstream (dynamic shape in the reference)
  int Tp = 0;
  for (int b = 0; b < B; ++b) Tp = std::max(Tp, n_picked[b]);
  *t_pick = Tp;
  if (Tp == 0) return 0;               // nothing picked: the reference would build [B, 0, V] logits
  GatherArgs ga{hid, idx, cnt, sc.xa, B, T, Tp, d};
  LAUNCH_TRY(launch_gather(ga, s), "feature_pick gather");
  const size_t actp = (size_t)B * Tp * d * 4;
  if (outs->picked) HIP_TRY(hipMemcpyAsync(outs->picked, sc.xa, actp, hipMemcpyDeviceToDevice, s));
  // ---- context helper, text decoder
  rc = run_stack(m, m->c_helper, sc.xa, B, Tp, sc, nullptr, nullptr, s);
  if (rc) return rc;
  if (outs->helper_out) HIP_TRY(hipMemcpyAsync(outs->helper_out, sc.xa, actp, hipMemcpyDeviceToDevice, s));
  // (no arg-max unless asked for: the text logits go to top-n / the beam search, and a head without it needs no combine launch)
  rc = run_stack(m, m->c_decoder, sc.xa, B, Tp, sc, outs->text_logits, outs->text_logits ? outs->text_argmax : (outs->text_argmax ? outs->text_argmax : amax), s);
  return rc;
}

int mi355asr_frame_argmax(const float* x, int32_t M, int32_t V, int32_t* out, void* stream) {
  if (!x || !out) return fail(MI355ASR_EINVAL, "null argument");
  if (M < 0 || V < 1) return fail(MI355ASR_EINVAL, "need M >= 0, V >= 1 (got %d, %d)", M, V);
  LAUNCH_TRY(launch_row_argmax(x, out, M, V, (hipStream_t)stream), "frame argmax");
  return 0;
}

// ---- feature_pick as its own entry point (the streaming path calls it between picker and decoder) --------------------
int mi355asr_feature_pick_count(const float* ctc, int32_t B, int32_t T, int32_t V, int32_t* idx, int32_t* cnt,
                                int32_t* counts_host, void* stream) {
  if (!ctc || !idx || !cnt || !counts_host) return fail(MI355ASR_EINVAL, "null argument");
  if (B < 1 || T < 0 || V < 2) return fail(MI355ASR_EINVAL, "need B >= 1, T >= 0, V >= 2 (got %d, %d, %d)", B, T, V);
  hipStream_t s = (hipStream_t)stream;
  if (T == 0) {
    for (int b = 0; b < B; ++b) counts_host[b] = 0;
    HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)B * sizeof(int32_t), s));
    return 0;
  }
  // the per-frame argmax is written to idx and compacted in place: pick_kernel reads a 64-frame group before it writes,
  // and a kept frame t goes to a slot <= t
  LAUNCH_TRY(launch_row_argmax(ctc, idx, B * T, V, s), "feature_pick argmax");
  PickArgs pa{idx, idx, cnt, B, T, V - 1};
  LAUNCH_TRY(launch_pick(pa, s), "feature_pick compaction");
  HIP_TRY(hipMemcpyAsync(counts_host, cnt, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int mi355asr_feature_pick_gather(const float* hidden, const float* ctc, const int32_t* idx, const int32_t* cnt, int32_t B,
                                 int32_t T, int32_t d, int32_t V, int32_t Tp, float* feat_out, float* ctc_out, void* stream) {
  if (!hidden || !idx || !cnt || !feat_out || (ctc_out && !ctc)) return fail(MI355ASR_EINVAL, "null argument");
  if (B < 1 || T < 0 || Tp < 0 || d < 1 || (ctc_out && V < 1))
    return fail(MI355ASR_EINVAL, "need B >= 1, T, Tp >= 0, d >= 1 (got B=%d T=%d Tp=%d d=%d V=%d)", B, T, Tp, d, V);
  hipStream_t s = (hipStream_t)stream;
  if (Tp == 0) return 0;
  GatherArgs ga{hidden, idx, cnt, feat_out, B, T, Tp, d};
  LAUNCH_TRY(launch_gather(ga, s), "feature_pick gather (hidden)");
  if (ctc_out) {
    GatherArgs gc{ctc, idx, cnt, ctc_out, B, T, Tp, V};
    LAUNCH_TRY(launch_gather(gc, s), "feature_pick gather (ctc)");
  }
  return 0;
}

// ---- ChunkConformer streaming ------------------------------------------------------------------------------
int mi355asr_chunk_front_stream_shape(const mi355asr_model* m, int32_t Lw, int32_t S, int32_t chunk_num, int32_t* nf,
                                      int32_t* t_out) {
  if (!m || !m->is_chunk || !nf || !t_out) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle / null argument");
  if (Lw < 1 || S < 0 || chunk_num < m->cfg.reduction_factor)
    return fail(MI355ASR_EINVAL, "need Lw >= 1, S >= 0, chunk_num >= reduction_factor (got %d, %d, %d)", Lw, S, chunk_num);
  int F, T1, T2;
  front_stream_shape(m, Lw, S, chunk_num, &F, nf, &T1, &T2, t_out);
  return 0;
}

int mi355asr_chunk_stream_workspace_bytes(const mi355asr_model* m, int32_t max_rows, int32_t Lw, int32_t S,
                                          int32_t chunk_num, size_t* bytes) {
  if (!m || !m->is_chunk || !bytes) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle / null argument");
  if (max_rows < 1 || Lw < 1 || S < 0 || chunk_num < 1) return fail(MI355ASR_EINVAL, "bad sizes");
  *bytes = make_stream_plan(m, max_rows, (Lw - 1) / m->dm.hop + 1, S + chunk_num).total;
  return 0;
}

int mi355asr_chunk_front_stream(mi355asr_model* m, const float* wav, int32_t Lw, const float* sub_cache, int32_t S,
                                int32_t chunk_num, float* front_out, float* new_sub, void* ws_, size_t ws_bytes,
                                void* stream) {
  if (!m || !m->is_chunk) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle");
  if (!m->finalized) return fail(MI355ASR_ESTATE, "weights not finalised: call mi355asr_finalize_weights first");
  if (!wav || !front_out || !new_sub || !ws_ || (S > 0 && !sub_cache)) return fail(MI355ASR_EINVAL, "null argument");
  if (Lw < 1 || S < 0 || chunk_num < m->cfg.reduction_factor)
    return fail(MI355ASR_EINVAL, "need Lw >= 1, S >= 0, chunk_num >= reduction_factor (got %d, %d, %d)", Lw, S, chunk_num);
  const auto& c = m->cfg;
  const int d = c.dmodel;
  int F, nf, T1, T2, Tout;
  front_stream_shape(m, Lw, S, chunk_num, &F, &nf, &T1, &T2, &Tout);
  const StreamPlan p = make_stream_plan(m, std::max(Tout, 1), F, S + nf);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  char* ws = (char*)ws_;
  hipStream_t s = (hipStream_t)stream;
  // valid Melspectrogram of the whole buffer (each call left-pads n_dft-1 zeros, as the layer does), last nf frames
  const int FT = ceil_div(F, 16);
  float* logp = (float*)(ws + p.logp);
  float* mel = (float*)(ws + p.mel);
  if (m->fft_ok) {
    FftStftArgs fa{wav, logp, (float*)(ws + p.pmax), m->fft_w1p, m->fft_w2p, m->fft_twc, m->fft_tws, m->fft_win,
                   1, Lw, F, m->dm.hop, c.n_dft - 1, m->dm.LP, 0};
    fa.w1s = m->fft_w1s; fa.w2s = m->fft_w2s; fa.w1h = m->fft_w1h; fa.w2h = m->fft_w2h;
    PROF(MI355ASR_K_STFT);
    LAUNCH_TRY(launch_fft_stft(fa, s), "stft (valid, fft)");
  } else {
    StftArgs st{};
    st.wav = wav; st.logp = logp; st.pmax = (float*)(ws + p.pmax); st.wp = m->dft_wp;
    st.B = 1; st.L = Lw; st.F = F; st.hop = m->dm.hop; st.pad_left = c.n_dft - 1; st.n_dft = c.n_dft;
    st.NT = m->dm.NT_dft; st.LP = m->dm.LP; st.nbins = m->dm.nbins; st.FT = FT; st.NCH = m->dm.NCH_dft;
    st.db10 = 0;
    PROF(MI355ASR_K_STFT);
    LAUNCH_TRY(launch_stft(st, s), "stft (valid)");
  }
  MelArgs me{};
  me.logp = logp; me.umax = nullptr; me.mel = mel; me.wp = m->mel_wp;
  me.B = 1; me.F = F; me.LP = m->dm.LP; me.nbins = m->dm.nbins; me.KBm = m->dm.KBm; me.NTm = m->dm.NTm;
  me.NM = c.n_mels; me.FT = FT; me.floor_db = 0.f;
  { PROF(MI355ASR_K_MEL); LAUNCH_TRY(launch_mel_auto(m, me, s), "mel (valid)"); }
  // new_sub = [sub cache ; last nf mel frames]  (ConvSubsampling.stream_call :75)
  const size_t mrow = (size_t)c.n_mels * 4;
  if (S > 0) HIP_TRY(hipMemcpyAsync(new_sub, sub_cache, S * mrow, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(new_sub + (size_t)S * c.n_mels, mel + (size_t)(F - nf) * c.n_mels, nf * mrow,
                         hipMemcpyDeviceToDevice, s));
  if (Tout < 1) return 0;
  // pad 2/2 on the mel axis only, two VALID 3x3 stride-2 convs, last Tout frames, Dense (:77-89)
  SubConvArgs sa{};
  sa.mel = new_sub; sa.out = (float*)(ws + p.sub); sa.w1 = m->c1_w; sa.b1 = m->c1_b; sa.w2p = m->c2_wp; sa.b2 = m->c2_b;
  sa.B = 1; sa.F = S + nf; sa.NM = c.n_mels; sa.T1 = T1; sa.F1 = m->dm.F1; sa.T2 = T2; sa.F2 = m->dm.F2;
  sa.st1 = 2; sa.pt1 = 0; sa.pf1 = 2; sa.pt2 = 0; sa.pf2 = 0;
  { PROF(MI355ASR_K_SUBCONV); LAUNCH_TRY(launch_subconv(d, sa, s), "conv subsampling (stream)"); }
  StreamGemmArgs lg{};
  lg.x = sa.out + (size_t)(T2 - Tout) * m->dm.F2 * d; lg.y = front_out; lg.wp = m->lin_wp; lg.bias = m->lin_b;
  lg.M = Tout; lg.K = m->dm.F2 * d; lg.NT = d / 16; lg.ldy = d; lg.n_valid = d;
  { PROF(MI355ASR_K_SUBLINEAR); LAUNCH_TRY(launch_stream_gemm(d, lg, s), "subsampling linear"); }
  return 0;
}

int mi355asr_chunk_stack_stream(mi355asr_model* m, int32_t stack, const float* x, int32_t T, const float* mha_cache,
                                int32_t Cm, const float* cnn_cache, int32_t Cc, float* hidden, float* logits,
                                int32_t* amax, float* new_mha, float* new_cnn, void* ws_, size_t ws_bytes,
                                void* stream) {
  if (!m || !m->is_chunk) return fail(MI355ASR_EINVAL, "not a ChunkConformer handle");
  if (!m->finalized) return fail(MI355ASR_ESTATE, "weights not finalised: call mi355asr_finalize_weights first");
  const StackDev* st = chunk_stack_by_id(m, stack);
  if (!st) return fail(MI355ASR_EINVAL, "stack=%d: 0 encoder, 1 picker, 2 helper, 3 decoder", stack);
  if (!x || !hidden || !new_mha || !new_cnn || !ws_ || (Cm > 0 && !mha_cache) || (Cc > 0 && !cnn_cache))
    return fail(MI355ASR_EINVAL, "null argument");
  if (T < 1 || Cm < 0 || Cc < 0) return fail(MI355ASR_EINVAL, "need T >= 1, Cm >= 0, Cc >= 0 (got %d, %d, %d)", T, Cm, Cc);
  if ((logits || amax) && !st->fc_wp) return fail(MI355ASR_EINVAL, "stack %d has no fully_connected head", stack);
  const int d = m->cfg.dmodel;
  const int N = std::max(Cm, Cc) + T;
  const StreamPlan p = make_stream_plan(m, N, 1, 1);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  char* ws = (char*)ws_;
  hipStream_t s = (hipStream_t)stream;
  Scratch sc{(float*)(ws + p.xa), (float*)(ws + p.xb), (float*)(ws + p.qkv),
             (float*)(ws + p.ctx), (float*)(ws + p.u), (float*)(ws + p.dw)};
  if (st->proj_wp) {
    GemmArgs pr{};
    pr.x = x; pr.y = sc.xa; pr.wp = st->proj_wp; pr.bias = st->proj_b;
    pr.M = T; pr.NT = d / 16; pr.ldy = d; pr.n_valid = d; pr.eps = kLnEps;
    { PROF(MI355ASR_K_CTC_PROJECT); LAUNCH_TRY(launch_gemm_rows(d, EPI_BIAS, false, pr, s), "project"); }
  } else {
    HIP_TRY(hipMemcpyAsync(sc.xa, x, (size_t)T * d * 4, hipMemcpyDeviceToDevice, s));
  }
  const size_t nb = st->blocks.size();
  for (size_t i = 0; i < nb; ++i) {
    int rc = run_block_stream(m, st->blocks[i], st->opts, sc, T, mha_cache ? mha_cache + i * (size_t)Cm * d : nullptr, Cm,
                              cnn_cache ? cnn_cache + i * (size_t)Cc * d : nullptr, Cc,
                              new_mha + i * (size_t)(Cm + T) * d, new_cnn + i * (size_t)(Cc + T) * d, s);
    if (rc) return rc;
  }
  HIP_TRY(hipMemcpyAsync(hidden, sc.xa, (size_t)T * d * 4, hipMemcpyDeviceToDevice, s));
  if (st->fc_wp && (logits || amax)) {
    GemmArgs hd{};
    hd.x = sc.xa; hd.y = logits; hd.wp = st->fc_wp; hd.bias = st->fc_b;
    hd.M = T; hd.NT = st->NT_fc; hd.ldy = st->num_classes; hd.n_valid = st->num_classes; hd.eps = kLnEps;
    hd.argmax_out = amax ? amax : (int32_t*)(ws + p.amax);
    { PROF(MI355ASR_K_CTC_HEAD); LAUNCH_TRY(launch_gemm_rows(d, EPI_HEAD, false, hd, s), "fully_connected"); }
  }
  return 0;
}
}  // extern "C"
