// Translator host side (conformer_blocks.py:439-566) and the stateful beam decoder handle, behind
// mi355asr_translator_* / mi355asr_beam_* of include/mi355asr.h.
#include "model.h"

namespace {
// =======================================================================================================
// Translator (conformer_blocks.py:505-548)
// =======================================================================================================
constexpr int kMaxTokens = 2048;   // rows of the positional-encoding table

struct TransPlan {
  size_t xa, xb, qkv, ctx, u, dw, kv, amax, h4, hsplit, total;
};
TransPlan make_trans_plan(const mi355asr_model* m, int B, int U, int T) {
  const size_t d = m->cfg.dmodel, M = (size_t)B * U;
  TransPlan p;
  size_t o = 0;
  auto take = [&](size_t floats) { size_t at = o; o = align256(o + floats * 4); return at; };
  p.xa = take(M * d); p.xb = take(M * d); p.qkv = take(M * 3 * d); p.ctx = take(M * d);
  p.u = take(M * d); p.dw = take(M * d); p.kv = take((size_t)B * T * 2 * d); p.amax = take(M);
  p.h4 = gemm16_for(m, M) ? take(M * 4 * d) : 0;
  p.hsplit = take(16 * M);          // per-range (maximum, class) pairs of the class head split over column ranges (up to 8)
  p.total = o;
  return p;
}

}  // namespace

namespace mi355 {

int finalize_translator(mi355asr_model* m, hipStream_t s) {
  const auto& c = m->cfg;
  const auto& tc = m->tcfg;
  const int d = c.dmodel, H = c.num_heads, hs = c.head_size, V = tc.tar_classes;
  ArenaBuilder ab;
  ab.ring_terms = m->cfg.gemm_dtype == 1 ? 1 : 3;
  const size_t o_emb = ab.put(m->host["inp_embedding/embeddings"].data);
  // positional_encoding.py:19-36: pe[pos, 2i] = sin(pos / 10000^(2i/d)), pe[pos, 2i+1] = cos(pos / 10000^(2i/d))
  std::vector<float> pe((size_t)kMaxTokens * d);
  for (int pos = 0; pos < kMaxTokens; ++pos)
    for (int i = 0; i < d; ++i) {
      const double ang = (double)pos / std::pow(10000.0, (double)(2 * (i / 2)) / d);
      pe[(size_t)pos * d + i] = (float)((i & 1) ? std::cos(ang) : std::sin(ang));
    }
  const size_t o_pe = ab.put(pe);
  StackOff so;
  for (int i = 0; i < tc.num_blocks; ++i) {
    const std::string p = "decoder_conformer_block_" + std::to_string(i);
    BlockOff o = pack_block(m, ab, p, d, H, hs, c.kernel_size);
    const auto& qk = m->host[p + "/mhsa_module/mha/query_kernel"].data;   // [H, d, hs]
    const auto& kk = m->host[p + "/mhsa_module/mha/key_kernel"].data;
    const auto& vk = m->host[p + "/mhsa_module/mha/value_kernel"].data;
    o.cross = true;
    o.xq_wp = ab.put(pack_p16([&](int i2, int n) { return qk[((size_t)(n / hs) * d + i2) * hs + n % hs]; }, d, d, d / 16));
    o.xkv_wp = ab.put(pack_p16(
        [&](int i2, int n) {
          const int r = n % d;
          const std::vector<float>& w = n < d ? kk : vk;
          return w[((size_t)(r / hs) * d + i2) * hs + r % hs];
        },
        d, 2 * d, 2 * d / 16));
    so.blocks.push_back(o);
  }
  const auto& fc = m->host["fully_connected/kernel"].data;
  const int ct = gemm_ct(d, EPI_HEAD);
  so.NT_fc = ceil_div(ceil_div(V, 16), ct) * ct;
  so.fc_w = ab.put(pack_p16([&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V, so.NT_fc));
  if (ring_packs_wanted(m)) put_ring_head(ab, so.fc_w, [&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V);
  put_head_slabs(ab, so.fc_w, [&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V, m->host["fully_connected/bias"].data.data());
  so.fc_b = ab.put_padded(m->host["fully_connected/bias"].data.data(), V, (size_t)so.NT_fc * 16);
  if (m->arena) { (void)hipFree(m->arena); m->arena = nullptr; }
  HIP_TRY(hipMalloc((void**)&m->arena, ab.buf.size() * sizeof(float)));
  m->arena_floats = ab.buf.size();
  HIP_TRY(hipMemcpyAsync(m->arena, ab.buf.data(), ab.buf.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  const float* base = m->arena;
  m->ring_of.clear();
  register_rings(m, ab, base);
  m->t_emb = base + o_emb;
  m->t_pe = base + o_pe;
  resolve_stack(m->t_stack, so, base, false, V);
  m->t_stack.opts.ksz = c.kernel_size;
  m->t_stack.opts.fc = c.fc_factor;
  for (auto& kv : m->host) { kv.second.data.clear(); kv.second.data.shrink_to_fit(); }
  m->finalized = true;
  return 0;
}

}  // namespace mi355

extern "C" {
// ---- Translator ----------------------------------------------------------------------------------------
int mi355asr_translator_create(const mi355asr_translator_config* cfg, mi355asr_model** out) {
  if (!cfg || !out) return fail(MI355ASR_EINVAL, "null argument");
  const auto& c = *cfg;
  if (c.dmodel != 144 && (c.dmodel % 128 != 0 || c.dmodel < 128 || c.dmodel > 1024))
    return fail(MI355ASR_EINVAL, "Translator: dmodel=%d, supported are 144 and multiples of 128 up to 1024", c.dmodel);
  if (c.num_heads * c.head_size != c.dmodel || !attention_head_size_ok(c.head_size))
    return fail(MI355ASR_EINVAL, "Translator: need num_heads*head_size == dmodel and a head size of 12, 16, 24, 32, 36, 48, 64, 72 or 128");
  if (c.kernel_size < 1 || c.kernel_size > 1024) return fail(MI355ASR_EINVAL, "kernel_size=%d: must be in 1 .. 1024", c.kernel_size);
  if (c.num_blocks < 1 || c.inp_classes < 1 || c.tar_classes < 2) return fail(MI355ASR_EINVAL, "Translator: bad block / class counts");
  auto* m = new mi355asr_model();
  m->is_translator = true;
  m->tcfg = c;
  std::memset(&m->cfg, 0, sizeof(m->cfg));
  m->cfg.dmodel = c.dmodel; m->cfg.head_size = c.head_size; m->cfg.num_heads = c.num_heads;
  m->cfg.kernel_size = c.kernel_size; m->cfg.fc_factor = c.fc_factor;
  std::memset(&m->dm, 0, sizeof(m->dm));
  const int d = c.dmodel;
  auto& ex = m->expected;
  ex.push_back({"inp_embedding/embeddings", {c.inp_classes, d}});
  for (int i = 0; i < c.num_blocks; ++i)
    add_block_expected(ex, "decoder_conformer_block_" + std::to_string(i), d, c.num_heads, c.head_size, c.kernel_size);
  ex.push_back({"fully_connected/kernel", {d, c.tar_classes}});
  ex.push_back({"fully_connected/bias", {c.tar_classes}});
  for (const auto& e : ex) m->host[e.name] = HostTensor{};
  *out = m;
  return 0;
}

int mi355asr_translator_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t U, int32_t T, size_t* bytes) {
  if (!m || !m->is_translator || !bytes) return fail(MI355ASR_EINVAL, "not a Translator handle / null argument");
  if (B < 1 || U < 1 || T < 1) return fail(MI355ASR_EINVAL, "B, U, T must be positive (got %d, %d, %d)", B, U, T);
  *bytes = make_trans_plan(m, B, U, T).total;
  return 0;
}

int mi355asr_translator_forward(mi355asr_model* m, const int32_t* ids, const float* enc, int32_t B, int32_t U,
                                int32_t T, float* logits, int32_t* amax, void* ws_, size_t ws_bytes, void* stream) {
  if (!m || !m->is_translator) return fail(MI355ASR_EINVAL, "not a Translator handle");
  if (!m->finalized) return fail(MI355ASR_ESTATE, "weights not finalised: call mi355asr_finalize_weights first");
  if (!ids || !enc || !ws_) return fail(MI355ASR_EINVAL, "null argument");
  if (B < 1 || U < 1 || T < 1) return fail(MI355ASR_EINVAL, "B, U, T must be positive (got %d, %d, %d)", B, U, T);
  if (U > kMaxTokens) return fail(MI355ASR_EINVAL, "U=%d exceeds the positional-encoding table (%d rows)", U, kMaxTokens);
  const TransPlan p = make_trans_plan(m, B, U, T);
  if (ws_bytes < p.total) return fail(MI355ASR_EINVAL, "workspace too small: %zu < %zu", ws_bytes, p.total);
  char* ws = (char*)ws_;
  hipStream_t s = (hipStream_t)stream;
  const int d = m->cfg.dmodel, M = B * U;
  Scratch sc{(float*)(ws + p.xa), (float*)(ws + p.xb), (float*)(ws + p.qkv),
             (float*)(ws + p.ctx), (float*)(ws + p.u), (float*)(ws + p.dw)};
  sc.h4 = (float*)(ws + p.h4);
  EmbedArgs ea{ids, m->t_emb, sc.xa, M, m->tcfg.inp_classes, d};
  LAUNCH_TRY(launch_embed(ea, s), "embedding");
  CrossAttn cr{enc, T, (float*)(ws + p.kv), m->t_pe};
  for (const auto& blk : m->t_stack.blocks) {
    int rc = run_block(m, blk, m->t_stack.opts, sc, B, U, nullptr, s, &cr);
    if (rc) return rc;
  }
  GemmArgs hd{};
  hd.x = sc.xa; hd.y = logits; hd.wp = m->t_stack.fc_wp; hd.bias = m->t_stack.fc_b;
  hd.M = M; hd.NT = m->t_stack.NT_fc; hd.ldy = m->tcfg.tar_classes; hd.n_valid = m->tcfg.tar_classes; hd.eps = kLnEps;
  hd.argmax_out = amax ? amax : (int32_t*)(ws + p.amax);
  // round 5: from 2048 rows on the class head runs on the two-term stream of pp_head_kernel (or the slab ring), as the CTC decoder's
  // and the ChunkConformer's heads do (try_head_ld: 144 -> 9160 over 5952 rows 0.44 ms on the fp32 MFMA kernel)
  {
    PROF(MI355ASR_K_CTC_HEAD);
    if (try_head_ld(m, hd, s, (float*)(ws + p.hsplit)) == 0) {
      if (hipGetLastError() != hipSuccess) return fail(MI355ASR_EHIP, "translator head launch failed");
      return 0;
    }
  }
  if (gemm16_for(m, M)) {
    Gemm16Args h16{};
    h16.x = sc.xa; h16.ldx = d; h16.bias = hd.bias; h16.y = logits; h16.ldy = hd.ldy; h16.M = M; h16.K = d; h16.NT = hd.NT;
    h16.n_valid = hd.n_valid; h16.eps = kLnEps; h16.argmax_out = hd.argmax_out;
    h16.part_max = 8; h16.part_v = (float*)(ws + p.hsplit); h16.part_i = reinterpret_cast<int32_t*>(h16.part_v + (size_t)8 * M);
    { PROF(MI355ASR_K_CTC_HEAD); LAUNCH_TRY(launch_gemm16(m, E16_HEAD, false, h16, m->t_stack.fc_wp, s), "translator head"); }
    return 0;
  }
  { PROF(MI355ASR_K_CTC_HEAD); LAUNCH_TRY(launch_gemm_rows(d, EPI_HEAD, false, hd, s), "translator head"); }
  return 0;
}

// ---- stateful BeamDecoder ----------------------------------------------------------------------------------
struct mi355asr_beam { void* st; int V, beam; };
int mi355asr_beam_create(int32_t V, int32_t beam_size, double cutoff_prob, int32_t cutoff_top_n, mi355asr_beam** out) {
  if (!out) return fail(MI355ASR_EINVAL, "null argument");
  if (V < 2 || beam_size < 1 || cutoff_top_n < 1 || !(cutoff_prob > 0.0) || cutoff_prob > 1.0)
    return fail(MI355ASR_EINVAL, "beam decoder: need V >= 2, beam_size >= 1, cutoff_top_n >= 1, 0 < cutoff_prob <= 1");
  auto* d = new mi355asr_beam{mi355asr_beam_state_new(V, beam_size, cutoff_prob, cutoff_top_n), V, beam_size};
  *out = d;
  return 0;
}
int mi355asr_beam_decode(mi355asr_beam* d, const float* probs, int32_t T, int32_t max_len, int32_t* ids, int32_t* lens,
                         float* scores, int32_t* n_hyp) {
  if (!d || !ids || !lens || !scores || !n_hyp || (T > 0 && !probs)) return fail(MI355ASR_EINVAL, "null argument");
  if (T < 0 || max_len < 1) return fail(MI355ASR_EINVAL, "T must be >= 0 and max_len >= 1 (got %d, %d)", T, max_len);
  *n_hyp = mi355asr_beam_state_decode(d->st, probs, T, max_len, ids, lens, scores);
  return 0;
}
int mi355asr_beam_reset(mi355asr_beam* d) {
  if (!d) return fail(MI355ASR_EINVAL, "null argument");
  mi355asr_beam_state_reset(d->st);
  return 0;
}
int mi355asr_beam_destroy(mi355asr_beam* d) {
  if (!d) return 0;
  mi355asr_beam_state_free(d->st);
  delete d;
  return 0;
}
}  // extern "C"
