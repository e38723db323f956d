// libmi355asr.so host side: model object, Keras-layout weight intake + packing into MFMA fragment order,
// workspace planning and the launch sequences behind the C ABI declared in include/mi355asr.h.
#include "model.h"

namespace mi355 {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

void same_pad(int n, int k, int s, int* out, int* before) {
  const int o = ceil_div(n, s);
  const int tot = std::max((o - 1) * s + k - n, 0);
  *out = o;
  *before = tot / 2;
}

void add_block_expected(std::vector<Expected>& ex, const std::string& p, int d, int H, int hs, int k,
                        bool keras_mha) {
  auto ln = [&](const std::string& q) {
    ex.push_back({q + "/gamma", {d}});
    ex.push_back({q + "/beta", {d}});
  };
  for (const char* ff : {"ff_module_1", "ff_module_2"}) {
    const std::string q = p + "/" + ff;
    ln(q + "/ln");
    ex.push_back({q + "/ffn1/kernel", {d, 4 * d}});
    ex.push_back({q + "/ffn1/bias", {4 * d}});
    ex.push_back({q + "/ffn2/kernel", {4 * d, d}});
    ex.push_back({q + "/ffn2/bias", {d}});
  }
  const std::string m = p + "/mhsa_module";
  ln(m + "/ln");
  if (keras_mha) {   // tf.keras.layers.MultiHeadAttention (chunk_conformer_blocks.py:147): biased q/k/v/out
    for (const char* w : {"query", "key", "value"}) {
      ex.push_back({m + "/mha/" + w + "/kernel", {d, H, hs}});
      ex.push_back({m + "/mha/" + w + "/bias", {H, hs}});
    }
    ex.push_back({m + "/mha/attention_output/kernel", {H, hs, d}});
    ex.push_back({m + "/mha/attention_output/bias", {d}});
  } else {
    ex.push_back({m + "/mha/query_kernel", {H, d, hs}});
    ex.push_back({m + "/mha/key_kernel", {H, d, hs}});
    ex.push_back({m + "/mha/value_kernel", {H, d, hs}});
    ex.push_back({m + "/mha/projection_kernel", {H, hs, d}});
    ex.push_back({m + "/mha/projection_bias", {d}});
  }
  const std::string c = p + "/conv_module";
  ln(c + "/ln");
  ex.push_back({c + "/pw_conv_1/kernel", {1, d, 2 * d}});
  ex.push_back({c + "/pw_conv_1/bias", {2 * d}});
  ex.push_back({c + "/dw_conv/depthwise_kernel", {k, d, 1}});
  ex.push_back({c + "/dw_conv/pointwise_kernel", {1, d, 2 * d}});
  ex.push_back({c + "/dw_conv/bias", {2 * d}});
  ex.push_back({c + "/bn/gamma", {2 * d}});
  ex.push_back({c + "/bn/beta", {2 * d}});
  ex.push_back({c + "/bn/moving_mean", {2 * d}});
  ex.push_back({c + "/bn/moving_variance", {2 * d}});
  ex.push_back({c + "/pw_conv_2/kernel", {1, 2 * d, d}});
  ex.push_back({c + "/pw_conv_2/bias", {d}});
  ln(p + "/ln");
}

// W[k][n] (k < K, n < N) -> P16 fragment order [ceil(K/16)][NTpad][64 lanes][4]
std::vector<float> pack_p16(const std::function<float(int, int)>& f, int K, int N, int NTpad) {
  const int KBT = ceil_div(K, 16);
  std::vector<float> out((size_t)KBT * NTpad * 256, 0.f);
  for (int kb = 0; kb < KBT; ++kb)
    for (int nt = 0; nt < NTpad; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, c = lane & 15;
        for (int j = 0; j < 4; ++j) {
          const int k = 16 * kb + 4 * g + j, n = 16 * nt + c;
          if (k < K && n < N) out[(((size_t)kb * NTpad + nt) * 64 + lane) * 4 + j] = f(k, n);
        }
      }
  return out;
}


// W[k][n] -> split-bf16 fragments for v_mfma_f32_16x16x32_bf16: [ceil(K/32) steps][N/16 tiles][3 terms][64 lanes][8],
// lane (r = lane & 15, g = lane >> 4) of tile nt holds, for column 16 nt + r, rows k = 32 step + 16 (j >> 2) + 4 g + (j & 3)
// (the two 16-blocks of a step side by side, as a lane's accumulator-layout float4 pair provides them); zero past K.
// Term t = round-to-nearest-even bf16 of what the terms before it left (three terms hold all 24 significand bits).
std::vector<float> pack_split32(const std::function<float(int, int)>& f, int K, int N) {
  const int steps = ceil_div(K, 32), NT = N / 16;
  std::vector<uint16_t> frag((size_t)steps * NT * 3 * 64 * 8, 0);
  auto rne = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
  for (int st = 0; st < steps; ++st)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * st + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3), n = 16 * nt + (lane & 15);
          float r = k < K ? f(k, n) : 0.f;
          for (int t = 0; t < 3; ++t) {
            const uint16_t hb = rne(r);
            const uint32_t back = (uint32_t)hb << 16;
            float hf; std::memcpy(&hf, &back, 4);
            r -= hf;
            frag[((((size_t)st * NT + nt) * 3 + t) * 64 + lane) * 8 + j] = hb;
          }
        }
  std::vector<float> as_f(frag.size() / 2);
  std::memcpy(as_f.data(), frag.data(), frag.size() * 2);
  return as_f;
}


// Appends the slabs of W (pack_split32 order) to a slab stream: one slab = 9 column tiles of one 32-wide step = 1728
// fragments of 16 bytes, padded to 1792 (SlabStream in fused.hip).  group_major: all steps of tile group 0, then of
// group 1, ... (the order a GEMM swept in column chunks consumes them); else step by step, its groups side by side.
void append_slabs(std::vector<float>& stream, const std::function<float(int, int)>& f, int K, int N, bool group_major) {
  const std::vector<float> sp = pack_split32(f, K, N);
  const int steps = ceil_div(K, 32), NT = N / 16, groups = NT / 9;
  const size_t used = 1728 * 4, stride = 1792 * 4;
  auto put = [&](int st, int gr) {
    const size_t at = stream.size();
    stream.resize(at + stride, 0.f);
    std::memcpy(stream.data() + at, sp.data() + ((size_t)st * NT + 9 * gr) * 192 * 4, used * sizeof(float));
  };
  if (group_major) { for (int gr = 0; gr < groups; ++gr) for (int st = 0; st < steps; ++st) put(st, gr); }
  else { for (int st = 0; st < steps; ++st) for (int gr = 0; gr < groups; ++gr) put(st, gr); }
}


// conv2 kernel [3][3][d][d] (HWIO) as split-bf16 fragments for subconv_split_ring_kernel: column chunks of NTc tiles (all
// nine at dmodel 144, eight otherwise).  Steps (subconv.hip): s < 4 KB: channel block cb = s / 4, tap pair p = s % 4 -- lane
// (r = lane & 15, g = lane >> 4) of column tile nt holds, for out channel 16 nt + r, in-channels 16 cb + 4 g + (j & 3) at tap
// 2 p + (j >> 2); then ceil(KB / 2) steps with the NINTH tap of two channel blocks: j < 4 -> block 2 i, j >= 4 -> block
// 2 i + 1 (zero past the last block).  Term t = round-to-nearest-even bf16 of what the terms before it left.
std::vector<float> pack_conv2_split(const std::vector<float>& c2, int d) {
  const int KBn = d / 16, steps = KBn * 4 + (KBn + 1) / 2, NTc = d == 144 ? 9 : 8, chunks = KBn / NTc;
  std::vector<uint16_t> frag((size_t)chunks * steps * NTc * 3 * 64 * 8);
  auto rne = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
  for (int ch = 0; ch < chunks; ++ch)
    for (int st = 0; st < steps; ++st)
      for (int nt = 0; nt < NTc; ++nt)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            int cb, q;
            if (st < KBn * 4) { cb = st / 4; q = 2 * (st % 4) + (j >> 2); }
            else { cb = 2 * (st - KBn * 4) + (j >> 2); q = 8; }
            const int cin = 16 * cb + 4 * (lane >> 4) + (j & 3), cout = 16 * (ch * NTc + nt) + (lane & 15);
            float r = cb < KBn ? c2[((size_t)q * d + cin) * d + cout] : 0.f;
            for (int t = 0; t < 3; ++t) {
              const uint16_t hb = rne(r);
              const uint32_t back = (uint32_t)hb << 16;
              float hf; std::memcpy(&hf, &back, 4);
              r -= hf;
              frag[(((((size_t)ch * steps + st) * NTc + nt) * 3 + t) * 64 + lane) * 8 + j] = hb;
            }
          }
  std::vector<float> as_f(frag.size() / 2);
  std::memcpy(as_f.data(), frag.data(), frag.size() * 2);
  return as_f;
}
// round-to-nearest-even fp16 of a float (host side of the two-term scheme; subnormals and overflow to infinity included)
uint16_t f16_rne(float v) {
  uint32_t u; std::memcpy(&u, &v, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // rounds to >= 65520: infinity
  if (u < 0x38800000u) {                                             // below 2^-14: subnormal, spacing 2^-24
    float f; std::memcpy(&f, &u, 4);
    const float r = f * 16777216.0f;                                  // exact
    const float q = std::nearbyintf(r);                               // ties to even
    return (uint16_t)(sign | (uint32_t)q);                            // q == 1024 is the smallest normal
  }
  const uint32_t mant = u & 0x7fffffu, exp = (u >> 23) - 112;         // 1 .. 30
  uint32_t h = (exp << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;             // carries into the exponent as it should
  return (uint16_t)(sign | h);
}
float f16_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, mnt = h & 1023u;
  float f;
  if (e == 0) f = (float)mnt * 5.9604644775390625e-8f;                // 2^-24
  else if (e == 31) f = mnt ? NAN : INFINITY;
  else { const uint32_t u = ((e + 112) << 23) | (mnt << 13); std::memcpy(&f, &u, 4); }
  uint32_t u; std::memcpy(&u, &f, 4); u |= sign; std::memcpy(&f, &u, 4);
  return f;
}
// largest power of two s with bound * s <= 2^15 (fp16's largest finite value is 65504: a factor of two to spare)
// max_shift: kernels that multiply two or three such scales in fp32 (attention: sq * sk, probabilities * sv) pass 40 and get
// 0 = "no usable bound" (the three-term kernel runs) for degenerate weights instead of a product that overflows to inf
float half_scale_for(double bound, int max_shift) {
  if (!(bound > 0.0) || !std::isfinite(bound)) return 0.f;
  int e; std::frexp(bound, &e);                                       // bound = f 2^e, f in [0.5, 1)
  const int k = 15 - e;
  if (k > max_shift || k < -max_shift) return max_shift < 100 ? 0.f : std::ldexp(1.0f, std::max(-100, std::min(100, k)));
  return std::ldexp(1.0f, k);
}
// pack_conv2_split's fragment order with TWO fp16 terms of kernel * wscale (subconv.hip, two-term scheme)
std::vector<float> pack_conv2_half(const std::vector<float>& c2, int d, float wscale) {
  const int KBn = d / 16, steps = KBn * 4 + (KBn + 1) / 2, NTc = d == 144 ? 9 : 8, chunks = KBn / NTc;
  std::vector<uint16_t> frag((size_t)chunks * steps * NTc * 2 * 64 * 8);
  for (int ch = 0; ch < chunks; ++ch)
    for (int st = 0; st < steps; ++st)
      for (int nt = 0; nt < NTc; ++nt)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            int cb, q;
            if (st < KBn * 4) { cb = st / 4; q = 2 * (st % 4) + (j >> 2); }
            else { cb = 2 * (st - KBn * 4) + (j >> 2); q = 8; }
            const int cin = 16 * cb + 4 * (lane >> 4) + (j & 3), cout = 16 * (ch * NTc + nt) + (lane & 15);
            const float v = (cb < KBn ? c2[((size_t)q * d + cin) * d + cout] : 0.f) * wscale;
            const uint16_t hi = f16_rne(v), lo = f16_rne(v - f16_to_float(hi));
            const size_t at = ((((size_t)ch * steps + st) * NTc + nt) * 2 * 64 + lane) * 8 + j;
            frag[at] = hi;
            frag[at + 64 * 8] = lo;
          }
  std::vector<float> as_f(frag.size() / 2);
  std::memcpy(as_f.data(), frag.data(), frag.size() * 2);
  return as_f;
}
// the subsampling Dense [K, 144] for sublinear_split_kernel: 1728 fragments per 32-wide step, padded to 7 x 256 (4 floats each)
std::vector<float> pack_linear_split(const std::vector<float>& lin, int K, int d) {
  const std::vector<float> sp = pack_split32([&](int k, int n) { return lin[(size_t)k * d + n]; }, K, d);
  const size_t steps = (size_t)ceil_div(K, 32), used = 1728 * 4, stride = 1792 * 4;
  std::vector<float> padded(steps * stride, 0.f);
  for (size_t st = 0; st < steps; ++st) std::memcpy(padded.data() + st * stride, sp.data() + st * used, used * sizeof(float));
  return padded;
}

// ---- pair-pipelined streams (fused_pp.hip; layout tables generated by tools/gen_pp.py) -----------------------------------
#include "pp_layout.inc"
namespace {
// one ring slot: kPpSlot fragments of 1 KB (256 floats) in the order of `lay`; src(desc) = the fragment's 256 floats
void put_pp_slot(std::vector<float>& stream, const PpFragDesc (&lay)[kPpSlot], const std::function<const float*(const PpFragDesc&)>& src) {
  const size_t at = stream.size();
  stream.resize(at + (size_t)kPpSlot * 256, 0.f);
  for (int i = 0; i < kPpSlot; ++i)
    if (lay[i].kind != 0) std::memcpy(stream.data() + at + (size_t)i * 256, src(lay[i]), 256 * sizeof(float));
}
}  // namespace
// pack_split32's fragment order with TWO fp16 terms of f * scale: [steps][NT][2 terms][64 lanes][8] (two-term scheme)
std::vector<float> pack_half32(const std::function<float(int, int)>& f, int K, int N, float scale) {
  const int steps = ceil_div(K, 32), NT = N / 16;
  std::vector<uint16_t> frag((size_t)steps * NT * 2 * 64 * 8, 0);
  for (int st = 0; st < steps; ++st)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * st + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3), n = 16 * nt + (lane & 15);
          const float v = (k < K ? f(k, n) : 0.f) * scale;
          const uint16_t hi = f16_rne(v), lo = f16_rne(v - f16_to_float(hi));
          const size_t at = ((((size_t)st * NT + nt) * 2) * 64 + lane) * 8 + j;
          frag[at] = hi;
          frag[at + 64 * 8] = lo;
        }
  std::vector<float> as_f(frag.size() / 2);
  std::memcpy(as_f.data(), frag.data(), frag.size() * 2);
  return as_f;
}
namespace {
float matrix_scale(const std::function<float(int, int)>& f, int K, int N) {   // power of two: max |f| * s in [2^14, 2^15)
  double mx = 0.0;
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) mx = std::max(mx, std::fabs((double)f(k, n)));
  const float s = half_scale_for(mx);
  return s > 0.f ? s : 1.0f;                  // an all-zero matrix
}
}  // namespace
// Chain y += W2 act(W1aug [x ; 1]) over P = H / 32 hidden pairs, units A, AP, (P - 2) x F, BP, B (2 P ring slots):
// an A fragment = W1aug step a, hidden tile 2 pair + b; a B fragment = W2 step `pair`, column tile a.  Returns the scales the
// two matrices were packed with and the bounds the kernel derives the operand scales from.
PpChainSc append_pp_chain(std::vector<float>& stream, const std::function<float(int, int)>& w1aug, int H, const std::function<float(int, int)>& w2,
                          std::vector<float>* plain1, std::vector<float>* plain2) {
  const int P = H / 32, NT1 = H / 16;
  PpChainSc sc;
  sc.sw1 = matrix_scale(w1aug, 145, H);
  sc.sw2 = matrix_scale(w2, H, 144);
  double l1 = 0.0, bm = 0.0;
  for (int n = 0; n < H; ++n) {
    double sum = 0.0;
    for (int k = 0; k < 144; ++k) sum += std::fabs((double)w1aug(k, n));
    l1 = std::max(l1, sum);
    bm = std::max(bm, std::fabs((double)w1aug(144, n)));
  }
  sc.l1 = (float)(l1 * (1.0 + 1e-6));         // rounded up: the bound has to hold in float
  sc.bmax = (float)(bm * (1.0 + 1e-6));
  const std::vector<float> sp1 = pack_half32(w1aug, 145, H, sc.sw1);      // [5 steps][NT1][2 terms][256]
  const std::vector<float> sp2 = pack_half32(w2, H, 144, sc.sw2);         // [P steps][9][2][256]
  auto fa = [&](int pair, const PpFragDesc& d) { return sp1.data() + (((size_t)d.a * NT1 + 2 * pair + d.b) * 2 + d.term) * 256; };
  auto fb = [&](int pair, const PpFragDesc& d) { return sp2.data() + (((size_t)pair * 9 + d.a) * 2 + d.term) * 256; };
  auto unit = [&](const PpFragDesc (&lay)[kPpSlot], int pa, int pb) {
    put_pp_slot(stream, lay, [&](const PpFragDesc& d) { return d.kind == 1 ? fa(pa, d) : fb(pb, d); });
  };
  unit(kPpLayout_A0, 0, -1);
  unit(kPpLayout_AP0, 1, -1);
  for (int p = 0; p + 2 < P; ++p) { unit(kPpLayout_F0, p + 2, p); unit(kPpLayout_F1, p + 2, p); }
  unit(kPpLayout_BP0, -1, P - 2);
  unit(kPpLayout_B0, -1, P - 1);
  if (plain1) *plain1 = sp1;                  // the same fragments in plain [step][tile][term] order (fused_ns.hip)
  if (plain2) *plain2 = sp2;
  return sc;
}
// A plain layer [145 (row 144 = bias), 144 * groups] in column groups of nine tiles, five S units (ring slots) per group;
// returns the power of two the matrix was packed with
float append_pp_plain(std::vector<float>& stream, const std::function<float(int, int)>& waug, int groups, std::vector<float>* plain) {
  const int NT = 9 * groups;
  const float sw = matrix_scale(waug, 145, 144 * groups);
  const std::vector<float> sp = pack_half32(waug, 145, 144 * groups, sw);
  for (int g = 0; g < groups; ++g)
    for (int st = 0; st < 5; ++st)
      put_pp_slot(stream, kPpLayout_S0, [&](const PpFragDesc& d) { return sp.data() + (((size_t)st * NT + 9 * g + d.a) * 2 + d.term) * 256; });
  if (plain) *plain = sp;
  return sw;
}


// The DFT kernels are model variables (time_frequency.py:62-75 creates them from backend.py:27-69 and a checkpoint
// may overwrite them).  When they are exactly window[n] * (cos, -+sin)(2 pi k n / 1024) the STFT runs as a
// 32 x 32 Cooley-Tukey factorisation (fft_stft.hip); otherwise the dense DFT GEMM stays.  MI355ASR_FFT=0 forces dense.
FftOff pack_fft(ArenaBuilder& ab, const std::vector<float>& re, const std::vector<float>& im, int n_dft, int nb) {
  FftOff o;
  if (mi355_env("MI355ASR_FFT", 1) == 0) return o;
  if (n_dft != 1024 || nb != 513) return o;
  const double two_pi = 6.283185307179586476925286766559;
  std::vector<double> ct(1024), st(1024);
  for (int i = 0; i < 1024; ++i) { ct[i] = std::cos(two_pi * i / 1024.0); st[i] = std::sin(two_pi * i / 1024.0); }
  std::vector<float> win(1024);
  for (int n = 0; n < 1024; ++n) win[n] = re[(size_t)n * nb];   // bin 0: cos = 1
  const double tol = 1e-6;
  bool neg = true, pos = true;   // imag = -w sin (reference) or +w sin: the power spectrum does not care
  for (int n = 0; n < 1024; ++n)
    for (int k = 0; k < nb; ++k) {
      const int a = (int)(((int64_t)k * n) & 1023);
      const double w = win[n];
      if (std::fabs(re[(size_t)n * nb + k] - w * ct[a]) > tol) return o;
      const double iv = im[(size_t)n * nb + k];
      if (std::fabs(iv + w * st[a]) > tol) neg = false;
      if (std::fabs(iv - w * st[a]) > tol) pos = false;
      if (!neg && !pos) return o;
    }
  // stage 1: K = n1 (32), columns [Re k1 (32) | Im k1 (32)] of W32^(n1 k1)
  // stage 1: K = n1 (32), columns [Re k1 (32) | Im k1 (32)] of W32^(n1 k1)
  auto f1 = [&](int k, int n) { const int a = ((k * (n & 31)) & 31) * 32; return (float)(n < 32 ? ct[a] : -st[a]); };
  // stage 2: K = [Re n2 (32) | Im n2 (32)], columns [Re k2 (16) | Im k2 (16)] of W32^(n2 k2)
  auto f2 = [&](int k, int n) {
    const int a = (((k & 31) * (n & 15)) & 31) * 32;
    if (k < 32) return (float)(n < 16 ? ct[a] : -st[a]);
    return (float)(n < 16 ? st[a] : ct[a]);
  };
  o.w1 = ab.put(pack_p16(f1, 32, 64, 4));
  o.w2 = ab.put(pack_p16(f2, 64, 32, 2));
  o.w1s = ab.put(pack_split32(f1, 32, 64));      // the same matrices as exact three-term bf16 fragments (round 3)
  o.w2s = ab.put(pack_split32(f2, 64, 32));
  o.w1h = ab.put(pack_half32(f1, 32, 64, 16384.f));      // two-term scheme: |cos|, |sin| <= 1 times 2^14
  o.w2h = ab.put(pack_half32(f2, 64, 32, 16384.f));
  std::vector<float> tc(1024), ts(1024);
  for (int k1 = 0; k1 < 32; ++k1)
    for (int n2 = 0; n2 < 32; ++n2) { tc[k1 * 32 + n2] = (float)ct[k1 * n2]; ts[k1 * 32 + n2] = (float)st[k1 * n2]; }
  o.twc = ab.put(tc);
  o.tws = ab.put(ts);
  o.win = ab.put(win);
  o.ok = true;
  return o;
}


MelBandOff pack_mel_band(ArenaBuilder& ab, const std::vector<float>& f2m, int nb, int n_mels) {
  MelBandOff o;
  // MI355ASR_MEL_BAND=0: always the dense mel GEMM
  static const bool on = mi355_env("MI355ASR_MEL_BAND", 1) != 0;
  if (!on) return o;
  std::vector<int> band(2 * (size_t)n_mels, 0);
  int bw = 4;
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < nb; ++k)
      if (f2m[(size_t)k * n_mels + m] != 0.f) { if (lo < 0) lo = k; hi = k; }
    if (lo >= 0) { band[2 * m] = lo; band[2 * m + 1] = hi - lo + 1; bw = std::max(bw, hi - lo + 1); }
  }
  if (bw > 64) return o;
  bw = (bw + 3) & ~3;
  const int lp = ((nb + 15) / 16) * 16;                          // the kernel reads bins [lo, lo + bw) of a row of >= lp floats
  for (int m = 0; m < n_mels; ++m)
    if (band[2 * m] + bw > lp) return o;
  std::vector<float> w((size_t)n_mels * bw, 0.f);
  for (int m = 0; m < n_mels; ++m)
    for (int j = 0; j < band[2 * m + 1]; ++j) w[(size_t)m * bw + j] = f2m[(size_t)(band[2 * m] + j) * n_mels + m];
  std::vector<float> band_f(band.size());
  std::memcpy(band_f.data(), band.data(), band.size() * sizeof(int));   // the arena is a float array: raw bits
  o.band = ab.put(band_f);
  o.bw = ab.put(w);
  o.BW = bw;
  o.ok = true;
  return o;
}
void use_mel_band(mi355asr_model* m, const MelBandOff& o, const float* base) {
  m->mel_band = o.ok ? reinterpret_cast<const int*>(base + o.band) : nullptr;
  m->mel_bw = o.ok ? base + o.bw : nullptr;
  m->mel_BW = o.ok ? o.BW : 0;
}
int launch_mel_auto(const mi355asr_model* m, MelArgs& me, hipStream_t s) {
  if (m->mel_band) {
    me.band = m->mel_band; me.bw = m->mel_bw; me.BW = m->mel_BW;
    if (launch_mel_band(me, s) == 0) return 0;
  }
  me.absmax = nullptr;          // the dense kernel does not produce the run-time maximum: the caller must not rely on it
  return launch_mel(me, s);
}

// Slab ring of gemm_ring.hip: [N / 128 chunks][K / 32 steps][8 column tiles][3 terms][64 lanes][8 bf16]; a GLU layer's
// chunk holds four value tiles and the four gate tiles that go with them.
void put_ring(ArenaBuilder& ab, size_t p16_off, const std::function<float(int, int)>& f, int K, int N, bool glu) {
  if (K % 128 != 0 || N % (glu ? 128 * 2 : 128) != 0) return;
  const int terms = ab.ring_terms;
  const std::vector<float> sp = pack_split32(f, K, N);       // [step][NT][3 terms][64 lanes][8 bf16]
  constexpr size_t TERM = 64 * 8 / 2;                         // floats per (step, column tile, term)
  const int steps = K / 32, NT = N / 16, chunks = NT / 8, half = NT / 2;
  std::vector<float> ring((size_t)chunks * steps * 8 * terms * TERM);
  for (int ch = 0; ch < chunks; ++ch)
    for (int st = 0; st < steps; ++st)
      for (int i = 0; i < 8; ++i) {
        const int tile = glu ? (i < 4 ? 4 * ch + i : half + 4 * ch + (i - 4)) : 8 * ch + i;
        // bf16 mode keeps term 0 only: round-to-nearest-even bf16 of the weight, what the bf16 arena holds as well
        std::memcpy(ring.data() + (((size_t)ch * steps + st) * 8 + i) * terms * TERM, sp.data() + ((size_t)st * NT + tile) * 3 * TERM,
                    terms * TERM * sizeof(float));
      }
  ab.ring_pairs.emplace_back(p16_off, ab.put(ring));
}
// The class head W[K, V] as a slab ring: V padded with zero columns to whole chunks of 128 (the kernel never looks at
// classes >= V).
void put_ring_head(ArenaBuilder& ab, size_t p16_off, const std::function<float(int, int)>& f, int K, int V) {
  if (K % 128 != 0 || V < 1) return;
  put_ring(ab, p16_off, [&](int k, int n) { return n < V ? f(k, n) : 0.f; }, K, ceil_div(V, 128) * 128, false);
}
// MI355ASR_GEMM_RING=0: the dense layers of dmodel 256 / 512 stay on the fp32-MFMA kernels (chain2 / gemm16<PF32>), or
// in bf16 mode on gemm16<PBf16>
// rows from which launch_gemm16 hands a dense layer to the ring kernels (crossover measured below)
long ring_min_rows() {
  static const long v = mi355_env("MI355ASR_RING_MIN_M", 1500);
  return v;
}
bool ring_packs_wanted(const mi355asr_model* m) {
  static const bool on = mi355_env("MI355ASR_GEMM_RING", 1) != 0;
  // (bf16 mode, dmodel 256: chain256_bf16_kernel reads the one-term ring packs at every row count)
  return on && m->cfg.dmodel % 128 == 0 &&
         (m->expected_rows < 0 || m->expected_rows >= ring_min_rows() || (m->cfg.gemm_dtype == 1 && m->cfg.dmodel == 256));
}
void register_rings(mi355asr_model* m, const ArenaBuilder& ab, const float* base) {
  for (const auto& pr : ab.ring_pairs) m->ring_of[base + pr.first] = base + pr.second;
  m->head_of.clear();
  for (const auto& hp : ab.head_pairs) m->head_of[base + hp.p16] = {base + hp.slabs, hp.groups, base + hp.pp, hp.pp_sw, base + hp.ns};
}
void put_head_slabs(ArenaBuilder& ab, size_t p16_off, const std::function<float(int, int)>& f, int d, int V, const float* bias) {
  if (d != 144 || V < 1) return;
  const int groups = ceil_div(ceil_div(V, 16), 9);
  std::vector<float> st;
  append_slabs(st, [&](int k, int n) { return n < V ? f(k, n) : 0.f; }, d, 144 * groups, true);
  const size_t o_st = ab.put(st);
  // the same matrix as the two-term fp16 stream of pp_head_kernel: column groups of nine tiles, five plain ring slots each,
  // the bias in row 144
  std::vector<float> pp, plain;
  const float sw = append_pp_plain(pp, [&](int k, int n) { return n < V ? (k < d ? f(k, n) : bias[n]) : 0.f; }, groups, &plain);
  const size_t o_pp = ab.put(pp);
  ab.head_pairs.push_back({p16_off, o_st, groups, o_pp, sw, ab.put(plain)});
}
int try_head_ld(const mi355asr_model* m, const GemmArgs& hd, hipStream_t s, float* split_scratch) {
  // a launch of the ring kernel costs as much for 250 rows as for 16 000: from 2048 rows on
  if (m->cfg.gemm_dtype != 0 || m->head_of.empty()) return -1;
  const auto it = m->head_of.find(hd.wp);
  if (it == m->head_of.end()) return -1;
  // round 6, small batches: one 16-token tile per workgroup, the column tiles split over its waves (fused_ns.hip)
  if (launch_ns1_head(hd, it->second.ns, it->second.pp_sw, it->second.groups, s) == 0) return 0;
  if (hd.M < 2048) return -1;
  // few rows and many classes (the Translator's 144 -> 9160 over ~6 000 rows is 93 row workgroups): the column groups split over
  // several workgroups per row tile, the per-range arg-max pairs combined by a second small launch (split_scratch: 16 M words)
  if ((split_scratch || (!hd.argmax_out && !hd.maxval_out)) && launch_pp_head_split(hd, it->second.pp, it->second.pp_sw, it->second.groups, pp_head_ranges(hd.M, it->second.groups), split_scratch, s) == 0) return 0;
  if (launch_pp_head(hd, it->second.pp, it->second.pp_sw, it->second.groups, s) == 0) return 0;
  return launch_head_ld(hd, it->second.slabs, it->second.groups, s);
}

BlockOff pack_block(mi355asr_model* m, ArenaBuilder& ab, const std::string& p, int d, int H, int hs, int k,
                    bool keras_mha) {
  auto T = [&](const std::string& n) -> const std::vector<float>& { return m->host[n].data; };
  BlockOff o;
  const bool rings = ring_packs_wanted(m);
  const char* ffn[2] = {"ff_module_1", "ff_module_2"};
  for (int i = 0; i < 2; ++i) {
    const std::string q = p + "/" + ffn[i];
    o.ff_ln_g[i] = ab.put(T(q + "/ln/gamma"));
    o.ff_ln_b[i] = ab.put(T(q + "/ln/beta"));
    const auto& w1 = T(q + "/ffn1/kernel");
    const auto& w2 = T(q + "/ffn2/kernel");
    o.ff_w1p[i] = ab.put(pack_p16([&](int kk, int n) { return w1[(size_t)kk * 4 * d + n]; }, d, 4 * d, 4 * d / 16));
    if (rings) put_ring(ab, o.ff_w1p[i], [&](int kk, int n) { return w1[(size_t)kk * 4 * d + n]; }, d, 4 * d, false);
    o.ff_b1[i] = ab.put(T(q + "/ffn1/bias"));
    o.ff_w2p[i] = ab.put(pack_p16([&](int kk, int n) { return w2[(size_t)kk * d + n]; }, 4 * d, d, d / 16));
    if (rings) put_ring(ab, o.ff_w2p[i], [&](int kk, int n) { return w2[(size_t)kk * d + n]; }, 4 * d, d, false);
    o.ff_b2[i] = ab.put(T(q + "/ffn2/bias"));
  }
  const std::string a = p + "/mhsa_module";
  std::function<float(int, int)> qkv_at;
  o.att_ln_g = ab.put(T(a + "/ln/gamma"));
  o.att_ln_b = ab.put(T(a + "/ln/beta"));
  if (keras_mha) {
    // Keras MHA kernels are [d, H, hs] = [d, d] row-major: column n = which*d + h*hs + o
    const auto &qk = T(a + "/mha/query/kernel"), &kk_ = T(a + "/mha/key/kernel"), &vk = T(a + "/mha/value/kernel");
    o.qkv_wp = ab.put(pack_p16(
        [&](int i, int n) {
          const int which = n / d, r = n % d;
          const std::vector<float>& w = which == 0 ? qk : (which == 1 ? kk_ : vk);
          return w[(size_t)i * d + r];
        },
        d, 3 * d, 3 * d / 16));
    std::vector<float> qb(3 * d);
    const auto &bq = T(a + "/mha/query/bias"), &bk = T(a + "/mha/key/bias"), &bv = T(a + "/mha/value/bias");
    for (int i = 0; i < d; ++i) { qb[i] = bq[i]; qb[d + i] = bk[i]; qb[2 * d + i] = bv[i]; }
    o.qkv_b = ab.put(qb);
    const auto& pk = T(a + "/mha/attention_output/kernel");  // [H, hs, d]: row k = h*hs + i
    o.out_wp = ab.put(pack_p16([&](int kk, int n) { return pk[(size_t)kk * d + n]; }, d, d, d / 16));
    o.out_b = ab.put(T(a + "/mha/attention_output/bias"));
    qkv_at = [&qk, &kk_, &vk, d](int i, int n) {
      const int which = n / d, r = n % d;
      const std::vector<float>& w = which == 0 ? qk : (which == 1 ? kk_ : vk);
      return w[(size_t)i * d + r];
    };
  } else {
  const auto& qk = T(a + "/mha/query_kernel");
  const auto& kk_ = T(a + "/mha/key_kernel");
  const auto& vk = T(a + "/mha/value_kernel");
  // einsum "BNI,HIO->BNHO": column n = which*d + h*hs + o  <-  kernel[h][i][o]
  o.qkv_wp = ab.put(pack_p16(
      [&](int i, int n) {
        const int which = n / d, r = n % d, h = r / hs, oo = r % hs;
        const std::vector<float>& w = which == 0 ? qk : (which == 1 ? kk_ : vk);
        return w[((size_t)h * d + i) * hs + oo];
      },
      d, 3 * d, 3 * d / 16));
  o.qkv_b = ab.put(std::vector<float>(3 * d, 0.f));
  const auto& pk = T(a + "/mha/projection_kernel");  // [H, hs, d]: row k = h*hs + i
  o.out_wp = ab.put(pack_p16([&](int kk, int n) { return pk[(size_t)kk * d + n]; }, d, d, d / 16));
  o.out_b = ab.put(T(a + "/mha/projection_bias"));
  qkv_at = [&qk, &kk_, &vk, d, hs](int i, int n) {
    const int which = n / d, r = n % d, h = r / hs, oo = r % hs;
    const std::vector<float>& w = which == 0 ? qk : (which == 1 ? kk_ : vk);
    return w[((size_t)h * d + i) * hs + oo];
  };
  }
  if (rings) {
    const auto& pk_r = keras_mha ? T(a + "/mha/attention_output/kernel") : T(a + "/mha/projection_kernel");
    put_ring(ab, o.qkv_wp, qkv_at, d, 3 * d, false);
    put_ring(ab, o.out_wp, [&](int kk, int n) { return pk_r[(size_t)kk * d + n]; }, d, d, false);
  }
  if (d != 144 && d % 64 == 0 && hs == 64) {
    // round 5: the same operand bounds for the head-size-64 models (attention_split64_kernel): q / k / v of the layer-at-a-time
    // projections.  In bf16 mode weights and activations are rounded to bf16 first: each factor grows by at most 2^-8.
    const auto &lg = T(a + "/ln/gamma"), &lb = T(a + "/ln/beta");
    std::vector<float> qb0(3 * d, 0.f);
    if (keras_mha) {
      const auto &bq = T(a + "/mha/query/bias"), &bk = T(a + "/mha/key/bias"), &bv = T(a + "/mha/value/bias");
      for (int i = 0; i < d; ++i) { qb0[i] = bq[i]; qb0[d + i] = bk[i]; qb0[2 * d + i] = bv[i]; }
    }
    const double lnb = std::sqrt((double)(d - 1));
    double bnd[3] = {0.0, 0.0, 0.0};
    for (int n = 0; n < 3 * d; ++n) {
      double sum = std::fabs((double)qb0[n]);
      for (int i = 0; i < d; ++i) sum += std::fabs((double)qkv_at(i, n)) * (lnb * std::fabs((double)lg[i]) + std::fabs((double)lb[i]));
      bnd[n / d] = std::max(bnd[n / d], sum);
    }
    bnd[0] *= 1.4426950408889634 / std::sqrt((double)hs);
    for (int k = 0; k < 3; ++k) o.att_h2[k] = half_scale_for(bnd[k] * 1.01, 40);
  }
  if (d == 144) {
    o.split = true;                      // the slab streams of the loader-wave kernels (fused.hip) and of the pair-pipelined ones (fused_pp.hip)
    // slab stream of ff1_qkv_ring_kernel: per hidden chunk of 144 the five steps of W1[:, chunk] and of W2[chunk, :],
    // then q, k, v (five steps each)
    const auto& f1 = T(p + "/ff_module_1/ffn1/kernel");
    const auto& f2 = T(p + "/ff_module_1/ffn2/kernel");
    std::vector<float> st;
    for (int ch = 0; ch < 4; ++ch) {
      append_slabs(st, [&](int kk, int n) { return f1[(size_t)kk * 4 * d + d * ch + n]; }, d, d, false);
      append_slabs(st, [&](int kk, int n) { return f2[(size_t)(d * ch + kk) * d + n]; }, d, d, false);
    }
    append_slabs(st, qkv_at, d, 3 * d, true);
    o.ff1_slabs = ab.put(st);
    // pair-pipelined stream (fused_pp.hip): ff_module_1 as 18 hidden pairs, bias in row 144 of W1; then q, k, v with the
    // q / k / v bias (zero for the reference's own attention layer) in row 144
    const auto& b1 = T(p + "/ff_module_1/ffn1/bias");
    const std::vector<float> qb = keras_mha ? [&] {
      std::vector<float> v(3 * d);
      const auto &bq = T(a + "/mha/query/bias"), &bk = T(a + "/mha/key/bias"), &bv = T(a + "/mha/value/bias");
      for (int i = 0; i < d; ++i) { v[i] = bq[i]; v[d + i] = bk[i]; v[2 * d + i] = bv[i]; }
      return v;
    }() : std::vector<float>(3 * d, 0.f);
    std::vector<float> pp, n1, n2, nq;
    o.pp_ff1_sc = append_pp_chain(pp, [&](int kk, int n) { return kk < d ? f1[(size_t)kk * 4 * d + n] : b1[n]; }, 4 * d,
                                  [&](int kk, int n) { return f2[(size_t)kk * d + n]; }, &n1, &n2);
    o.pp_sw_qkv = append_pp_plain(pp, [&](int kk, int n) { return kk < d ? qkv_at(kk, n) : qb[n]; }, 3, &nq);
    o.pp_ff1 = ab.put(pp);
    o.ns = true;                         // the same fragments in plain order, for the N-split kernels (fused_ns.hip: small batches)
    o.ns_ff1_w1 = ab.put(n1); o.ns_ff1_w2 = ab.put(n2); o.ns_qkv = ab.put(nq);
    // Operand bounds for the two-term attention kernel: a LayerNorm output lies in sqrt(d - 1) |gamma_i| + |beta_i|, so
    // |q_n|, |k_n|, |v_n| <= sum_i |W_in| (sqrt(d - 1) |gamma_i| + |beta_i|) + |b_n| (q times the query scale and log2 e,
    // which the kernel folds into it)
    {
      const auto &lg = T(a + "/ln/gamma"), &lb = T(a + "/ln/beta");
      const double lnb = std::sqrt((double)(d - 1));
      double bnd[3] = {0.0, 0.0, 0.0};
      for (int n = 0; n < 3 * d; ++n) {
        double sum = std::fabs((double)qb[n]);
        for (int i = 0; i < d; ++i) sum += std::fabs((double)qkv_at(i, n)) * (lnb * std::fabs((double)lg[i]) + std::fabs((double)lb[i]));
        bnd[n / d] = std::max(bnd[n / d], sum);
      }
      bnd[0] *= 1.4426950408889634 / std::sqrt((double)hs);
      // fp32 mode: 1.0001 covers the rounding of the bound's own evaluation.  bf16 mode (gemm_dtype 1: the generic per-layer path
      // rounds weights AND activations to bf16 before the projections, each factor growing by up to 2^-8) takes the 1.01 margin of
      // the head-size-64 bounds above, so that bound * scale <= 2^15 holds there too (round-5 advice)
      const double margin = m->cfg.gemm_dtype == 1 ? 1.01 : 1.0001;
      o.att_h2[0] = half_scale_for(bnd[0] * margin, 40);
      o.att_h2[1] = half_scale_for(bnd[1] * margin, 40);
      o.att_h2[2] = half_scale_for(bnd[2] * margin, 40);
    }
  }
  const std::string c = p + "/conv_module";
  o.cv_ln_g = ab.put(T(c + "/ln/gamma"));
  o.cv_ln_b = ab.put(T(c + "/ln/beta"));
  const auto& pw1 = T(c + "/pw_conv_1/kernel");
  o.pw1_wp = ab.put(pack_p16([&](int kk, int n) { return pw1[(size_t)kk * 2 * d + n]; }, d, 2 * d, 2 * d / 16));
  if (rings) put_ring(ab, o.pw1_wp, [&](int kk, int n) { return pw1[(size_t)kk * 2 * d + n]; }, d, 2 * d, true);
  if (o.split) {
    // slab stream of out_glu_ring_kernel: out-projection (5 slabs), then pw_conv_1 step by step (value | gate)
    const auto& pk2 = keras_mha ? T(a + "/mha/attention_output/kernel") : T(a + "/mha/projection_kernel");
    std::vector<float> st;
    append_slabs(st, [&](int kk, int n) { return pk2[(size_t)kk * d + n]; }, d, d, false);
    append_slabs(st, [&](int kk, int n) { return pw1[(size_t)kk * 2 * d + n]; }, d, 2 * d, false);
    o.og_slabs = ab.put(st);
    // two-term fp16 stream of pp_out_glu_kernel: out projection (one group of nine tiles), then pw_conv_1's value tiles and
    // gate tiles (two groups), five plain ring slots each, the biases in row 144
    const auto& ob = keras_mha ? T(a + "/mha/attention_output/bias") : T(a + "/mha/projection_bias");
    const auto& p1b = T(c + "/pw_conv_1/bias");
    std::vector<float> pp, no, np1;
    o.pp_sw_out = append_pp_plain(pp, [&](int kk, int n) { return kk < d ? pk2[(size_t)kk * d + n] : ob[n]; }, 1, &no);
    o.pp_sw_pw1 = append_pp_plain(pp, [&](int kk, int n) { return kk < d ? pw1[(size_t)kk * 2 * d + n] : p1b[n]; }, 2, &np1);
    o.pp_og = ab.put(pp);
    o.ns_out = ab.put(no); o.ns_pw1 = ab.put(np1);
  }
  o.pw1_b = ab.put(T(c + "/pw_conv_1/bias"));
  o.dw_w = ab.put(T(c + "/dw_conv/depthwise_kernel"));  // [k, d, 1] == [k][d]
  const auto& pc = T(c + "/dw_conv/pointwise_kernel");
  o.pc_w1p = ab.put(pack_p16([&](int kk, int n) { return pc[(size_t)kk * 2 * d + n]; }, d, 2 * d, 2 * d / 16));
  if (rings) put_ring(ab, o.pc_w1p, [&](int kk, int n) { return pc[(size_t)kk * 2 * d + n]; }, d, 2 * d, false);
  o.pc_b1 = ab.put(T(c + "/dw_conv/bias"));
  {
    const auto &g = T(c + "/bn/gamma"), &b = T(c + "/bn/beta"), &mu = T(c + "/bn/moving_mean"),
               &var = T(c + "/bn/moving_variance");
    std::vector<float> s(2 * d), t(2 * d);
    for (int i = 0; i < 2 * d; ++i) {
      s[i] = g[i] / std::sqrt(var[i] + kBnEps);
      t[i] = b[i] - mu[i] * s[i];
    }
    o.bn_s = ab.put(s);
    o.bn_t = ab.put(t);
  }
  const auto& pw2 = T(c + "/pw_conv_2/kernel");
  o.pw2_wp = ab.put(pack_p16([&](int kk, int n) { return pw2[(size_t)kk * d + n]; }, 2 * d, d, d / 16));
  if (rings) put_ring(ab, o.pw2_wp, [&](int kk, int n) { return pw2[(size_t)kk * d + n]; }, 2 * d, d, false);
  if (o.split) {
    // slab stream of tail_ff2_ring_kernel: per hidden chunk of 144 the five steps of W1[:, chunk] and of W2[chunk, :]
    // for the conv tail (pointwise 144 -> 288, pw_conv_2 288 -> 144), then for FFModule 2 (144 -> 576 -> 144)
    const auto& f1 = T(p + "/ff_module_2/ffn1/kernel");
    const auto& f2 = T(p + "/ff_module_2/ffn2/kernel");
    std::vector<float> st;
    for (int ch = 0; ch < 2; ++ch) {
      append_slabs(st, [&](int kk, int n) { return pc[(size_t)kk * 2 * d + d * ch + n]; }, d, d, false);
      append_slabs(st, [&](int kk, int n) { return pw2[(size_t)(d * ch + kk) * d + n]; }, d, d, false);
    }
    for (int ch = 0; ch < 4; ++ch) {
      append_slabs(st, [&](int kk, int n) { return f1[(size_t)kk * 4 * d + d * ch + n]; }, d, d, false);
      append_slabs(st, [&](int kk, int n) { return f2[(size_t)(d * ch + kk) * d + n]; }, d, d, false);
    }
    o.tail_slabs = ab.put(st);
    // pair-pipelined stream: the conv tail as 9 hidden pairs with the folded BatchNorm in the weights -- column n of the
    // pointwise kernel times scale[n], row 144 = bias[n] * scale[n] + shift[n] (products formed in double, rounded once) --
    // then ff_module_2 as 18 pairs with its bias in row 144
    const auto& pcb = T(c + "/dw_conv/bias");
    const auto& fb1 = T(p + "/ff_module_2/ffn1/bias");
    std::vector<float> bs(2 * d), bt(2 * d);
    {
      const auto &g = T(c + "/bn/gamma"), &b = T(c + "/bn/beta"), &mu = T(c + "/bn/moving_mean"), &var = T(c + "/bn/moving_variance");
      for (int i = 0; i < 2 * d; ++i) { bs[i] = g[i] / std::sqrt(var[i] + kBnEps); bt[i] = b[i] - mu[i] * bs[i]; }
    }
    std::vector<float> pp, c1, c2, n1, n2;
    o.pp_tail_sc[0] = append_pp_chain(pp, [&](int kk, int n) {
      return kk < d ? (float)((double)pc[(size_t)kk * 2 * d + n] * (double)bs[n]) : (float)((double)pcb[n] * (double)bs[n] + (double)bt[n]);
    }, 2 * d, [&](int kk, int n) { return pw2[(size_t)kk * d + n]; }, &c1, &c2);
    o.pp_tail_sc[1] = append_pp_chain(pp, [&](int kk, int n) { return kk < d ? f1[(size_t)kk * 4 * d + n] : fb1[n]; }, 4 * d,
                                      [&](int kk, int n) { return f2[(size_t)kk * d + n]; }, &n1, &n2);
    o.pp_tail = ab.put(pp);
    o.ns_cv_w1 = ab.put(c1); o.ns_cv_w2 = ab.put(c2); o.ns_ff2_w1 = ab.put(n1); o.ns_ff2_w2 = ab.put(n2);
  }
  o.pw2_b = ab.put(T(c + "/pw_conv_2/bias"));
  o.ln_g = ab.put(T(p + "/ln/gamma"));
  o.ln_b = ab.put(T(p + "/ln/beta"));
  (void)H;
  (void)k;
  return o;
}

BlockDev resolve(const BlockOff& o, const float* base) {
  BlockDev b;
  if (o.cross) { b.xq_wp = base + o.xq_wp; b.xkv_wp = base + o.xkv_wp; }
  for (int i = 0; i < 2; ++i) {
    b.ff_ln_g[i] = base + o.ff_ln_g[i];
    b.ff_ln_b[i] = base + o.ff_ln_b[i];
    b.ff_w1p[i] = base + o.ff_w1p[i];
    b.ff_b1[i] = base + o.ff_b1[i];
    b.ff_w2p[i] = base + o.ff_w2p[i];
    b.ff_b2[i] = base + o.ff_b2[i];
  }
  b.att_ln_g = base + o.att_ln_g; b.att_ln_b = base + o.att_ln_b;
  b.qkv_wp = base + o.qkv_wp; b.qkv_b = base + o.qkv_b;
  b.out_wp = base + o.out_wp; b.out_b = base + o.out_b;
  b.cv_ln_g = base + o.cv_ln_g; b.cv_ln_b = base + o.cv_ln_b;
  b.pw1_wp = base + o.pw1_wp; b.pw1_b = base + o.pw1_b;
  if (o.split) { b.og_slabs = base + o.og_slabs; b.ff1_slabs = base + o.ff1_slabs; b.tail_slabs = base + o.tail_slabs; b.pp_ff1 = base + o.pp_ff1; b.pp_tail = base + o.pp_tail; b.pp_ff1_sc = o.pp_ff1_sc; b.pp_sw_qkv = o.pp_sw_qkv; b.pp_tail_sc[0] = o.pp_tail_sc[0]; b.pp_tail_sc[1] = o.pp_tail_sc[1]; b.pp_og = base + o.pp_og; b.pp_sw_out = o.pp_sw_out; b.pp_sw_pw1 = o.pp_sw_pw1; }
  if (o.ns) {
    b.ns_ff1_w1 = base + o.ns_ff1_w1; b.ns_ff1_w2 = base + o.ns_ff1_w2; b.ns_qkv = base + o.ns_qkv; b.ns_out = base + o.ns_out; b.ns_pw1 = base + o.ns_pw1;
    b.ns_cv_w1 = base + o.ns_cv_w1; b.ns_cv_w2 = base + o.ns_cv_w2; b.ns_ff2_w1 = base + o.ns_ff2_w1; b.ns_ff2_w2 = base + o.ns_ff2_w2;
  }
  b.att_h2[0] = o.att_h2[0]; b.att_h2[1] = o.att_h2[1]; b.att_h2[2] = o.att_h2[2];
  b.dw_w = base + o.dw_w;
  b.pc_w1p = base + o.pc_w1p; b.pc_b1 = base + o.pc_b1;
  b.bn_s = base + o.bn_s; b.bn_t = base + o.bn_t;
  b.pw2_wp = base + o.pw2_wp; b.pw2_b = base + o.pw2_b;
  b.ln_g = base + o.ln_g; b.ln_b = base + o.ln_b;
  return b;
}

// Layer-at-a-time GEMM family (bf16.hip) instead of the fused / chained fp32 kernels: in bf16 mode, and in fp32 for
// dmodel values those kernels are not instantiated for (e.g. 512 = ConformerL).
bool use_gemm16(const mi355asr_model* m) {
  static const bool force = mi355_env("MI355ASR_GEMM16", 0) != 0;
  // dmodel 256 with slab rings (gemm_ring.hip): one launch per dense layer on the split-bf16 pipe beats the fp32 chains
  return force || m->cfg.gemm_dtype == 1 || (m->cfg.dmodel != 144 && m->cfg.dmodel != 256) ||
         (m->cfg.dmodel == 256 && !m->ring_of.empty());
}
// Very few rows (one streaming chunk of 13 frames, the Translator's token stream): the fused kernels give each 16-row tile
// to ONE wave that walks a whole run of layers serially -- a fused launch takes as long for 16 rows as for 16 000 -- and one
// launch per layer with K / column splitting is as fast.  From a few tiles on the fused path wins: round 4 measured ONE
// utterance (ms per recognize(), fused vs layer-at-a-time) 10 s / 250 rows 1.234 vs 1.386, 5 s / 125 rows 1.143 vs 1.263,
// 2 s / 50 rows 1.118 vs 1.224, and B = 2, 3 at 10 s 1.241 / 1.252 vs 1.513 / 1.651 (profiles/r04_batch_sweep.md; the
// round-2 ring kernels had crossed at ~800 rows, which is where this threshold stood until round 4).  MI355ASR_SMALL_M overrides.
bool gemm16_for(const mi355asr_model* m, size_t M) {
  static const long small_m = mi355_env("MI355ASR_SMALL_M", 48);
  return use_gemm16(m) || (long)M <= small_m;
}
int launch_gemm16(const mi355asr_model* m, int epi, bool ln, Gemm16Args& g, const float* wp, hipStream_t s) {
  // long batches of dmodel 256 / 512: the same layer with the weights as a slab ring shared by eight waves
  // (gemm_ring.hip): fp32 operands exactly split into three bf16 terms, or one bf16 term in bf16 mode
  // crossover measured with 10 s utterances (tools/model_batch_sweep.py; ms per batch, ring vs per-wave streams):
  // ConformerM B = 4: 3.81 vs 3.10, 8: 4.09 vs 4.26, 16: 4.47 vs 5.19; ConformerL 4: 6.95 vs 6.57, 8: 7.59 vs 9.63, 16: 10.3 vs 18.5
  if ((long)g.M >= ring_min_rows() && !m->ring_of.empty()) {
    const auto it = m->ring_of.find(wp);
    g.wp = wp;
    // round 6, bf16 mode, K = 256, from 8 192 rows: the rows resident in LDS, the column tiles split over the waves (bf16.hip)
    if (it != m->ring_of.end() && m->cfg.gemm_dtype == 1 && launch_gemm256_bf16(epi, ln, g, it->second, s) == 0) return 0;
    if (it != m->ring_of.end() && launch_gemm_ring(epi, ln, g, it->second, m->cfg.gemm_dtype == 1 ? 1 : 3, s) == 0) return 0;
  }
  if (m->cfg.gemm_dtype == 1) { g.wp = m->w16(wp); return launch_gemm16_bf16(epi, ln, g, s); }
  g.wp = wp;
  return launch_gemm16_f32(epi, ln, g, s);
}

// ---- workspace plan (byte offsets, 256-byte aligned) --------------------------------------------------


size_t wavpick_floats(const mi355asr_model* m, int Bp, int Lmax);

// Bp = number of independent encoder inputs (utterances, or utterances x blocks when streaming),
// F mel frames and T encoder frames per input.
Plan make_plan(const mi355asr_model* m, int Bp, int F, int T) {
  const int d = m->cfg.dmodel;
  const size_t M = (size_t)Bp * T;
  Plan p;
  size_t o = 0;
  auto take = [&](size_t floats) {
    size_t at = o;
    o = align256(o + floats * 4);
    return at;
  };
  p.xa = take(M * d);
  p.xb = take(M * d);
  p.qkv = take(M * 3 * d);
  p.ctx = take(M * d);
  p.u = take(M * d);
  p.dw = take(M * d);
  p.h4 = gemm16_for(m, M) ? take(M * 4 * d) : 0;   // before logp: the block-only entry points size to p.logp
  p.enc = take(M * d);
  p.amax = take(M);
  const int FT = ceil_div(F, 16);
  p.logp = take((size_t)Bp * F * m->dm.LP);
  p.pmax = take((size_t)Bp * std::max(FT * m->dm.NCH_dft, F));
  p.umax = take(Bp);
  p.mel = take((size_t)Bp * F * m->cfg.n_mels);
  p.sub = take(M * m->dm.F2 * d);
  p.wv_floats = wavpick_floats(m, Bp, F * m->dm.hop);      // add_wav_info branch (0 when off); L <= F * hop
  p.wv = p.wv_floats ? take(p.wv_floats) : 0;
  p.total = o;
  return p;
}


int geometry(const mi355asr_model* m, int B, int L, Geometry* g) {
  const auto& c = m->cfg;
  if (B <= 0 || L <= 0) return fail(MI355ASR_EINVAL, "B and L must be positive (B=%d, L=%d)", B, L);
  g->nblk = 1;
  g->Lb = L;
  if (c.chunk_size > 0) {
    if (L % c.chunk_size != 0)
      return fail(MI355ASR_EINVAL, "streaming encoder: L=%d is not a multiple of chunk_size=%d "
                  "(the reference reshapes [B,L,1]->[-1,chunk,1], conformer_blocks.py:585)", L, c.chunk_size);
    g->nblk = L / c.chunk_size;
    g->Lb = c.chunk_size;
  }
  g->Bp = B * g->nblk;
  g->F = ceil_div(g->Lb, m->dm.hop);
  g->T1 = ceil_div(g->F, m->dm.st1);
  g->T = ceil_div(g->T1, 2);
  return 0;
}

// ---- launch sequences ---------------------------------------------------------------------------------

// One ConformerBlock (conformer_blocks.py:259-265).  Input in sc.xa, output to `out` (or sc.xa if null).
// Input in sc.xa; output to `out`, or (out == nullptr) left in sc.xa -- the fused path ping-pongs xa/xb by swapping
// the two pointers in `sc` instead of copying.
// RBlock of the Translator (conformer_blocks.py:455-463, 496-503): the attention is a cross-attention with
// q = LN(x + PE) and k = v = the encoder output (T_enc frames per utterance), everything else is a ConformerBlock.

static bool fused_env_on() {
  static const bool on = mi355_env("MI355ASR_FUSED", 1) != 0;
  return on;
}
bool block_takes_pre(const mi355asr_model* m, const BlockDev& w, size_t M) {
  return m->cfg.dmodel == 144 && fused_env_on() && !gemm16_for(m, M) && w.pp_ff1 && w.ff1_slabs && ff1_pre_selected();
}

int run_block(const mi355asr_model* m, const BlockDev& w, const BlockOpts& bo, Scratch& sc, int B, int T,
              float* out, hipStream_t s, const CrossAttn* cross, const BlockDev* next, bool* ff1_done, bool skip_ff1) {
  if (ff1_done) *ff1_done = false;
  const int d = m->cfg.dmodel, H = m->cfg.num_heads, hs = m->cfg.head_size;
  const int ksz = bo.ksz;
  const float fc = bo.fc;
  const int M = B * T;
  const bool fused_env = fused_env_on();
  if (bo.pre_pp && (cross || skip_ff1 || !block_takes_pre(m, w, (size_t)M)))
    return fail(MI355ASR_ESTATE, "run_block: a layer in front of a block that cannot take it");
  if (gemm16_for(m, M)) {
    // one launch per dense layer (bf16.hip: bf16 or fp32 operands); LayerNorm / softmax / activations / depthwise conv in fp32
    auto g16 = [&](const float* x, int ldx, int K, const float* wp, const float* bias, int NT, float* y, int ldy) {
      Gemm16Args g{};
      g.x = x; g.ldx = ldx; g.K = K; g.wp = wp; g.bias = bias; g.NT = NT; g.y = y; g.ldy = ldy;
      g.M = M; g.n_valid = 16 * NT; g.eps = kLnEps; g.scale = 1.0f;
      return g;
    };
    // round 4: bf16 mode, dmodel 256: FFModule and ConvModule tail as ONE launch each (bf16.hip: chain256_bf16_kernel; the
    // hidden activation stays in LDS) -- MI355ASR_CHAIN256=0: one gemm16 / gemm_ring launch per layer
    static const bool chain_env = mi355_env("MI355ASR_CHAIN256", 1) != 0;
    auto ring = [&](const float* wp) -> const float* { const auto it = m->ring_of.find(wp); return it == m->ring_of.end() ? nullptr : it->second; };
    const float* cr[6] = {ring(w.ff_w1p[0]), ring(w.ff_w2p[0]), ring(w.ff_w1p[1]), ring(w.ff_w2p[1]), ring(w.pc_w1p), ring(w.pw2_wp)};
    const bool chain256 = m->cfg.gemm_dtype == 1 && d == 256 && chain_env && !cross && cr[0] && cr[1] && cr[2] && cr[3] && cr[4] && cr[5];
    auto ffn = [&](int i, const float* x, float* y, const float* fg, const float* fb) -> int {
      if (chain256) {
        Chain2Args ca{};
        ca.x = x; ca.res = x; ca.y = y; ca.ln_g = w.ff_ln_g[i]; ca.ln_b = w.ff_ln_b[i];
        ca.w1p = cr[2 * i]; ca.b1 = w.ff_b1[i]; ca.w2p = cr[2 * i + 1]; ca.b2 = w.ff_b2[i];
        ca.fln_g = fg; ca.fln_b = fb; ca.scale = fc; ca.eps = kLnEps; ca.M = M;
        PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_chain256_bf16(0, ca, s), "ff module");
        return 0;
      }
      Gemm16Args a1 = g16(x, d, d, w.ff_w1p[i], w.ff_b1[i], 4 * d / 16, sc.h4, 4 * d);
      a1.ln_g = w.ff_ln_g[i]; a1.ln_b = w.ff_ln_b[i];
      { PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_gemm16(m, E16_SWISH, true, a1, w.ff_w1p[i], s), "ffn1"); }
      Gemm16Args a2 = g16(sc.h4, 4 * d, 4 * d, w.ff_w2p[i], w.ff_b2[i], d / 16, y, d);
      a2.res = x; a2.scale = fc; a2.fln_g = fg; a2.fln_b = fb;
      { PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_gemm16(m, E16_RES, false, a2, w.ff_w2p[i], s), "ffn2"); }
      return 0;
    };
    int rc = ffn(0, sc.xa, sc.xb, nullptr, nullptr);
    if (rc) return rc;
    AttnArgs at{};
    at.ctx = sc.ctx; at.B = B; at.Tq = T; at.H = H; at.D = d;
    at.win_front = bo.win_front; at.win_back = bo.win_back;
    if (cross) {
      // RBlock: q = (LN(xb + PE) Wq) / sqrt(hs) ; [k | v] = enc [Wk | Wv]
      AddPeArgs pa{sc.xb, cross->pe, sc.u, B, T, d};
      { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_add_pe(pa, s), "positional encoding"); }
      Gemm16Args q = g16(sc.u, d, d, w.xq_wp, w.qkv_b, d / 16, sc.qkv, d);
      q.ln_g = w.att_ln_g; q.ln_b = w.att_ln_b; q.qscale = 1.0f / std::sqrt((float)hs); q.qtiles = d / 16;
      { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm16(m, E16_QKV, true, q, w.xq_wp, s), "cross-attention query projection"); }
      Gemm16Args kv = g16(cross->enc, d, d, w.xkv_wp, w.qkv_b, 2 * d / 16, cross->kv, 2 * d);
      kv.M = B * cross->T_enc;
      { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm16(m, E16_BIAS, false, kv, w.xkv_wp, s), "cross-attention key/value projection"); }
      at.q = sc.qkv; at.ldq = d; at.k = cross->kv; at.v = cross->kv + d; at.ldk = 2 * d; at.Tk = cross->T_enc;
    } else {
      Gemm16Args q = g16(sc.xb, d, d, w.qkv_wp, w.qkv_b, 3 * d / 16, sc.qkv, 3 * d);
      q.ln_g = w.att_ln_g; q.ln_b = w.att_ln_b; q.qscale = 1.0f / std::sqrt((float)hs); q.qtiles = d / 16;
      { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm16(m, E16_QKV, true, q, w.qkv_wp, s), "qkv"); }
      at.q = sc.qkv; at.k = sc.qkv + d; at.v = sc.qkv + 2 * d; at.ldq = 3 * d; at.ldk = 3 * d; at.Tk = T;
      at.h2_sq = w.att_h2[0]; at.h2_sk = w.att_h2[1]; at.h2_sv = w.att_h2[2];      // q / k / v are the block's own projections (0: no bound known)
    }
    { PROF(MI355ASR_K_ATTN); LAUNCH_TRY(launch_attention(hs, at, s), "attention"); }
    Gemm16Args op = g16(sc.ctx, d, d, w.out_wp, w.out_b, d / 16, sc.xa, d);
    op.res = sc.xb;
    { PROF(MI355ASR_K_ATTN_OUT); LAUNCH_TRY(launch_gemm16(m, E16_RES, false, op, w.out_wp, s), "attention out"); }
    Gemm16Args gl = g16(sc.xa, d, d, w.pw1_wp, w.pw1_b, 2 * d / 16, sc.u, d);
    gl.ln_g = w.cv_ln_g; gl.ln_b = w.cv_ln_b; gl.n_valid = d;
    { PROF(MI355ASR_K_PW1_GLU); LAUNCH_TRY(launch_gemm16(m, E16_GLU, true, gl, w.pw1_wp, s), "pw_conv_1 + GLU"); }
    DwArgs dwa{};
    dwa.u = sc.u; dwa.y = sc.dw; dwa.wd = w.dw_w; dwa.B = B; dwa.T = T; dwa.D = d;
    dwa.pad_left = bo.causal ? ksz - 1 : (ksz - 1) / 2;
    { PROF(MI355ASR_K_DWCONV); LAUNCH_TRY(launch_dwconv(ksz, dwa, s), "depthwise conv"); }
    if (chain256) {
      Chain2Args ca{};
      ca.x = sc.dw; ca.res = sc.xa; ca.y = sc.xb; ca.w1p = cr[4]; ca.b1 = w.pc_b1; ca.aff_s = w.bn_s; ca.aff_t = w.bn_t;
      ca.w2p = cr[5]; ca.b2 = w.pw2_b; ca.scale = 1.0f; ca.eps = kLnEps; ca.M = M;
      { PROF(MI355ASR_K_CONV_TAIL); LAUNCH_TRY(launch_chain256_bf16(1, ca, s), "conv module tail"); }
      return ffn(1, sc.xb, out ? out : sc.xa, w.ln_g, w.ln_b);
    }
    Gemm16Args pc = g16(sc.dw, d, d, w.pc_w1p, w.pc_b1, 2 * d / 16, sc.h4, 2 * d);
    pc.aff_s = w.bn_s; pc.aff_t = w.bn_t;
    { PROF(MI355ASR_K_CONV_TAIL); LAUNCH_TRY(launch_gemm16(m, E16_AFFSWISH, false, pc, w.pc_w1p, s), "pointwise + BN + swish"); }
    Gemm16Args p2 = g16(sc.h4, 2 * d, 2 * d, w.pw2_wp, w.pw2_b, d / 16, sc.xb, d);
    p2.res = sc.xa;
    { PROF(MI355ASR_K_CONV_TAIL); LAUNCH_TRY(launch_gemm16(m, E16_RES, false, p2, w.pw2_wp, s), "pw_conv_2"); }
    return ffn(1, sc.xb, out ? out : sc.xa, w.ln_g, w.ln_b);
  }
  // round 6: the Translator's RBlock takes the fused kernels too -- its query projection (of LayerNorm(x1 + PE)) rides in the
  // ff_module_1 launch of the pair-pipelined kernel, keys / values come from the encoder output through their own projection
  const bool fused_cross = cross && ff1_qkv_pp_selected(w.ff1_slabs != nullptr, w.pp_ff1 != nullptr) && !bo.pre_pp && !skip_ff1 && !next;
  if (d == 144 && fused_env && (!cross || fused_cross)) {
    // token-local runs of layers in one launch each (fused.hip); attention and the depthwise conv mix tokens
    const float qscale = 1.0f / std::sqrt((float)hs);
    // round 5: q / k / v of a block travel head-major ([B, H, T, 36] planes) whenever their producer is a pair-pipelined kernel and
    // their consumer the two-term attention_split_kernel -- a pure function of the shapes, the switches and the block's weights, so
    // the producer (this block's own ff_module_1 launch, or the previous block's tail) and the consumer agree without a flag
    // being passed between launches.  MI355ASR_QKV_HEAD_MAJOR=0: token-major rows as before.
    auto attn_args = [&](const BlockDev& bw, bool hm) {
      AttnArgs at{};
      at.q = sc.qkv; at.k = sc.qkv + (hm ? (size_t)M * d : (size_t)d); at.v = sc.qkv + (hm ? 2 * (size_t)M * d : 2 * (size_t)d); at.ctx = sc.ctx;
      at.B = B; at.Tq = T; at.Tk = T; at.H = H; at.D = d; at.ldq = hm ? hs : 3 * d; at.ldk = hm ? hs : 3 * d;
      at.win_front = bo.win_front; at.win_back = bo.win_back;
      at.h2_sq = bw.att_h2[0]; at.h2_sk = bw.att_h2[1]; at.h2_sv = bw.att_h2[2];      // q / k / v are the block's own projections
      at.head_major = hm ? 1 : 0;
      return at;
    };
    auto qkv_head_major = [&](const BlockDev& bw) {
      static const bool on = mi355_env("MI355ASR_QKV_HEAD_MAJOR", 1) != 0;
      return on && !cross && ff1_qkv_pp_selected(bw.ff1_slabs != nullptr, bw.pp_ff1 != nullptr) && attention_takes_head_major(hs, attn_args(bw, true));
    };
    auto ff1_args = [&](const BlockDev& bw, const float* x0, float* x1) {
      Ff1QkvArgs k1{};
      k1.x0 = x0; k1.x1 = x1; k1.qkv = sc.qkv;
      k1.ff_ln_g = bw.ff_ln_g[0]; k1.ff_ln_b = bw.ff_ln_b[0]; k1.ff_w1p = bw.ff_w1p[0]; k1.ff_b1 = bw.ff_b1[0];
      k1.ff_w2p = bw.ff_w2p[0]; k1.ff_b2 = bw.ff_b2[0];
      k1.att_ln_g = bw.att_ln_g; k1.att_ln_b = bw.att_ln_b; k1.qkv_wp = bw.qkv_wp; k1.qkv_b = bw.qkv_b;
      k1.fc = fc; k1.qscale = qscale; k1.eps = kLnEps; k1.M = M; k1.slabs = bw.ff1_slabs; k1.pp_slabs = bw.pp_ff1; k1.pp_sc = bw.pp_ff1_sc; k1.pp_sw_qkv = bw.pp_sw_qkv;
      k1.ns_w1 = bw.ns_ff1_w1; k1.ns_w2 = bw.ns_ff1_w2; k1.ns_qkv = bw.ns_qkv;
      if (qkv_head_major(bw)) { k1.qkv_T = T; k1.qkv_H = H; }
      return k1;
    };
    if (!skip_ff1) {
      Ff1QkvArgs k1 = ff1_args(w, sc.xa, sc.xb);
      if (bo.pre_pp) { k1.pre_x = bo.pre_x; k1.pre_pp = bo.pre_pp; k1.pre_sw = bo.pre_sw; k1.pre_chunks = bo.pre_chunks; }
      if (cross) { k1.xq_pe = cross->pe; k1.xq_U = T; }
      PROF(MI355ASR_K_FF1_QKV); LAUNCH_TRY(launch_ff1_qkv(k1, s), "ff_module_1 + qkv");
    }
    AttnArgs at = attn_args(w, qkv_head_major(w));
    if (cross) {
      // [k | v] = enc [Wk | Wv]  [B * T_enc, 2 d]; q sits at columns 0..143 of the [M, 3 d] rows the ff_module_1 launch wrote
      GemmArgs kv{};
      kv.x = cross->enc; kv.y = cross->kv; kv.wp = w.xkv_wp; kv.bias = w.qkv_b;   // qkv_b: 3d zeros (no bias)
      kv.M = B * cross->T_enc; kv.NT = 2 * d / 16; kv.ldy = 2 * d; kv.n_valid = 2 * d; kv.eps = kLnEps;
      { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm_rows(d, EPI_BIAS, false, kv, s), "cross-attention key/value projection"); }
      at.q = sc.qkv; at.ldq = 3 * d; at.k = cross->kv; at.v = cross->kv + d; at.ldk = 2 * d; at.Tk = cross->T_enc; at.head_major = 0;
      at.h2_sq = 0.f; at.h2_sk = 0.f; at.h2_sv = 0.f;          // no operand bounds for the encoder's rows: three exact terms
    }
    OutGluArgs k2{};
    k2.ctx = sc.ctx; k2.x1 = sc.xb; k2.x2 = sc.xa; k2.u = sc.u;
    k2.out_wp = w.out_wp; k2.out_b = w.out_b; k2.cv_ln_g = w.cv_ln_g; k2.cv_ln_b = w.cv_ln_b;
    k2.pw1_wp = w.pw1_wp; k2.pw1_b = w.pw1_b; k2.eps = kLnEps; k2.M = M;
    k2.og_slabs = w.og_slabs; k2.pp_slabs = w.pp_og; k2.pp_sw_out = w.pp_sw_out; k2.pp_sw_pw1 = w.pp_sw_pw1;
    k2.ns_out = w.ns_out; k2.ns_pw1 = w.ns_pw1;
    DwArgs dwa{};
    dwa.u = sc.u; dwa.y = sc.dw; dwa.wd = w.dw_w; dwa.B = B; dwa.T = T; dwa.D = d;
    dwa.pad_left = bo.causal ? ksz - 1 : (ksz - 1) / 2;
    // round 3: the depthwise conv rides in the prologue of the pair-pipelined tail kernel (no launch, dw never in HBM)
    const bool dw_fold = w.pp_tail && w.tail_slabs && tail_pp_selected() && pp_dw_fold_ok(T, ksz);
    TailFf2Args k4{};
    if (dw_fold) { k4.dw_u = sc.u; k4.dw_wd = w.dw_w; k4.dw_T = T; k4.dw_pad = dwa.pad_left; }
    k4.M = M; k4.pp_slabs = w.pp_tail;
    // round 4: so does out-projection + GLU (x2 and u never in HBM either): the block is attention + one launch
    const bool og_fold = dw_fold && pp_og_fold_ok(k4, k2);
    k4.dw = sc.dw; k4.x2 = sc.xa; k4.y = out ? out : sc.xb;
    k4.pc_w1p = w.pc_w1p; k4.pc_b1 = w.pc_b1; k4.bn_s = w.bn_s; k4.bn_t = w.bn_t; k4.pw2_wp = w.pw2_wp; k4.pw2_b = w.pw2_b;
    k4.ff_ln_g = w.ff_ln_g[1]; k4.ff_ln_b = w.ff_ln_b[1]; k4.ff_w1p = w.ff_w1p[1]; k4.ff_b1 = w.ff_b1[1];
    k4.ff_w2p = w.ff_w2p[1]; k4.ff_b2 = w.ff_b2[1]; k4.ln_g = w.ln_g; k4.ln_b = w.ln_b;
    k4.fc = fc; k4.eps = kLnEps; k4.M = M; k4.slabs = w.tail_slabs; k4.pp_slabs = w.pp_tail; k4.pp_sc[0] = w.pp_tail_sc[0]; k4.pp_sc[1] = w.pp_tail_sc[1];
    k4.ns_cv_w1 = w.ns_cv_w1; k4.ns_cv_w2 = w.ns_cv_w2; k4.ns_ff_w1 = w.ns_ff2_w1; k4.ns_ff_w2 = w.ns_ff2_w2;
    // The folded launches (OGF) read x1 from sc.xb, INCLUDING the halo frames of the neighbouring workgroups (the window of
    // the depthwise conv), so nothing in such a launch may write sc.xb: a workgroup that starts after its neighbour has
    // stored the block output there would read y as x1 (grids above one workgroup per CU).  x2 is never materialised in
    // that mode, so sc.xa is free: the block output goes there and the xa/xb swap is skipped.
    TailFf2Args k4og = k4;
    if (!out) k4og.y = sc.xa;
    const bool tail_ff1_case = next && !out && ff1_done && tail_ff1_available() && k4.slabs && next->ff1_slabs;
    // round 6, small batches (up to MI355ASR_NS1_MAX_M rows): one 16-token tile per workgroup (fused_ns.hip).  The block runs as
    // [attention + out-projection + GLU] -> [depthwise conv + tail (+ the next block's ff_module_1 + qkv)]: two launches, the first
    // of which takes the attention along when it can (at most 256 frames, operand bounds known).  Same buffer roles as the folded
    // pair-pipelined launches below: x2 / u go to sc.xa / sc.u, which are free in that mode; every workgroup reads and writes its
    // own rows of sc.xa only.
    // (its own shape rule: 16-frame tiles waste little at any length, so the 64-frame criterion of the folds below does not apply)
    if (w.pp_tail && w.tail_slabs && tail_pp_selected() && pp_enabled() && ksz == 32 && ns1_rows_ok(M)) {
      const bool head_fold_case = og_fold && !tail_ff1_case && bo.head && bo.head_pp && bo.head_done && pp_head_fold_ok(M, bo.head->n_valid, bo.head_groups);
      TailFf2Args kt = tail_ff1_case ? k4 : k4og;
      kt.dw_u = sc.u; kt.dw_wd = w.dw_w; kt.dw_T = T; kt.dw_pad = dwa.pad_left;
      if (tail_ff1_case) kt.y = nullptr;
      const Ff1QkvArgs kn = tail_ff1_case ? ff1_args(*next, nullptr, sc.xa) : Ff1QkvArgs{};
      if (!head_fold_case && ns1_block_ok(kt, tail_ff1_case ? &kn : nullptr, k2)) {
        OutGluArgs kg = k2;
        const bool fuse_attn = ns1_attn_ok(hs, at);
        if (fuse_attn) {
          kg.attn = 1; kg.aq = at.q; kg.ak = at.k; kg.av = at.v; kg.a_T = at.Tk; kg.a_H = at.H; kg.a_ldq = at.ldq; kg.a_ldk = at.ldk;
          kg.a_head_major = at.head_major; kg.a_sq = at.h2_sq; kg.a_sk = at.h2_sk; kg.a_sv = at.h2_sv;
        } else {
          PROF(MI355ASR_K_ATTN); LAUNCH_TRY(launch_attention(hs, at, s), "attention");
        }
        if (tail_ff1_case) {
          PROF(MI355ASR_K_TAIL_FF1);
          LAUNCH_TRY(launch_ns1_og_tail(kt, &kn, kg, s), "small-batch block + next ff_module_1");
          *ff1_done = true;
          std::swap(sc.xa, sc.xb);
        } else {
          PROF(MI355ASR_K_TAIL_FF2);
          LAUNCH_TRY(launch_ns1_og_tail(kt, nullptr, kg, s), "small-batch block");      // y is in sc.xa (or `out`): no swap
        }
        return 0;
      }
    }
    { PROF(MI355ASR_K_ATTN); LAUNCH_TRY(launch_attention(hs, at, s), "attention"); }
    if (!og_fold) { PROF(MI355ASR_K_OUT_GLU); LAUNCH_TRY(launch_out_glu(k2, s), "out-projection + GLU"); }
    if (!dw_fold) { PROF(MI355ASR_K_DWCONV); LAUNCH_TRY(launch_dwconv(ksz, dwa, s), "depthwise conv"); }
    // out-projection + GLU as its own launch, once, when a folded launcher declines (it declines before launching anything)
    bool og_pending = og_fold;
    auto unfold = [&]() -> int {
      if (!og_pending) return 0;
      og_pending = false;
      PROF(MI355ASR_K_OUT_GLU); LAUNCH_TRY(launch_out_glu(k2, s), "out-projection + GLU");
      return 0;
    };
    if (tail_ff1_case) {
      // the block output feeds only the next block's ff_module_1: keep it in registers, write x1 (into the buffer the
      // next block knows as sc.xb after the swap below -- this block's x2, which each workgroup has consumed) and qkv
      TailFf2Args kf = k4;
      kf.y = nullptr;
      const Ff1QkvArgs kn = ff1_args(*next, nullptr, sc.xa);
      PROF(MI355ASR_K_TAIL_FF1);
      bool launched = og_fold && launch_pp_og_tail_ff1(kf, kn, k2, s) == 0;
      if (!launched) {
        if (int rc = unfold()) return rc;
        launched = launch_tail_ff1(kf, kn, s) == 0;
      }
      if (launched) {
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return fail(MI355ASR_EHIP, "conv tail + ff_module_2 + next ff_module_1: %s", hipGetErrorString(e_));
        *ff1_done = true;
        std::swap(sc.xa, sc.xb);
        return 0;
      }
    }
    {
      PROF(MI355ASR_K_TAIL_FF2);
      // round 4: the class head behind the CTC decoder's last block rides in this launch (the block output itself is then
      // stored only if somebody asked for it)
      if (bo.head && bo.head_pp && bo.head_done && og_pending && pp_head_fold_ok(M, bo.head->n_valid, bo.head_groups)) {
        TailFf2Args kh = k4og;
        if (!out) kh.y = nullptr;
        kh.head_pp = bo.head_pp; kh.head_sw = bo.head_sw; kh.head_groups = bo.head_groups; kh.head_ldy = bo.head->ldy;
        kh.head_nvalid = bo.head->n_valid; kh.head_y = bo.head->y; kh.head_argmax = bo.head->argmax_out; kh.head_maxval = bo.head->maxval_out;
        if (launch_pp_og_tail_ff2(kh, k2, s) == 0) {
          if (hipGetLastError() != hipSuccess) return fail(MI355ASR_EHIP, "block tail + class head launch failed");
          *bo.head_done = true;
          return 0;                                    // nothing was stored: the block's input stays where it was
        }
      }
      if (og_pending && launch_pp_og_tail_ff2(k4og, k2, s) == 0) {
        if (hipGetLastError() != hipSuccess) return fail(MI355ASR_EHIP, "out-projection + GLU + conv tail + ff_module_2 launch failed");
        return 0;                                      // y is in sc.xa (or `out`): no swap
      }
      if (int rc = unfold()) return rc;
      LAUNCH_TRY(launch_tail_ff2(k4, s), "conv tail + ff_module_2");
    }
    if (!out) std::swap(sc.xa, sc.xb);
    return 0;
  }
  // ff_module_1: xb = xa + fc * FFN(LN(xa))
  Chain2Args f1{};
  f1.x = sc.xa; f1.res = sc.xa; f1.y = sc.xb;
  f1.ln_g = w.ff_ln_g[0]; f1.ln_b = w.ff_ln_b[0];
  f1.w1p = w.ff_w1p[0]; f1.b1 = w.ff_b1[0]; f1.w2p = w.ff_w2p[0]; f1.b2 = w.ff_b2[0];
  f1.scale = fc; f1.eps = kLnEps; f1.M = M;
  { PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_chain2(d, 0, f1, s), "ff_module_1"); }
  AttnArgs at{};
  at.ctx = sc.ctx; at.B = B; at.Tq = T; at.H = H; at.D = d;
  at.win_front = bo.win_front; at.win_back = bo.win_back;
  if (cross) {
    // q = (LN(xb + PE) Wq) / sqrt(hs)  [M, d] ; [k | v] = enc [Wk | Wv]  [B*T_enc, 2d]
    AddPeArgs pa{sc.xb, cross->pe, sc.u, B, T, d};
    { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_add_pe(pa, s), "positional encoding"); }
    GemmArgs q{};
    q.x = sc.u; q.y = sc.qkv; q.ln_g = w.att_ln_g; q.ln_b = w.att_ln_b; q.wp = w.xq_wp; q.bias = w.qkv_b;
    q.M = M; q.NT = d / 16; q.ldy = d; q.n_valid = d; q.eps = kLnEps;
    q.qscale = 1.0f / std::sqrt((float)hs); q.qtiles = d / 16;
    { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm_rows(d, EPI_QKV, true, q, s), "cross-attention query projection"); }
    GemmArgs kv{};
    kv.x = cross->enc; kv.y = cross->kv; kv.wp = w.xkv_wp; kv.bias = w.qkv_b;   // qkv_b: 3d zeros (no bias)
    kv.M = B * cross->T_enc; kv.NT = 2 * d / 16; kv.ldy = 2 * d; kv.n_valid = 2 * d; kv.eps = kLnEps;
    { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm_rows(d, EPI_BIAS, false, kv, s), "cross-attention key/value projection"); }
    at.q = sc.qkv; at.ldq = d; at.k = cross->kv; at.v = cross->kv + d; at.ldk = 2 * d; at.Tk = cross->T_enc;
  } else {
    // mhsa: qkv = LN(xb) Wqkv (q pre-scaled)
    GemmArgs q{};
    q.x = sc.xb; q.y = sc.qkv; q.ln_g = w.att_ln_g; q.ln_b = w.att_ln_b; q.wp = w.qkv_wp; q.bias = w.qkv_b;
    q.M = M; q.NT = 3 * d / 16; q.ldy = 3 * d; q.n_valid = 3 * d; q.eps = kLnEps;
    q.qscale = 1.0f / std::sqrt((float)hs); q.qtiles = d / 16;
    { PROF(MI355ASR_K_QKV); LAUNCH_TRY(launch_gemm_rows(d, EPI_QKV, true, q, s), "qkv projection"); }
    at.q = sc.qkv; at.k = sc.qkv + d; at.v = sc.qkv + 2 * d; at.ldq = 3 * d; at.ldk = 3 * d; at.Tk = T;
  }
  { PROF(MI355ASR_K_ATTN); LAUNCH_TRY(launch_attention(hs, at, s), "attention"); }
  // xa = xb + ctx Wo + bo
  GemmArgs op{};
  op.x = sc.ctx; op.y = sc.xa; op.res = sc.xb; op.wp = w.out_wp; op.bias = w.out_b;
  op.M = M; op.NT = d / 16; op.ldy = d; op.n_valid = d; op.eps = kLnEps;
  { PROF(MI355ASR_K_ATTN_OUT); LAUNCH_TRY(launch_gemm_rows(d, EPI_RESIDUAL, false, op, s), "attention out-projection"); }
  // conv module: u = GLU(LN(xa) Wpw1 + b)
  GemmArgs g{};
  g.x = sc.xa; g.y = sc.u; g.ln_g = w.cv_ln_g; g.ln_b = w.cv_ln_b; g.wp = w.pw1_wp; g.bias = w.pw1_b;
  g.M = M; g.NT = 2 * d / 16; g.ldy = d; g.n_valid = d; g.eps = kLnEps;
  { PROF(MI355ASR_K_PW1_GLU); LAUNCH_TRY(launch_gemm_rows(d, EPI_GLU, true, g, s), "pw_conv_1 + GLU"); }
  DwArgs dwa{};
  dwa.u = sc.u; dwa.y = sc.dw; dwa.wd = w.dw_w; dwa.B = B; dwa.T = T; dwa.D = d;
  // Keras 'same', stride 1: total k-1, before = (k-1)//2 ; 'causal': all k-1 on the left
  dwa.pad_left = bo.causal ? ksz - 1 : (ksz - 1) / 2;
  { PROF(MI355ASR_K_DWCONV); LAUNCH_TRY(launch_dwconv(ksz, dwa, s), "depthwise conv"); }
  // xb = xa + pw2( swish( BN( dw Wpc + bpc ) ) ) + b2
  Chain2Args cv{};
  cv.x = sc.dw; cv.res = sc.xa; cv.y = sc.xb;
  cv.w1p = w.pc_w1p; cv.b1 = w.pc_b1; cv.aff_s = w.bn_s; cv.aff_t = w.bn_t; cv.w2p = w.pw2_wp; cv.b2 = w.pw2_b;
  cv.scale = 1.0f; cv.eps = kLnEps; cv.M = M;
  { PROF(MI355ASR_K_CONV_TAIL); LAUNCH_TRY(launch_chain2(d, 1, cv, s), "conv module tail"); }
  // ff_module_2 + block LayerNorm
  Chain2Args f2{};
  f2.x = sc.xb; f2.res = sc.xb; f2.y = out ? out : sc.xa;
  f2.ln_g = w.ff_ln_g[1]; f2.ln_b = w.ff_ln_b[1];
  f2.w1p = w.ff_w1p[1]; f2.b1 = w.ff_b1[1]; f2.w2p = w.ff_w2p[1]; f2.b2 = w.ff_b2[1];
  f2.fln_g = w.ln_g; f2.fln_b = w.ln_b;
  f2.scale = fc; f2.eps = kLnEps; f2.M = M;
  { PROF(MI355ASR_K_FFN); LAUNCH_TRY(launch_chain2(d, 0, f2, s), "ff_module_2 + LayerNorm"); }
  return 0;
}

int run_mel(const mi355asr_model* m, const float* wav, int Bp, int Lb, int F, float* logp, float* pmax,
            float* umax, float* mel, hipStream_t s) {
  const auto& c = m->cfg;
  if (c.mel_layer_type == 1) {
    // LEAF: Gabor conv + squared modulus + Gaussian pooling (partials in the log-power scratch), then PCEN + instance norm
    int nf, pl;
    same_pad(Lb, 401, m->dm.hop, &nf, &pl);
    // leaf_terms = 0: fp32 MFMA kernel, tiles of 128 positions; 2 / 3: split-bf16 kernel, tiles of 512 (leaf.hip)
    const int tile = m->leaf_terms ? kLeafSplitTile : 128, nrel = m->leaf_terms ? kLeafSplitSlots : 4;
    const int NH = ceil_div(Lb, tile);
    if (m->leaf_terms) {
      LeafConvArgs la{wav, m->leaf_wsplit, m->leaf_gcoef, logp, m->leaf_p0, m->leaf_p1, Bp, Lb, F, NH, m->dm.hop, pl};
      PROF(MI355ASR_K_STFT);
      LAUNCH_TRY(launch_leaf_conv_pool_split(m->leaf_terms, la, s), "leaf gabor conv (split bf16) + pooling");
    } else {
      LeafConvArgs la{wav, m->leaf_wp, m->leaf_gcoef, logp, m->leaf_p0, m->leaf_p1, Bp, Lb, F, NH, m->dm.hop, pl};
      PROF(MI355ASR_K_STFT);
      LAUNCH_TRY(launch_leaf_conv_pool(la, s), "leaf gabor conv + pooling");
    }
    LeafPcenArgs lp{logp, m->leaf_alpha, m->leaf_delta, m->leaf_root, m->leaf_smooth, m->leaf_gamma, m->leaf_beta, mel,
                    Bp, F, NH, m->dm.hop, pl, tile, nrel};
    { PROF(MI355ASR_K_MEL); LAUNCH_TRY(launch_leaf_pcen_norm(lp, s), "leaf PCEN + instance norm"); }
    return 0;
  }
  const int FT = ceil_div(F, 16);
  int out, before;
  same_pad(Lb, c.n_dft, m->dm.hop, &out, &before);
  StftArgs st{};
  st.wav = wav; st.logp = logp; st.pmax = pmax; st.wp = m->dft_wp;
  st.B = Bp; st.L = Lb; st.F = F; st.hop = m->dm.hop; st.pad_left = before; st.n_dft = c.n_dft;
  st.NT = m->dm.NT_dft; st.LP = m->dm.LP; st.nbins = m->dm.nbins; st.FT = FT; st.NCH = m->dm.NCH_dft;
  st.db10 = 1;
  int npart = FT * m->dm.NCH_dft;
  if (m->fft_ok) {
    FftStftArgs fa{wav, logp, pmax, m->fft_w1p, m->fft_w2p, m->fft_twc, m->fft_tws, m->fft_win,
                   Bp, Lb, F, m->dm.hop, before, m->dm.LP, 1};
    fa.w1s = m->fft_w1s; fa.w2s = m->fft_w2s; fa.w1h = m->fft_w1h; fa.w2h = m->fft_w2h;
    { PROF(MI355ASR_K_STFT); LAUNCH_TRY(launch_fft_stft(fa, s), "stft (fft)"); }
    npart = F;
  } else {
    PROF(MI355ASR_K_STFT);
    LAUNCH_TRY(launch_stft(st, s), "stft");
  }
  UttMaxArgs um{pmax, umax, npart};
  { PROF(MI355ASR_K_UTT_MAX); LAUNCH_TRY(launch_utt_max(um, Bp, s), "utterance max"); }
  MelArgs me{};
  me.logp = logp; me.umax = umax; me.mel = mel; me.wp = m->mel_wp;
  me.B = Bp; me.F = F; me.LP = m->dm.LP; me.nbins = m->dm.nbins; me.KBm = m->dm.KBm; me.NTm = m->dm.NTm;
  me.NM = c.n_mels; me.FT = FT; me.floor_db = -80.0f;
  if (c.mel_layer_type == 2) { PROF(MI355ASR_K_MEL); LAUNCH_TRY(launch_db_norm(me, s), "dB (Spectrogram layer)"); }
  else { PROF(MI355ASR_K_MEL); LAUNCH_TRY(launch_mel_auto(m, me, s), "dB + mel"); }
  return 0;
}

// mel_bounded: the features come from this handle's own frontend (|mel| <= 80 x the filters' L1 norm), which the two-term
// fp16 kernel's operand scale relies on; features handed in by the caller take the three-term bf16 kernel
int run_subsampling(const mi355asr_model* m, const float* mel, int Bp, int F, float* sub, float* out,
                    hipStream_t s, bool mel_bounded, bool* defer_dense) {
  if (defer_dense) *defer_dense = false;
  const auto& c = m->cfg;
  const int d = c.dmodel;
  int T1, pt1, T2, pt2;
  same_pad(F, 3, m->dm.st1, &T1, &pt1);
  same_pad(T1, 3, 2, &T2, &pt2);
  SubConvArgs sa{};
  sa.mel = mel; sa.out = sub; sa.w1 = m->c1_w; sa.b1 = m->c1_b; sa.w2p = m->c2_wp; sa.b2 = m->c2_b; sa.w2s = m->c2_wsplit;
  static const bool force_half = mi355_env("MI355ASR_SUBCONV_TERMS", -1) == 22;   // 22: also for caller-supplied features (tests)
  if (m->c2_whalf && (mel_bounded || force_half)) { sa.w2h = m->c2_whalf; sa.h_scale = m->c2_hscale; sa.h_wscale = m->c2_wscale; }
  // conv1 on the matrix pipe needs the frontend's own bound on |mel| for its fp16 planes: not for caller-supplied features
  if (sa.w2h && mel_bounded) { sa.c1_mscale = m->c1_mscale; sa.c1_wscale = m->c1_wscale; }
  sa.B = Bp; sa.F = F; sa.NM = c.n_mels; sa.T1 = T1; sa.F1 = m->dm.F1; sa.T2 = T2; sa.F2 = m->dm.F2;
  sa.st1 = m->dm.st1; sa.pt1 = pt1; sa.pf1 = m->dm.pf1; sa.pt2 = pt2; sa.pf2 = m->dm.pf2;
  { PROF(MI355ASR_K_SUBCONV); LAUNCH_TRY(launch_subconv(d, sa, s), "conv subsampling"); }
  if (use_gemm16(m)) {
    Gemm16Args lg{};
    lg.x = sub; lg.ldx = m->dm.F2 * d; lg.bias = m->lin_b; lg.y = out; lg.ldy = d;
    lg.M = Bp * T2; lg.K = m->dm.F2 * d; lg.NT = d / 16; lg.n_valid = d; lg.eps = kLnEps;
    { PROF(MI355ASR_K_SUBLINEAR); LAUNCH_TRY(launch_gemm16(m, E16_BIAS, false, lg, m->lin_wp, s), "subsampling linear"); }
    return 0;
  }
  StreamGemmArgs lg{};
  lg.x = sub; lg.y = out; lg.wp = m->lin_wp; lg.bias = m->lin_b;
  lg.M = Bp * T2; lg.K = m->dm.F2 * d; lg.NT = d / 16; lg.ldy = d; lg.n_valid = d;
  // MI355ASR_SUBLINEAR_SPLIT: 1 (default) = split-bf16 ring-DMA kernel (fused.hip) from 4096 rows, 2 = for any row
  // count, 0 = the fp32-MFMA stream_gemm_kernel
  static const int lin_split = (int)mi355_env("MI355ASR_SUBLINEAR_SPLIT", 1);
  if (lin_split && m->lin_wsplit && (lg.M >= 4096 || lin_split == 2)) {
    if (defer_dense && m->lin_pp && pp_sublinear_ok(lg, m->lin_pp)) { *defer_dense = true; return 0; }   // the caller folds it into the first block
    PROF(MI355ASR_K_SUBLINEAR);
    // round 4: two fp16 terms with a scale per (token, 144-wide chunk) -- no bound on the operand needed, so caller-supplied
    // features take it too; the three-term kernel behind MI355ASR_PP_SUBLINEAR=0 / MI355ASR_PP=0
    if (m->lin_pp && launch_pp_sublinear(lg, m->lin_pp, m->lin_pp_sw, s) == 0) return 0;
    if (launch_sublinear_split(lg, m->lin_wsplit, s) == 0) return 0;
  }
  {
    PROF(MI355ASR_K_SUBLINEAR);
    // round 6, small batches (the fp32 stream_gemm kernel took 38 us for one utterance): one 16-token tile per workgroup, the 144-wide
    // chunks of a row split over its eight waves, on the two-term fragments of pp_sublinear_kernel (MI355ASR_SUBLINEAR_SPLIT=0: off)
    if (lin_split && m->lin_ns && pp_sublinear_ok(lg, m->lin_pp) && launch_ns1_sublinear(lg, m->lin_ns, m->lin_pp_sw, s) == 0) return 0;
    if (launch_stream_gemm(d, lg, s) == 0) return 0;
  }
  // K = F2 * dmodel that is not a multiple of 32 (the plain Spectrogram layer: F2 = 129): the layer-at-a-time kernel
  Gemm16Args g16{};
  g16.x = sub; g16.ldx = m->dm.F2 * d; g16.bias = m->lin_b; g16.y = out; g16.ldy = d;
  g16.M = Bp * T2; g16.K = m->dm.F2 * d; g16.NT = d / 16; g16.n_valid = d; g16.eps = kLnEps;
  { PROF(MI355ASR_K_SUBLINEAR); LAUNCH_TRY(launch_gemm16(m, E16_BIAS, false, g16, m->lin_wp, s), "subsampling linear"); }
  return 0;
}


int check_ready(const mi355asr_model* m, bool need_encoder = false) {
  if (!m) return fail(MI355ASR_EINVAL, "null model handle");
  if (!m->finalized) return fail(MI355ASR_ESTATE, "weights not finalised: call mi355asr_finalize_weights first");
  if (m->is_chunk) return fail(MI355ASR_ESTATE, "ChunkConformer handle: use mi355asr_chunk_predict");
  if (m->is_translator) return fail(MI355ASR_ESTATE, "Translator handle: use mi355asr_translator_forward");
  if (need_encoder && !m->cfg.has_encoder) return fail(MI355ASR_ESTATE, "model was created without an encoder (has_encoder=0)");
  return 0;
}


// WavePickModel.get_scales (wav_model.py:132-146): prime factors of hop_size merged down to four strides, descending
std::vector<int> wave_pick_scales(int num) {
  std::vector<int> sc;
  while (num > 1) {
    int i = 2;
    while (i < 100 && num % i != 0) ++i;
    if (i >= 100) return {};
    num /= i;
    sc.push_back(i);
  }
  while (sc.size() > 4) {
    std::vector<int> ns(sc.begin() + 2, sc.end());
    ns.push_back(sc[0] * sc[1]);
    std::sort(ns.begin(), ns.end());
    sc = ns;
  }
  std::reverse(sc.begin(), sc.end());
  return sc;
}

// floats of scratch the add_wav_info branch needs for Bp inputs of at most Lmax samples
size_t wavpick_floats(const mi355asr_model* m, int Bp, int Lmax) {
  if (!m->cfg.add_wav_info || m->wp_stride0 == 0) return 0;
  const size_t T0 = ceil_div(Lmax, m->wp_stride0);
  size_t s1 = 0, t = T0;
  for (const auto& st : m->wp_stages) { t = ceil_div((int)t, st.stride); s1 = std::max(s1, (t + 8) * (size_t)st.c); }
  return (size_t)Bp * ((T0 + 8) * 32 * 2 + 4 * s1) + 1024;
}

// xa[Bp*T, d] += WavePickModel(wav)  (conformer_blocks.py:344-348); every Conv1D is a GEMM over overlapping rows of a
// padded channels-last copy of its input (wavpick.hip)
int run_wavpick(const mi355asr_model* m, const float* wav, int Bp, int Lb, int T, float* xa, float* wv, hipStream_t s) {
  const int d = m->cfg.dmodel;
  const float slope = 0.3f;                          // tf.keras.layers.LeakyReLU() default
  const int T0 = ceil_div(Lb, m->wp_stride0);
  size_t s1 = 0;
  { int t = T0; for (const auto& st : m->wp_stages) { t = ceil_div(t, st.stride); s1 = std::max(s1, (size_t)(t + 8) * st.c); } }
  float* bufA = wv;
  float* bufP = bufA + (size_t)Bp * (T0 + 8) * 32;
  float* bufY = bufP + (size_t)Bp * (T0 + 8) * 32;
  float* bufH = bufY + (size_t)Bp * s1;
  float* bufG = bufH + (size_t)Bp * s1;
  float* bufS = bufG + (size_t)Bp * s1;
  int out0, pl0;
  same_pad(Lb, 7, m->wp_stride0, &out0, &pl0);
  WpSepConvArgs sa{wav, m->wp_dw, m->wp_pw, m->wp_b, bufA, Bp, Lb, T0, m->wp_stride0, pl0, slope};
  LAUNCH_TRY(launch_wp_sepconv(sa, s), "wav_layer separable conv");
  auto conv = [&](const float* xpad, int Tpad, int cin, int k, int stride, int Tout, const float* wp, const float* bias, int cout,
                  float* y, const float* res) -> int {
    Gemm16Args g{};
    g.x = xpad; g.ldx = stride * cin; g.K = k * cin; g.wp = wp; g.bias = bias; g.NT = cout / 16; g.y = y; g.ldy = cout;
    g.M = Bp * Tout; g.n_valid = cout; g.eps = kLnEps; g.scale = 1.0f; g.res = res;
    g.rpb = Tout; g.bstride = (long long)Tpad * cin;
    LAUNCH_TRY(launch_gemm16_f32(res ? E16_RES : E16_BIAS, false, g, s), "wav_layer conv1d");
    return 0;
  };
  const float *cur = bufA, *cur2 = nullptr;
  int Tc = T0, cin = 32;
  for (const auto& st : m->wp_stages) {
    int Tn, lo;
    same_pad(Tc, 3, st.stride, &Tn, &lo);
    const int hi = std::max((Tn - 1) * st.stride + 3 - Tc, 0) - lo;
    WpPadActArgs pz{cur, cur2, bufP, Bp, Tc, cin, lo, hi, 0, 1.0f};
    LAUNCH_TRY(launch_wp_pad_act(pz, s), "wav_layer zero pad");
    int rc = conv(bufP, Tc + lo + hi, cin, 3, st.stride, Tn, st.cw, st.cb, st.c, bufY, nullptr);
    if (rc) return rc;
    // TFResidualStack (wav_model.py:57-104)
    WpPadActArgs pr{bufY, nullptr, bufP, Bp, Tn, st.c, 2, 2, 1, slope};
    LAUNCH_TRY(launch_wp_pad_act(pr, s), "wav_layer reflect pad + LeakyReLU");
    rc = conv(bufP, Tn + 4, st.c, 5, 1, Tn, st.w5, st.b5, st.c, bufH, nullptr);
    if (rc) return rc;
    WpPadActArgs pg{bufH, nullptr, bufG, Bp, Tn, st.c, 0, 0, 0, slope};
    LAUNCH_TRY(launch_wp_pad_act(pg, s), "wav_layer LeakyReLU");
    rc = conv(bufY, Tn, st.c, 1, 1, Tn, st.ws, st.bs, st.c, bufS, nullptr);
    if (rc) return rc;
    rc = conv(bufG, Tn, st.c, 1, 1, Tn, st.w1, st.b1, st.c, bufH, nullptr);
    if (rc) return rc;
    cur = bufS; cur2 = bufH; Tc = Tn; cin = st.c;    // the sum of the two branches is formed by the next padded copy
  }
  if (Tc != T) return fail(MI355ASR_EINVAL, "add_wav_info: the waveform branch yields %d frames, the frontend %d", Tc, T);
  WpPadActArgs pf{cur, cur2, bufP, Bp, Tc, cin, 3, 3, 0, 1.0f};
  LAUNCH_TRY(launch_wp_pad_act(pf, s), "wav_layer zero pad");
  return conv(bufP, Tc + 6, cin, 7, 1, Tc, m->wp_fw, m->wp_fb, d, xa, xa);
}

// the encoder's block stack as stream256_kernel's arguments; false: not its shape (or a ring pack is missing, or switched off)
bool stream256_args(const mi355asr_model* m, int B, int T, const float* x, float* y, S256Args& sa) {
  static const bool on = mi355_env("MI355ASR_STREAM256", 1) != 0;
  const int nb = m->cfg.num_blocks;
  if (!on || m->cfg.gemm_dtype != 1 || m->cfg.dmodel != 256 || m->cfg.num_heads != 4 || m->cfg.head_size != 64 ||
      !stream256_shape_ok(B, T, nb, m->cfg.kernel_size) || (int)m->enc_blocks.size() < nb)
    return false;
  auto ring = [&](const float* wp) -> const void* { const auto it = m->ring_of.find(wp); return it == m->ring_of.end() ? nullptr : it->second; };
  sa.x = x; sa.y = y; sa.B = B; sa.T = T; sa.nblocks = nb; sa.ksz = m->cfg.kernel_size; sa.pad_left = (m->cfg.kernel_size - 1) / 2;
  sa.fc = m->cfg.fc_factor; sa.qscale = 1.0f / std::sqrt((float)m->cfg.head_size); sa.eps = kLnEps;
  for (int i = 0; i < nb; ++i) {
    const BlockDev& w = m->enc_blocks[i];
    S256Block& b = sa.blk[i];
    for (int k = 0; k < 2; ++k) {
      b.ff_ln_g[k] = w.ff_ln_g[k]; b.ff_ln_b[k] = w.ff_ln_b[k]; b.ff_b1[k] = w.ff_b1[k]; b.ff_b2[k] = w.ff_b2[k];
      b.ff_w1[k] = ring(w.ff_w1p[k]); b.ff_w2[k] = ring(w.ff_w2p[k]);
      if (!b.ff_w1[k] || !b.ff_w2[k]) return false;
    }
    b.att_ln_g = w.att_ln_g; b.att_ln_b = w.att_ln_b; b.qkv_b = w.qkv_b; b.out_b = w.out_b;
    b.qkv_w = ring(w.qkv_wp); b.out_w = ring(w.out_wp);
    b.cv_ln_g = w.cv_ln_g; b.cv_ln_b = w.cv_ln_b; b.pw1_b = w.pw1_b; b.dw_w = w.dw_w; b.pc_b1 = w.pc_b1; b.bn_s = w.bn_s; b.bn_t = w.bn_t;
    b.pw2_b = w.pw2_b;
    b.pw1_w = ring(w.pw1_wp); b.pc_w1 = ring(w.pc_w1p); b.pw2_w = ring(w.pw2_wp);
    b.ln_g = w.ln_g; b.ln_b = w.ln_b;
    if (!b.qkv_w || !b.out_w || !b.pw1_w || !b.pc_w1 || !b.pw2_w || w.xq_wp) return false;
  }
  return true;
}

int encoder_impl(mi355asr_model* m, const float* wav, const Geometry& g, const Plan& p, char* ws, float* enc_out,
                 hipStream_t s) {
  float* logp = (float*)(ws + p.logp);
  float* mel = (float*)(ws + p.mel);
  int rc = run_mel(m, wav, g.Bp, g.Lb, g.F, logp, (float*)(ws + p.pmax), (float*)(ws + p.umax), mel, s);
  if (rc) return rc;
  Scratch sc{(float*)(ws + p.xa), (float*)(ws + p.xb), (float*)(ws + p.qkv),
             (float*)(ws + p.ctx), (float*)(ws + p.u), (float*)(ws + p.dw)};
  sc.h4 = (float*)(ws + p.h4);
  // round 4: the subsampling Dense rides in the first block's ff_module_1 + qkv launch when both run on the two-term stream
  const int nb = m->cfg.num_blocks;
  bool dense_deferred = false;
  const bool may_defer = nb > 0 && !m->cfg.add_wav_info && block_takes_pre(m, m->enc_blocks[0], (size_t)g.Bp * g.T);
  rc = run_subsampling(m, mel, g.Bp, g.F, (float*)(ws + p.sub), sc.xa, s, true, may_defer ? &dense_deferred : nullptr);
  if (rc) return rc;
  if (m->cfg.add_wav_info) {
    rc = run_wavpick(m, wav, g.Bp, g.Lb, g.T, sc.xa, (float*)(ws + p.wv), s);
    if (rc) return rc;
  }
  // round 5: the streaming shapes (bf16 mode, dmodel 256, chunks of <= 16 rows): the whole block stack as ONE launch, one workgroup
  // per chunk (stream256.hip) -- MI355ASR_STREAM256=0: one launch per layer / module as before
  {
    S256Args sa{};
    if (stream256_args(m, g.Bp, g.T, sc.xa, enc_out, sa)) {
      PROF(MI355ASR_K_ENC_STACK);
      if (launch_stream256(sa, s) == 0) {          // -1 (not its shape after all): the per-layer loop below runs instead
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MI355ASR_EHIP, "launch encoder block stack: %s", hipGetErrorString(e));
        return 0;
      }
    }
  }
  bool ff1_done = false;
  for (int i = 0; i < nb; ++i) {
    BlockOpts bo;
    bo.ksz = m->cfg.kernel_size;
    bo.fc = m->cfg.fc_factor;
    if (i == 0 && dense_deferred) { bo.pre_x = (float*)(ws + p.sub); bo.pre_pp = m->lin_pp; bo.pre_sw = m->lin_pp_sw; bo.pre_chunks = m->dm.F2; }
    const bool skip = ff1_done;
    rc = run_block(m, m->enc_blocks[i], bo, sc, g.Bp, g.T, i == nb - 1 ? enc_out : nullptr, s, nullptr,
                   i + 1 < nb ? &m->enc_blocks[i + 1] : nullptr, &ff1_done, skip);
    if (rc) return rc;
  }
  if (nb == 0) HIP_TRY(hipMemcpyAsync(enc_out, sc.xa, (size_t)g.Bp * g.T * m->cfg.dmodel * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

int ctc_impl(mi355asr_model* m, const float* enc, int B, int T, const Plan& p, char* ws, float* logits,
             int32_t* amax, hipStream_t s) {
  const int d = m->cfg.dmodel;
  const int M = B * T;
  Scratch sc{(float*)(ws + p.xa), (float*)(ws + p.xb), (float*)(ws + p.qkv),
             (float*)(ws + p.ctx), (float*)(ws + p.u), (float*)(ws + p.dw)};
  sc.h4 = (float*)(ws + p.h4);
  const bool bf16 = gemm16_for(m, M);
  // round 4: the projection rides in the first decoder block's ff_module_1 + qkv launch
  const bool proj_fold = m->proj_pp && m->cfg.ctc_num_blocks > 0 && block_takes_pre(m, m->ctc_blocks[0], (size_t)M);
  if (proj_fold) {
    // (nothing here: the first block below computes it)
  } else if (bf16) {
    Gemm16Args pr{};
    pr.x = enc; pr.ldx = d; pr.bias = m->proj_b; pr.y = sc.xa; pr.ldy = d;
    pr.M = M; pr.K = d; pr.NT = d / 16; pr.n_valid = d; pr.eps = kLnEps;
    { PROF(MI355ASR_K_CTC_PROJECT); LAUNCH_TRY(launch_gemm16(m, E16_BIAS, false, pr, m->proj_wp, s), "ctc project"); }
  } else {
    // on its own: the same two-term stream through pp_sublinear_kernel (one chunk) -- bit-identical to the folded form --, else fp32 MFMA
    StreamGemmArgs sp{};
    sp.x = enc; sp.y = sc.xa; sp.M = M; sp.K = d; sp.NT = d / 16; sp.ldy = d; sp.n_valid = d;
    PROF(MI355ASR_K_CTC_PROJECT);
    if (!(m->proj_pp && m->cfg.gemm_dtype == 0 && launch_pp_sublinear(sp, m->proj_pp, m->proj_pp_sw, s) == 0)) {
      GemmArgs pr{};
      pr.x = enc; pr.y = sc.xa; pr.wp = m->proj_wp; pr.bias = m->proj_b;
      pr.M = M; pr.NT = d / 16; pr.ldy = d; pr.n_valid = d; pr.eps = kLnEps;
      LAUNCH_TRY(launch_gemm_rows(d, EPI_BIAS, false, pr, s), "ctc project");
    }
  }
  // round 4: the class head rides in the last block's tail launch where pp_head_kernel would have run (try_head_ld's conditions)
  GemmArgs hdf{};
  hdf.y = logits; hdf.bias = m->fc_b; hdf.M = M; hdf.NT = m->NT_fc; hdf.ldy = m->cfg.num_classes; hdf.n_valid = m->cfg.num_classes; hdf.eps = kLnEps;
  hdf.argmax_out = amax ? amax : (int32_t*)(ws + p.amax);
  bool head_done = false;
  const auto head_it = (!bf16 && m->cfg.gemm_dtype == 0 && M >= 2048) ? m->head_of.find(m->fc_wp) : m->head_of.end();
  for (int i = 0; i < m->cfg.ctc_num_blocks; ++i) {
    BlockOpts bo;
    bo.ksz = m->cfg.ctc_kernel_size;
    bo.fc = m->cfg.ctc_fc_factor;
    if (i == 0 && proj_fold) { bo.pre_x = enc; bo.pre_pp = m->proj_pp; bo.pre_sw = m->proj_pp_sw; bo.pre_chunks = 1; }
    if (i == m->cfg.ctc_num_blocks - 1 && head_it != m->head_of.end() && head_it->second.pp) {
      bo.head = &hdf; bo.head_pp = head_it->second.pp; bo.head_sw = head_it->second.pp_sw; bo.head_groups = head_it->second.groups; bo.head_done = &head_done;
    }
    int rc = run_block(m, m->ctc_blocks[i], bo, sc, B, T, nullptr, s);
    if (rc) return rc;
  }
  if (head_done) return 0;
  if (bf16) {
    Gemm16Args hd{};
    hd.x = sc.xa; hd.ldx = d; hd.bias = m->fc_b; hd.y = logits; hd.ldy = m->cfg.num_classes;
    hd.M = M; hd.K = d; hd.NT = m->NT_fc; hd.n_valid = m->cfg.num_classes; hd.eps = kLnEps;
    hd.argmax_out = amax ? amax : (int32_t*)(ws + p.amax);
    // the hidden buffer of the ff modules (M x 4 d floats, planned whenever gemm16_for(m, M)) is free here: per-range winners of a split head
    hd.part_max = 8;
    hd.part_v = sc.h4;
    hd.part_i = reinterpret_cast<int32_t*>(sc.h4 + (size_t)hd.part_max * M);
    { PROF(MI355ASR_K_CTC_HEAD); LAUNCH_TRY(launch_gemm16(m, E16_HEAD, false, hd, m->fc_wp, s), "ctc head"); }
    return 0;
  }
  GemmArgs hd{};
  hd.x = sc.xa; hd.y = logits; hd.wp = m->fc_wp; hd.bias = m->fc_b;
  hd.M = M; hd.NT = m->NT_fc; hd.ldy = m->cfg.num_classes; hd.n_valid = m->cfg.num_classes; hd.eps = kLnEps;
  hd.argmax_out = amax ? amax : (int32_t*)(ws + p.amax);
  {
    PROF(MI355ASR_K_CTC_HEAD);
    if (try_head_ld(m, hd, s) == 0) return 0;
    LAUNCH_TRY(launch_gemm_rows(d, EPI_HEAD, false, hd, s), "ctc head");
  }
  return 0;
}


}  // namespace mi355

// =======================================================================================================
// C ABI
// =======================================================================================================
thread_local int mi355asr_last_scheme = SCHEME_F32;   // launch.h: note_scheme

extern "C" {

const char* mi355asr_last_error(void) { return g_err; }
const char* mi355asr_version(void) { return "mi355asr 0.1 (gfx950, fp32 operands as fp16 pairs / bf16 triples on the MFMA pipe)"; }

int mi355asr_create(const mi355asr_config* cfg, mi355asr_model** out) {
  if (!cfg || !out) return fail(MI355ASR_EINVAL, "null argument");
  mi355asr_config c = *cfg;
  // the plain Spectrogram layer feeds all n_dft / 2 + 1 dB bins to the subsampling convs: from here on they are the
  // "mel" axis of every shape (n_mels of the config is not used by that layer, conformer_blocks.py:318-323)
  if (c.mel_layer_type == 2) c.n_mels = c.n_dft / 2 + 1;
  if (c.dmodel != 144 && (c.dmodel % 128 != 0 || c.dmodel < 128 || c.dmodel > 1024))
    return fail(MI355ASR_EINVAL, "dmodel=%d: supported are 144 (ConformerS), 256 (ConformerM / StreamingS) and other multiples of 128 up to 1024 (512 = ConformerL)", c.dmodel);
  if (c.num_heads * c.head_size != c.dmodel)
    return fail(MI355ASR_EINVAL, "num_heads*head_size (%d*%d) must equal dmodel (%d)", c.num_heads, c.head_size, c.dmodel);
  // round 6: the reference's constructors take any head size / kernel size (conformer_blocks.py:278-294); outside the shipped YAMLs'
  // values (36 / 64, 32 / 5) the block runs on slower general kernels instead of failing (attention_kernel<HS, ...>, dwconv_any_kernel)
  if (!attention_head_size_ok(c.head_size))
    return fail(MI355ASR_EINVAL, "head_size=%d: attention kernels are instantiated for 12, 16, 24, 32, 36, 48, 64, 72 and 128", c.head_size);
  if (c.kernel_size < 1 || c.kernel_size > 1024)
    return fail(MI355ASR_EINVAL, "kernel_size=%d: must be in 1 .. 1024", c.kernel_size);
  if (c.num_classes > 0 && (c.ctc_kernel_size < 1 || c.ctc_kernel_size > 1024))
    return fail(MI355ASR_EINVAL, "ctc_kernel_size=%d: must be in 1 .. 1024", c.ctc_kernel_size);
  if (c.reduction_factor != 2 && c.reduction_factor != 4 && c.reduction_factor != 6 && c.reduction_factor != 8)
    return fail(MI355ASR_EINVAL, "reduction_factor=%d: 2, 4, 6 or 8 (conv1 time stride 1 .. 4)", c.reduction_factor);
  if (c.mel_layer_type < 0 || c.mel_layer_type > 2)
    return fail(MI355ASR_EINVAL, "mel_layer_type=%d: 0 (Melspectrogram), 1 (leaf) or 2 (Spectrogram)", c.mel_layer_type);
  if (c.mel_layer_type == 1 && (c.n_mels != 80 || c.stride_ms * c.sample_rate / 1000 != 160 || c.sample_rate != 16000))
    return fail(MI355ASR_EINVAL, "leaf frontend: instantiated for 80 filters, 16 kHz, 10 ms stride (window 401, hop 160)");
  if (c.gemm_dtype != 0 && c.gemm_dtype != 1) return fail(MI355ASR_EINVAL, "gemm_dtype=%d: 0 (fp32 MFMA) or 1 (bf16 MFMA)", c.gemm_dtype);
  if (c.n_dft != 1024) return fail(MI355ASR_EINVAL, "n_dft=%d: the reference hard-codes 1024 (conformer_blocks.py:312)", c.n_dft);
  if (c.mel_layer_type != 2 && c.n_mels != 80 && c.n_mels != 128)
    return fail(MI355ASR_EINVAL, "n_mels=%d: mel kernel instantiated for 80 and 128", c.n_mels);
  if (c.num_blocks < 0 || c.ctc_num_blocks < 0 || c.chunk_size < 0 || c.num_classes < 0)
    return fail(MI355ASR_EINVAL, "negative count in config");
  auto* m = new mi355asr_model();
  m->cfg = c;
  {
    const int t = (int)mi355_env("MI355ASR_LEAF_TERMS", -1);   // 0 = fp32 MFMA Gabor conv; 2 / 3 = bf16 terms per operand
    if (t == 0 || t == 2 || t == 3) m->leaf_terms = t;
  }
  Dims& dm = m->dm;
  dm.hop = c.stride_ms * c.sample_rate / 1000;
  if (dm.hop <= 0) { delete m; return fail(MI355ASR_EINVAL, "stride_ms*sample_rate/1000 must be positive"); }
  dm.nbins = c.n_dft / 2 + 1;
  dm.NT_dft = ceil_div(2 * dm.nbins, 16);          // 65 tiles of (re,im)-interleaved columns
  dm.NT_dft = ceil_div(dm.NT_dft, 13) * 13;
  dm.NCH_dft = dm.NT_dft / 13;
  dm.LP = ceil_div(8 * dm.NT_dft, 16) * 16;        // log-power row stride (bins), 16-byte aligned rows
  dm.KBm = ceil_div(ceil_div(dm.nbins, 16), 2) * 2;   // sweep_k consumes k-blocks in pairs (zero-padded)
  dm.NTm = c.n_mels / 16;
  dm.st1 = c.reduction_factor / 2;
  same_pad(c.n_mels, 3, 2, &dm.F1, &dm.pf1);
  same_pad(dm.F1, 3, 2, &dm.F2, &dm.pf2);
  const int d = c.dmodel;
  auto& ex = m->expected;
  if (c.has_encoder && c.mel_layer_type == 1) {
    // leaf_audio.frontend.Leaf variables (frontend.py:106-160)
    ex.push_back({"mel_layer/tfbanks_preemp/kernel", {2, 1, 1}});
    ex.push_back({"mel_layer/tfbanks_complex_conv/kernel", {c.n_mels, 2}});
    ex.push_back({"mel_layer/learnable_pooling/kernel", {1, 1, c.n_mels, 1}});
    for (const char* n : {"alpha", "delta", "root"}) ex.push_back({std::string("mel_layer/PCEN/") + n, {c.n_mels}});
    ex.push_back({"mel_layer/PCEN/EMA/smooth", {c.n_mels}});
    ex.push_back({"mel_layer/tfbanks_instancenorm/gamma", {c.n_mels}});
    ex.push_back({"mel_layer/tfbanks_instancenorm/beta", {c.n_mels}});
  }
  if (c.has_encoder) {
    if (c.mel_layer_type != 1) {
    ex.push_back({"mel_layer/real_kernels", {c.n_dft, 1, 1, dm.nbins}});
    ex.push_back({"mel_layer/imag_kernels", {c.n_dft, 1, 1, dm.nbins}});
    if (c.mel_layer_type == 0) ex.push_back({"mel_layer/freq2mel", {dm.nbins, c.n_mels}});
    }
    ex.push_back({"conv_subsampling/conv1/kernel", {3, 3, 1, d}});
    ex.push_back({"conv_subsampling/conv1/bias", {d}});
    ex.push_back({"conv_subsampling/conv2/kernel", {3, 3, d, d}});
    ex.push_back({"conv_subsampling/conv2/bias", {d}});
    ex.push_back({"conv_subsampling/linear/kernel", {dm.F2 * d, d}});
    ex.push_back({"conv_subsampling/linear/bias", {d}});
    for (int i = 0; i < c.num_blocks; ++i)
      add_block_expected(ex, "conformer_block_" + std::to_string(i), d, c.num_heads, c.head_size, c.kernel_size);
  }
  if (c.add_wav_info != 0 && c.add_wav_info != 1) { delete m; return fail(MI355ASR_EINVAL, "add_wav_info=%d: 0 or 1", c.add_wav_info); }
  if (c.has_encoder && c.add_wav_info) {
    // WavePickModel(dmodel, hop_size * reduction_factor) (conformer_blocks.py:329-331, wav_model.py:108-131)
    const std::vector<int> scales = wave_pick_scales(dm.hop * c.reduction_factor);
    if (scales.size() != 4) { delete m; return fail(MI355ASR_EINVAL, "add_wav_info: hop_size*reduction_factor=%d does not factor into four strides", dm.hop * c.reduction_factor); }
    m->wp_stride0 = scales[0];
    ex.push_back({"wav_layer/sep_conv/depthwise_kernel", {7, 1, 1}});
    ex.push_back({"wav_layer/sep_conv/pointwise_kernel", {1, 1, 32}});
    ex.push_back({"wav_layer/sep_conv/bias", {32}});
    int cin = 32;
    for (int i = 1; i < 4; ++i) {
      const int ch = std::min(32 * (i + 1), d);
      mi355asr_model::WavStage st{};
      st.cin = cin; st.c = ch; st.stride = scales[i];
      m->wp_stages.push_back(st);
      const std::string n = std::to_string(i);
      ex.push_back({"wav_layer/conv_" + n + "/kernel", {3, cin, ch}});
      ex.push_back({"wav_layer/conv_" + n + "/bias", {ch}});
      ex.push_back({"wav_layer/res_" + n + "/conv5/kernel", {5, ch, ch}});
      ex.push_back({"wav_layer/res_" + n + "/conv5/bias", {ch}});
      ex.push_back({"wav_layer/res_" + n + "/conv1/kernel", {1, ch, ch}});
      ex.push_back({"wav_layer/res_" + n + "/conv1/bias", {ch}});
      ex.push_back({"wav_layer/res_" + n + "/shortcut/kernel", {1, ch, ch}});
      ex.push_back({"wav_layer/res_" + n + "/shortcut/bias", {ch}});
      cin = ch;
    }
    ex.push_back({"wav_layer/final/kernel", {7, cin, d}});
    ex.push_back({"wav_layer/final/bias", {d}});
  }
  if (c.num_classes > 0) {
    ex.push_back({"project/kernel", {d, d}});
    ex.push_back({"project/bias", {d}});
    for (int i = 0; i < c.ctc_num_blocks; ++i)
      add_block_expected(ex, "decoder_conformer_block_" + std::to_string(i), d, c.num_heads, c.head_size,
                         c.ctc_kernel_size);
    ex.push_back({"fully_connected/kernel", {d, c.num_classes}});
    ex.push_back({"fully_connected/bias", {c.num_classes}});
  }
  *out = m;
  return 0;
}

int mi355asr_profile_schemes(const mi355asr_model* m, int32_t* scheme_out, int32_t n) {
  if (!m || !scheme_out || n < MI355ASR_NUM_KERNELS)
    return fail(MI355ASR_EINVAL, "profile_schemes needs an array of at least %d entries", MI355ASR_NUM_KERNELS);
  for (int i = 0; i < MI355ASR_NUM_KERNELS; ++i) scheme_out[i] = m->prof_scheme[i];
  return 0;
}

int mi355asr_profile_enable(mi355asr_model* m, int32_t on) {
  if (!m) return fail(MI355ASR_EINVAL, "null model handle");
  m->prof = on != 0;
  return 0;
}

int mi355asr_profile_read(mi355asr_model* m, double* ms_out, int64_t* count_out, int32_t n, int32_t reset) {
  if (!m || !ms_out || !count_out || n < MI355ASR_NUM_KERNELS)
    return fail(MI355ASR_EINVAL, "profile_read needs arrays of at least %d entries", MI355ASR_NUM_KERNELS);
  for (auto& p : m->ev_pending) {
    HIP_TRY(hipEventSynchronize(p.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.e0, p.e1));
    m->prof_ms[p.cat] += ms;
    m->prof_cnt[p.cat] += 1;
    m->ev_free.push_back(p.e0);
    m->ev_free.push_back(p.e1);
  }
  m->ev_pending.clear();
  for (int i = 0; i < MI355ASR_NUM_KERNELS; ++i) {
    ms_out[i] = m->prof_ms[i];
    count_out[i] = m->prof_cnt[i];
    if (reset) { m->prof_ms[i] = 0; m->prof_cnt[i] = 0; }
  }
  return 0;
}

int mi355asr_destroy(mi355asr_model* m) {
  if (!m) return 0;
  for (auto& p : m->ev_pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
  for (auto e : m->ev_free) (void)hipEventDestroy(e);
  if (m->arena) (void)hipFree(m->arena);
  if (m->arena16) (void)hipFree(m->arena16);
  delete m;
  return 0;
}

// test hook (not in the public header): the host's fp16 rounding of the two-term scheme's weight packs
void mi355asr_test_f16_rne(const float* in, int32_t n, uint16_t* half_bits, float* back) {
  for (int i = 0; i < n; ++i) { half_bits[i] = f16_rne(in[i]); back[i] = f16_to_float(half_bits[i]); }
}

int mi355asr_set_expected_rows(mi355asr_model* m, int64_t rows) {
  if (!m) return fail(MI355ASR_EINVAL, "null handle");
  if (m->finalized) return fail(MI355ASR_ESTATE, "mi355asr_set_expected_rows must come before mi355asr_finalize_weights");
  m->expected_rows = rows < 0 ? -1 : (long)std::min<int64_t>(rows, 1L << 40);
  return 0;
}

int mi355asr_stft_mode(const mi355asr_model* m) {
  if (!m || !m->finalized || (!m->is_chunk && !m->cfg.has_encoder)) return -1;
  if (!m->is_chunk && m->cfg.mel_layer_type == 1) return -1;     // LEAF frontend: no STFT
  return m->fft_ok ? 1 : 0;
}
int mi355asr_num_weights(const mi355asr_model* m) { return m ? (int)m->expected.size() : 0; }
const char* mi355asr_weight_name(const mi355asr_model* m, int32_t i) {
  if (!m || i < 0 || i >= (int)m->expected.size()) return nullptr;
  return m->expected[i].name.c_str();
}

int mi355asr_weight_shape(const mi355asr_model* m, int32_t i, int32_t* rank, int64_t* dims, int32_t max_rank) {
  if (!m || !rank || i < 0 || i >= (int)m->expected.size()) return fail(MI355ASR_EINVAL, "weight index %d out of range", i);
  const auto& d = m->expected[i].dims;
  *rank = (int32_t)d.size();
  if ((int)d.size() > max_rank || (!dims && !d.empty())) return fail(MI355ASR_EINVAL, "dims array too small for rank %d", (int)d.size());
  for (size_t k = 0; k < d.size(); ++k) dims[k] = d[k];
  return 0;
}

int mi355asr_load_weight(mi355asr_model* m, const char* name, const float* data, int32_t rank, const int64_t* dims) {
  if (!m || !name || !data || rank < 0 || (rank > 0 && !dims)) return fail(MI355ASR_EINVAL, "null argument");
  const Expected* e = nullptr;
  for (const auto& x : m->expected)
    if (x.name == name) { e = &x; break; }
  if (!e) return fail(MI355ASR_EWEIGHT, "unknown weight '%s' for this configuration", name);
  // compare shapes with singleton axes squeezed (Keras keeps [n_dft,1,1,nb], [1,d,2d], [k,d,1])
  std::vector<int64_t> got, want;
  int64_t n = 1;
  for (int i = 0; i < rank; ++i) { n *= dims[i]; if (dims[i] != 1) got.push_back(dims[i]); }
  for (auto v : e->dims) if (v != 1) want.push_back(v);
  if (got != want || n != e->numel()) {
    std::string gs, ws_;
    for (int i = 0; i < rank; ++i) gs += (i ? "," : "") + std::to_string(dims[i]);
    for (size_t i = 0; i < e->dims.size(); ++i) ws_ += (i ? "," : "") + std::to_string(e->dims[i]);
    return fail(MI355ASR_EWEIGHT, "weight '%s': shape [%s] does not match expected [%s]", name, gs.c_str(), ws_.c_str());
  }
  HostTensor& t = m->host[name];
  t.data.assign(data, data + n);
  t.set = true;
  m->finalized = false;
  return 0;
}

int mi355asr_load_weight_typed(mi355asr_model* m, const char* name, const void* data, int32_t dtype, int32_t rank,
                               const int64_t* dims) {
  if (!data || rank < 0 || (rank > 0 && !dims)) return fail(MI355ASR_EINVAL, "null argument");
  if (dtype == MI355ASR_DT_F32) return mi355asr_load_weight(m, name, (const float*)data, rank, dims);
  int64_t n = 1;
  for (int i = 0; i < rank; ++i) n *= dims[i];
  if (n < 0 || n > ((int64_t)1 << 32)) return fail(MI355ASR_EINVAL, "weight '%s': bad element count", name ? name : "?");
  std::vector<float> v((size_t)n);
  if (dtype == MI355ASR_DT_F64) {
    const double* p = (const double*)data;
    for (int64_t i = 0; i < n; ++i) v[i] = (float)p[i];
  } else if (dtype == MI355ASR_DT_BF16) {
    const uint16_t* p = (const uint16_t*)data;
    for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)p[i] << 16; std::memcpy(&v[i], &u, 4); }
  } else if (dtype == MI355ASR_DT_F16) {
    const uint16_t* p = (const uint16_t*)data;
    for (int64_t i = 0; i < n; ++i) {
      const uint32_t h = p[i], sign = (h & 0x8000u) << 16, e = (h >> 10) & 31, f = h & 1023;
      uint32_t u;
      if (e == 0) {
        if (f == 0) u = sign;
        else {                                           // subnormal half: normalise
          int sh = 0;
          uint32_t ff = f;
          while (!(ff & 1024)) { ff <<= 1; ++sh; }
          u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((ff & 1023) << 13);
        }
      } else if (e == 31) u = sign | 0x7f800000u | (f << 13);
      else u = sign | ((e + 112) << 23) | (f << 13);
      std::memcpy(&v[i], &u, 4);
    }
  } else {
    return fail(MI355ASR_EINVAL, "weight '%s': unknown dtype %d", name ? name : "?", dtype);
  }
  return mi355asr_load_weight(m, name, v.data(), rank, dims);
}

int mi355asr_finalize_weights(mi355asr_model* m, void* stream) {
  if (!m) return fail(MI355ASR_EINVAL, "null model handle");
  for (const auto& e : m->expected)
    if (!m->host.count(e.name) || !m->host[e.name].set) return fail(MI355ASR_EWEIGHT, "missing weight '%s'", e.name.c_str());
  if (m->is_chunk) return finalize_chunk(m, (hipStream_t)stream);
  if (m->is_translator) return finalize_translator(m, (hipStream_t)stream);
  const auto& c = m->cfg;
  const Dims& dm = m->dm;
  const int d = c.dmodel;
  ArenaBuilder ab;
  ab.ring_terms = m->cfg.gemm_dtype == 1 ? 1 : 3;
  size_t o_dft = 0, o_mel = 0, o_c1w = 0, o_c1b = 0, o_c2w = 0, o_c2b = 0, o_lw = 0, o_lb = 0, o_c2s = 0, o_lws = 0, o_c2h = 0, o_lpp = 0, o_lns = 0;
  float lin_pp_sw = 1.f, c1_l1 = 0.f, c1_bmax = 0.f, c1_ms = 0.f, c1_ws = 0.f;
  float c2_hs = 0.f, c2_ws = 0.f;
  FftOff fo;
  MelBandOff mbo;
  std::vector<BlockOff> eo, co;
  size_t o_leafw = 0, o_leafs = 0, o_lg = 0, o_la = 0, o_ld = 0, o_lr = 0, o_ls = 0, o_lga = 0, o_lbe = 0;
  if (c.has_encoder && c.mel_layer_type == 1) {
    // Gabor filters from (center, sigma) with the layer's constraint (convolution.py:137-153, impulse_responses.py:39-64)
    const int K = 401, NF = c.n_mels;
    const auto& gk = m->host["mel_layer/tfbanks_complex_conv/kernel"].data;
    const double pi = 3.14159265358979323846, s2l2 = std::sqrt(2.0 * std::log(2.0));
    std::vector<double> re((size_t)NF * K), im((size_t)NF * K);
    for (int f = 0; f < NF; ++f) {
      const double mu = std::min(std::max((double)gk[2 * f], 0.0), pi);
      const double sg = std::min(std::max((double)gk[2 * f + 1], 4.0 * s2l2 / pi), K * s2l2 / pi);
      const double den = 1.0 / (std::sqrt(2.0 * pi) * sg);
      for (int t = 0; t < K; ++t) {
        const double tt = t - K / 2, gs = std::exp(-tt * tt / (2.0 * sg * sg));
        re[(size_t)f * K + t] = den * std::cos(mu * tt) * gs;
        im[(size_t)f * K + t] = den * std::sin(mu * tt) * gs;
      }
    }
    o_leafw = ab.put(pack_p16([&](int k, int n) { return k < K ? (float)((n & 1) ? im[(size_t)(n >> 1) * K + k] : re[(size_t)(n >> 1) * K + k]) : 0.f; },
                           26 * 16, 2 * NF, 2 * NF / 16));
    {
      // split-bf16 fragments [13 k-blocks of 32 taps][10 column tiles][terms][64 lanes][8]: lane (r = lane & 15, g = lane >> 4)
      // holds taps 32 kb + 8 g + 0..7 of channel 16 nt + r; term t = round-to-nearest-even bf16 of what terms < t left
      const int NS = m->leaf_terms ? m->leaf_terms : 3;
      std::vector<uint16_t> frag((size_t)13 * 10 * NS * 64 * 8);
      auto rne = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
      for (int kb = 0; kb < 13; ++kb)
        for (int nt = 0; nt < 10; ++nt)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int tap = 32 * kb + 8 * (lane >> 4) + j, ch = 16 * nt + (lane & 15);
              float r = tap < K ? (float)((ch & 1) ? im[(size_t)(ch >> 1) * K + tap] : re[(size_t)(ch >> 1) * K + tap]) : 0.f;
              for (int t = 0; t < NS; ++t) {
                const uint16_t hb = rne(r);
                const uint32_t back = (uint32_t)hb << 16;
                float hf; std::memcpy(&hf, &back, 4);
                r -= hf;
                frag[((((size_t)kb * 10 + nt) * NS + t) * 64 + lane) * 8 + j] = hb;
              }
            }
      std::vector<float> as_f(frag.size() / 2);
      std::memcpy(as_f.data(), frag.data(), frag.size() * 2);
      o_leafs = ab.put(as_f);
    }
    const auto& ps = m->host["mel_layer/learnable_pooling/kernel"].data;
    std::vector<float> gc(NF);
    for (int f = 0; f < NF; ++f) {           // impulse_responses.gaussian_lowpass (:103-119), as exp2 coefficients
      const double sg = std::min(std::max((double)ps[f], 2.0 / K), 0.5);
      const double den = sg * 0.5 * (K - 1);
      gc[f] = (float)(-0.5 * 1.4426950408889634 / (den * den));
    }
    o_lg = ab.put(gc);
    o_la = ab.put(m->host["mel_layer/PCEN/alpha"].data);
    o_ld = ab.put(m->host["mel_layer/PCEN/delta"].data);
    o_lr = ab.put(m->host["mel_layer/PCEN/root"].data);
    o_ls = ab.put(m->host["mel_layer/PCEN/EMA/smooth"].data);
    o_lga = ab.put(m->host["mel_layer/tfbanks_instancenorm/gamma"].data);
    o_lbe = ab.put(m->host["mel_layer/tfbanks_instancenorm/beta"].data);
    const auto& pk = m->host["mel_layer/tfbanks_preemp/kernel"].data;
    m->leaf_p0 = pk[0]; m->leaf_p1 = pk[1];
  }
  if (c.has_encoder) {
  const int nb = dm.nbins;
  if (c.mel_layer_type != 1) {
  const auto& re = m->host["mel_layer/real_kernels"].data;
  const auto& im = m->host["mel_layer/imag_kernels"].data;
  // DFT columns interleaved (re, im) per bin so that power = x^2 + y^2 / z^2 + w^2 inside one lane
  o_dft = ab.put(pack_p16(
      [&](int k, int n) { const int bin = n >> 1; return (n & 1) ? im[(size_t)k * nb + bin] : re[(size_t)k * nb + bin]; },
      c.n_dft, 2 * nb, dm.NT_dft));
  fo = pack_fft(ab, re, im, c.n_dft, nb);
  if (c.mel_layer_type == 0) {
  const auto& f2m = m->host["mel_layer/freq2mel"].data;
  o_mel = ab.put(pack_p16([&](int k, int n) { return k < nb ? f2m[(size_t)k * c.n_mels + n] : 0.f; }, dm.KBm * 16, c.n_mels, dm.NTm));
  mbo = pack_mel_band(ab, f2m, nb, c.n_mels);
  }
  }
  (void)nb;
  o_c1w = ab.put(m->host["conv_subsampling/conv1/kernel"].data);  // [3][3][1][d] == [(i*3+j)*d + c]
  o_c1b = ab.put(m->host["conv_subsampling/conv1/bias"].data);
  const auto& c2 = m->host["conv_subsampling/conv2/kernel"].data;              // [3][3][d][d]
  // K order (c-block, kt, kf, 16): k' = (cb*9 + q)*16 + r  <->  (q = kt*3+kf, c = 16*cb + r)
  o_c2w = ab.put(pack_p16(
      [&](int kp, int n) {
        const int kb = kp / 16, r = kp % 16, cb = kb / 9, q = kb % 9;
        return c2[((size_t)q * d + (16 * cb + r)) * d + n];
      },
      9 * d, d, d / 16));
  o_c2b = ab.put(m->host["conv_subsampling/conv2/bias"].data);
  const bool c2_split = d == 144 || d == 256 || d == 512;
  if (c2_split) {
    // split-bf16 fragments for subconv_split_ring_kernel: column chunks of NTc tiles (all nine at dmodel 144, eight
    // otherwise); step s = 5 cb + pair; lane (r = lane & 15, g = lane >> 4) of column tile nt holds, for out channel
    // 16 nt + r, in-channels 16 cb + 4 g + (j & 3) at tap 2 pair + (j >> 2) (the tenth tap is zero); term t =
    // round-to-nearest-even bf16 of what the terms before it left
    o_c2s = ab.put(pack_conv2_split(c2, d));
    // Two-term fp16 scheme (subconv.hip): needs a bound of conv1's output.  The frontend's dB values lie in [-80, 0]
    // (floor_db, relative to the utterance maximum), a mel value in 80 x the filter's L1 norm; LEAF features have no bound.
    static const int terms_env = (int)mi355_env("MI355ASR_SUBCONV_TERMS", 2);
    if ((terms_env == 2 || terms_env == 22) && c.mel_layer_type != 1) {
      double mb = 80.0;
      if (c.mel_layer_type == 0) {
        const auto& f2m = m->host["mel_layer/freq2mel"].data;
        double l1 = 0.0;
        for (int mm = 0; mm < c.n_mels; ++mm) {
          double sum = 0.0;
          for (int k = 0; k < dm.nbins; ++k) sum += std::fabs((double)f2m[(size_t)k * c.n_mels + mm]);
          l1 = std::max(l1, sum);
        }
        mb *= l1;
      }
      const auto& w1 = m->host["conv_subsampling/conv1/kernel"].data;
      const auto& b1 = m->host["conv_subsampling/conv1/bias"].data;
      double bx = 0.0, wmax = 0.0, l1max = 0.0, bmax = 0.0;
      for (int ch = 0; ch < d; ++ch) {
        double sum = 0.0;
        for (int t = 0; t < 9; ++t) sum += std::fabs((double)w1[(size_t)t * d + ch]);
        bx = std::max(bx, std::fabs((double)b1[ch]) + mb * sum);
        l1max = std::max(l1max, sum);
        bmax = std::max(bmax, std::fabs((double)b1[ch]));
      }
      c1_l1 = (float)(l1max * (1.0 + 1e-6));
      c1_bmax = (float)(bmax * (1.0 + 1e-6));
      {                                            // conv1 on the matrix pipe: |mel| <= mb, the largest |conv1 weight|
        double w1max = 0.0;
        for (float v : w1) w1max = std::max(w1max, std::fabs((double)v));
        c1_ms = half_scale_for(mb * (1.0 + 1e-6));
        c1_ws = half_scale_for(w1max);
      }
      for (float v : c2) wmax = std::max(wmax, std::fabs((double)v));
      c2_hs = half_scale_for(bx);
      c2_ws = half_scale_for(wmax);
      if (c2_hs > 0.f && c2_ws > 0.f) o_c2h = ab.put(pack_conv2_half(c2, d, c2_ws));
    }
  }
  const auto& lin = m->host["conv_subsampling/linear/kernel"].data;
  o_lw = ab.put(pack_p16([&](int k, int n) { return lin[(size_t)k * d + n]; }, dm.F2 * d, d, d / 16));
  if (ring_packs_wanted(m)) put_ring(ab, o_lw, [&](int k, int n) { return lin[(size_t)k * d + n]; }, dm.F2 * d, d, false);
  o_lb = ab.put(m->host["conv_subsampling/linear/bias"].data);
  if (d == 144) {
    // the same kernel for sublinear_split_kernel: 1728 fragments per 32-wide step, padded to 7 x 256 (4 floats each)
    o_lws = ab.put(pack_linear_split(lin, dm.F2 * d, d));
    // two-term fp16 stream (pp_sublinear_kernel): chunk f = rows 144 f .. 144 f + 143 of the kernel, the bias in row 144 of chunk 0
    const auto& lb = m->host["conv_subsampling/linear/bias"].data;
    std::vector<float> pp, lin_plain;
    lin_pp_sw = append_pp_plain(pp, [&](int k, int n) {
      const int f = n / d, col = n - f * d;
      return k < d ? lin[((size_t)f * d + k) * d + col] : (f == 0 ? lb[col] : 0.f);
    }, dm.F2, &lin_plain);
    o_lpp = ab.put(pp);
    o_lns = ab.put(lin_plain);
  }
  for (int i = 0; i < c.num_blocks; ++i)
    eo.push_back(pack_block(m, ab, "conformer_block_" + std::to_string(i), d, c.num_heads, c.head_size, c.kernel_size));
  }
  struct WavOff { size_t cw, cb, w5, b5, w1, b1, ws, bs; };
  std::vector<WavOff> wo;
  size_t o_wdw = 0, o_wpw = 0, o_wb = 0, o_wfw = 0, o_wfb = 0;
  if (c.has_encoder && c.add_wav_info) {
    // a Conv1D kernel [k, cin, cout] is already the [k*cin, cout] matrix of the GEMM over k overlapping channels-last rows
    auto conv_w = [&](const std::string& name, int K, int N) {
      const auto& w = m->host[name].data;
      return ab.put(pack_p16([&](int k, int n) { return w[(size_t)k * N + n]; }, K, N, N / 16));
    };
    o_wdw = ab.put(m->host["wav_layer/sep_conv/depthwise_kernel"].data);
    o_wpw = ab.put(m->host["wav_layer/sep_conv/pointwise_kernel"].data);
    o_wb = ab.put(m->host["wav_layer/sep_conv/bias"].data);
    for (size_t i = 0; i < m->wp_stages.size(); ++i) {
      const auto& st = m->wp_stages[i];
      const std::string n = std::to_string(i + 1);
      WavOff w{};
      w.cw = conv_w("wav_layer/conv_" + n + "/kernel", 3 * st.cin, st.c);
      w.cb = ab.put(m->host["wav_layer/conv_" + n + "/bias"].data);
      w.w5 = conv_w("wav_layer/res_" + n + "/conv5/kernel", 5 * st.c, st.c);
      w.b5 = ab.put(m->host["wav_layer/res_" + n + "/conv5/bias"].data);
      w.w1 = conv_w("wav_layer/res_" + n + "/conv1/kernel", st.c, st.c);
      w.b1 = ab.put(m->host["wav_layer/res_" + n + "/conv1/bias"].data);
      w.ws = conv_w("wav_layer/res_" + n + "/shortcut/kernel", st.c, st.c);
      w.bs = ab.put(m->host["wav_layer/res_" + n + "/shortcut/bias"].data);
      wo.push_back(w);
    }
    o_wfw = conv_w("wav_layer/final/kernel", 7 * m->wp_stages.back().c, d);
    o_wfb = ab.put(m->host["wav_layer/final/bias"].data);
  }
  size_t o_pw = 0, o_pb = 0, o_fw = 0, o_fb = 0, o_ppp = 0;
  float proj_pp_sw = 1.f;
  if (c.num_classes > 0) {
    const auto& pj = m->host["project/kernel"].data;
    o_pw = ab.put(pack_p16([&](int k, int n) { return pj[(size_t)k * d + n]; }, d, d, d / 16));
    // (bf16 mode: the one-term ring of the projection, for gemm256_bf16_kernel at many rows -- config 3's 16 640)
    if (ring_packs_wanted(m) && ab.ring_terms == 1) put_ring(ab, o_pw, [&](int k, int n) { return pj[(size_t)k * d + n]; }, d, d, false);
    o_pb = ab.put(m->host["project/bias"].data);
    if (d == 144) {
      const auto& pb = m->host["project/bias"].data;
      std::vector<float> pp;
      proj_pp_sw = append_pp_plain(pp, [&](int k, int n) { return k < d ? pj[(size_t)k * d + n] : pb[n]; }, 1);
      o_ppp = ab.put(pp);
    }
    for (int i = 0; i < c.ctc_num_blocks; ++i)
      co.push_back(pack_block(m, ab, "decoder_conformer_block_" + std::to_string(i), d, c.num_heads, c.head_size,
                              c.ctc_kernel_size));
    const auto& fc = m->host["fully_connected/kernel"].data;
    const int V = c.num_classes;
    const int ct = gemm_ct(d, EPI_HEAD);
    m->NT_fc = ceil_div(ceil_div(V, 16), ct) * ct;
    o_fw = ab.put(pack_p16([&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V, m->NT_fc));
    if (ring_packs_wanted(m)) put_ring_head(ab, o_fw, [&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V);
    put_head_slabs(ab, o_fw, [&](int k, int n) { return fc[(size_t)k * V + n]; }, d, V, m->host["fully_connected/bias"].data.data());
    o_fb = ab.put_padded(m->host["fully_connected/bias"].data.data(), V, (size_t)m->NT_fc * 16);
  }
  if (m->arena) { (void)hipFree(m->arena); m->arena = nullptr; }
  HIP_TRY(hipMalloc((void**)&m->arena, ab.buf.size() * sizeof(float)));
  m->arena_floats = ab.buf.size();
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(m->arena, ab.buf.data(), ab.buf.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // ab.buf is freed on return
  const float* base = m->arena;
  m->ring_of.clear();
  register_rings(m, ab, base);
  m->dft_wp = base + o_dft; m->mel_wp = base + o_mel;
  m->fft_ok = fo.ok;
  use_mel_band(m, mbo, base);
  m->fft_w1p = base + fo.w1; m->fft_w2p = base + fo.w2; m->fft_twc = base + fo.twc; m->fft_tws = base + fo.tws;
  m->fft_w1s = base + fo.w1s; m->fft_w2s = base + fo.w2s; m->fft_w1h = base + fo.w1h; m->fft_w2h = base + fo.w2h;
  m->fft_win = base + fo.win;
  m->c1_w = base + o_c1w; m->c1_b = base + o_c1b; m->c2_wp = base + o_c2w; m->c2_b = base + o_c2b; m->c2_wsplit = ((d == 144 || d == 256 || d == 512) && c.has_encoder) ? base + o_c2s : nullptr;
  m->c2_whalf = o_c2h ? base + o_c2h : nullptr; m->c2_hscale = c2_hs; m->c2_wscale = c2_ws;
  m->c1_l1 = c1_l1; m->c1_bmax = c1_bmax; m->c1_mscale = c1_ms; m->c1_wscale = c1_ws;
  m->lin_wsplit = (d == 144 && c.has_encoder) ? base + o_lws : nullptr;
  m->lin_pp = (d == 144 && c.has_encoder) ? base + o_lpp : nullptr;
  m->lin_ns = (d == 144 && c.has_encoder) ? base + o_lns : nullptr;
  m->lin_pp_sw = lin_pp_sw;
  m->proj_pp = (d == 144 && c.num_classes > 0) ? base + o_ppp : nullptr;
  m->proj_pp_sw = proj_pp_sw;
  m->lin_wp = base + o_lw; m->lin_b = base + o_lb;
  m->proj_wp = base + o_pw; m->proj_b = base + o_pb; m->fc_wp = base + o_fw; m->fc_b = base + o_fb;
  m->leaf_wp = base + o_leafw; m->leaf_wsplit = base + o_leafs; m->leaf_gcoef = base + o_lg; m->leaf_alpha = base + o_la; m->leaf_delta = base + o_ld;
  m->leaf_root = base + o_lr; m->leaf_smooth = base + o_ls; m->leaf_gamma = base + o_lga; m->leaf_beta = base + o_lbe;
  m->wp_dw = base + o_wdw; m->wp_pw = base + o_wpw; m->wp_b = base + o_wb; m->wp_fw = base + o_wfw; m->wp_fb = base + o_wfb;
  for (size_t i = 0; i < wo.size(); ++i) {
    auto& st = m->wp_stages[i];
    st.cw = base + wo[i].cw; st.cb = base + wo[i].cb; st.w5 = base + wo[i].w5; st.b5 = base + wo[i].b5;
    st.w1 = base + wo[i].w1; st.b1 = base + wo[i].b1; st.ws = base + wo[i].ws; st.bs = base + wo[i].bs;
  }
  if (m->arena16) { (void)hipFree(m->arena16); m->arena16 = nullptr; }
  if (c.gemm_dtype == 1) {
    const size_t n16 = (m->arena_floats + 3) & ~(size_t)3;
    HIP_TRY(hipMalloc((void**)&m->arena16, n16 * sizeof(unsigned short)));
    if (launch_to_bf16(m->arena, m->arena16, m->arena_floats & ~(size_t)3, s) != 0) return fail(MI355ASR_EINVAL, "bf16 conversion failed");
    HIP_TRY(hipStreamSynchronize(s));
  }
  m->enc_blocks.clear();
  m->ctc_blocks.clear();
  for (auto& o : eo) m->enc_blocks.push_back(resolve(o, base));
  for (auto& o : co) m->ctc_blocks.push_back(resolve(o, base));
  m->finalized = true;
  return 0;
}

int mi355asr_out_frames(const mi355asr_model* m, int32_t L, int32_t* mel_frames, int32_t* enc_frames) {
  if (!m) return fail(MI355ASR_EINVAL, "null model handle");
  Geometry g;
  int rc = geometry(m, 1, L, &g);
  if (rc) return rc;
  if (mel_frames) *mel_frames = g.F * g.nblk;
  if (enc_frames) *enc_frames = g.T * g.nblk;
  return 0;
}

int mi355asr_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t L, size_t* bytes) {
  if (!m || !bytes) return fail(MI355ASR_EINVAL, "null argument");
  Geometry g;
  int rc = geometry(m, B, L, &g);
  if (rc) return rc;
  *bytes = make_plan(m, g.Bp, g.F, g.T).total;
  return 0;
}

int mi355asr_ctc_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t T, size_t* bytes) {
  if (!m || !bytes || B <= 0 || T <= 0) return fail(MI355ASR_EINVAL, "bad argument");
  *bytes = make_plan(m, B, 16, T).logp;
  return 0;
}

int mi355asr_encoder_forward(mi355asr_model* m, const float* wav, int32_t B, int32_t L, float* enc_out, void* ws,
                             size_t ws_bytes, void* stream) {
  int rc = check_ready(m, true);
  if (rc) return rc;
  if (!wav || !enc_out || !ws) return fail(MI355ASR_EINVAL, "null device pointer");
  Geometry g;
  rc = geometry(m, B, L, &g);
  if (rc) return rc;
  const Plan p = make_plan(m, g.Bp, g.F, g.T);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  return encoder_impl(m, wav, g, p, (char*)ws, enc_out, (hipStream_t)stream);
}

int mi355asr_ctc_forward(mi355asr_model* m, const float* enc, int32_t B, int32_t T, float* logits, int32_t* amax,
                         void* ws, size_t ws_bytes, void* stream) {
  int rc = check_ready(m);
  if (rc) return rc;
  if (m->cfg.num_classes <= 0) return fail(MI355ASR_ESTATE, "model was created without a CTC head (num_classes=0)");
  if (!enc || !ws || B <= 0 || T <= 0) return fail(MI355ASR_EINVAL, "bad argument");
  const Plan p = make_plan(m, B, 16, T);
  if (ws_bytes < p.logp) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.logp);
  return ctc_impl(m, enc, B, T, p, (char*)ws, logits, amax, (hipStream_t)stream);
}

int mi355asr_ctc_greedy(const int32_t* frame_argmax, const int32_t* in_len, int32_t B, int32_t T, int32_t blank,
                        int32_t* ids, int32_t* out_len, void* stream) {
  if (!frame_argmax || !ids || !out_len || B <= 0 || T <= 0) return fail(MI355ASR_EINVAL, "bad argument");
  CollapseArgs ca{frame_argmax, in_len, ids, out_len, B, T, blank};
  LAUNCH_TRY(launch_collapse(ca, (hipStream_t)stream), "ctc collapse");
  return 0;
}

int mi355asr_ctc_prefix_beam_host(const float* probs, const int32_t* in_len, int32_t B, int32_t T, int32_t V,
                                  int32_t beam_size, double cutoff_prob, int32_t cutoff_top_n, int32_t num_threads,
                                  int32_t max_len, int32_t* ids, int32_t* lens, float* scores, int32_t* n_hyp) {
  if (!probs || !ids || !lens || !scores || !n_hyp) return fail(MI355ASR_EINVAL, "null pointer");
  if (B <= 0 || T <= 0 || V < 2 || beam_size <= 0 || max_len <= 0 || cutoff_top_n <= 0)
    return fail(MI355ASR_EINVAL, "bad beam-search argument (B=%d T=%d V=%d beam=%d max_len=%d top_n=%d)", B, T, V,
                beam_size, max_len, cutoff_top_n);
  return mi355asr_beam_host_impl(probs, in_len, B, T, V, beam_size, cutoff_prob, cutoff_top_n, num_threads, max_len, ids,
                                 lens, scores, n_hyp);
}

int mi355asr_ctc_prefix_beam(const float* x, int32_t is_logits, const int32_t* in_len, int32_t B, int32_t T, int32_t V,
                             int32_t beam_size, double cutoff_prob, int32_t cutoff_top_n, int32_t num_threads,
                             int32_t max_len, int32_t* ids, int32_t* lens, float* scores, int32_t* n_hyp, void* ws,
                             size_t ws_bytes, void* stream) {
  if (!x || !ids || !lens || !scores || !n_hyp || !ws) return fail(MI355ASR_EINVAL, "null pointer");
  if (B <= 0 || T <= 0 || V < 2 || beam_size <= 0 || max_len <= 0 || cutoff_top_n <= 0)
    return fail(MI355ASR_EINVAL, "bad beam-search argument");
  if (!(cutoff_prob < 1.0))
    return fail(MI355ASR_EINVAL, "cutoff_prob >= 1 disables pruning in the reference (every class is visited): use "
                "mi355asr_ctc_prefix_beam_host for that mode");
  const int N = std::min(cutoff_top_n, V);
  if (N > 128) return fail(MI355ASR_EINVAL, "cutoff_top_n=%d: the selection kernel supports up to 128", cutoff_top_n);
  const size_t frames = (size_t)B * T;
  const size_t need = frames * N * (sizeof(int32_t) + sizeof(float));
  if (ws_bytes < need) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  int32_t* d_idx = (int32_t*)ws;
  float* d_p = (float*)((char*)ws + frames * N * sizeof(int32_t));
  if (mi355asr_launch_topn(x, (int)frames, V, N, is_logits, d_idx, d_p, s) != 0)
    return fail(MI355ASR_EHIP, "top-n kernel launch failed (V=%d needs %zu bytes of LDS)", V, (size_t)V * 4);
  // MI355ASR_BEAM_DEVICE=0: the prefix search on host threads (beam.hip) instead of the device kernel (beam_device.hip)
  static const bool dev_env = mi355_env("MI355ASR_BEAM_DEVICE", 1) != 0;
  const size_t need_dev = ((need + 255) & ~(size_t)255) + mi355asr_beam_device_ws_bytes(B, T, beam_size, max_len);
  if (dev_env && mi355asr_beam_device_applicable(V, N, beam_size) && ws_bytes >= need_dev) {
    char* w = (char*)ws + ((need + 255) & ~(size_t)255);
    BeamDeviceArgs a{};
    a.top_idx = d_idx; a.top_p = d_p; a.B = B; a.T = T; a.V = V; a.N = N; a.beam = beam_size;
    a.cutoff_top_n = cutoff_top_n; a.max_len = max_len; a.cutoff_prob = cutoff_prob;
    int32_t* d_len = nullptr;
    long long* d_prof = nullptr;
    (void)mi355asr_beam_device_carve(w, B, T, beam_size, max_len, &a, &d_len, &d_prof);   // the same layout the size query adds up
    // MI355ASR_BEAM_PROF=1: clock counters of utterance 0's search, printed per call (where a frame's time goes)
    static const bool prof_env = mi355_env("MI355ASR_BEAM_PROF", 0) != 0;
    if (prof_env) {
      a.prof = d_prof;
      HIP_TRY(hipMemsetAsync(a.prof, 0, 16 * sizeof(long long), s));
    }
    // The four results sit next to each other in the workspace (mi355asr_beam_device_carve): ONE copy into a pinned staging buffer of
    // the calling thread, then host copies -- the caller's arrays are pageable (NumPy), and four hipMemcpyAsync into pageable memory are
    // four staged, synchronous copies: ~120 us behind a 2.4 ms search (round 6, kernel trace of config 5); the lengths go up through
    // the same buffer (behind the results' span), so the search is launched without a host-side wait.  The buffer is kept for the
    // thread's lifetime (never freed: at process exit the runtime may be gone before a destructor would run).
    const size_t n_ids = (size_t)B * beam_size * max_len * sizeof(int32_t), n_lens = (size_t)B * beam_size * sizeof(int32_t),
                 n_scores = (size_t)B * beam_size * sizeof(float), n_nh = (size_t)B * sizeof(int32_t);
    const size_t span = (size_t)((const char*)a.n_hyp - (const char*)a.ids) + n_nh, want = span + n_nh + 64;
    static thread_local char* stage = nullptr;
    static thread_local size_t stage_cap = 0;
    if (want > stage_cap) {
      if (stage) (void)hipHostFree(stage);
      stage = nullptr; stage_cap = 0;
      void* q = nullptr;
      if (hipHostMalloc(&q, want + want / 4, hipHostMallocDefault) == hipSuccess) { stage = (char*)q; stage_cap = want + want / 4; }
      else (void)hipGetLastError();
    }
    if (in_len) {
      const void* src = in_len;
      if (stage) { std::memcpy(stage + span, in_len, n_nh); src = stage + span; }
      HIP_TRY(hipMemcpyAsync(d_len, src, n_nh, hipMemcpyHostToDevice, s));
      a.in_len = d_len;
    }
    if (mi355asr_launch_beam_device(&a, s) != 0) return fail(MI355ASR_EHIP, "device beam search launch failed");
    long long prof[16] = {0};
    if (stage) {
      HIP_TRY(hipMemcpyAsync(stage, a.ids, span, hipMemcpyDeviceToHost, s));
    } else {
      HIP_TRY(hipMemcpyAsync(ids, a.ids, n_ids, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(lens, a.lens, n_lens, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(scores, a.scores, n_scores, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(n_hyp, a.n_hyp, n_nh, hipMemcpyDeviceToHost, s));
    }
    if (a.prof) HIP_TRY(hipMemcpyAsync(prof, a.prof, sizeof(prof), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (stage) {
      std::memcpy(ids, stage, n_ids);
      std::memcpy(lens, stage + ((const char*)a.lens - (const char*)a.ids), n_lens);
      std::memcpy(scores, stage + ((const char*)a.scores - (const char*)a.ids), n_scores);
      std::memcpy(n_hyp, stage + ((const char*)a.n_hyp - (const char*)a.ids), n_nh);
    }
    if (a.prof) {
      const double f = (double)std::max(1ll, prof[8]);
      fprintf(stderr, "[mi355asr] beam %d, utterance 0: %lld frames (%lld redone by the radix path); clocks per frame: entries %.0f, "
              "keys+ranks %.0f, keep %.0f, radix path %.0f (keys %.0f, select %.0f, compaction %.0f); inside the entry phase: thread 0 "
              "%.0f (%.0f up to the parent search), wave 3's candidate list %.0f (%.0f cumulative cut-off)\n", beam_size, prof[8],
              prof[4], prof[0] / f, prof[1] / f, prof[2] / f, prof[3] / f, prof[5] / f, prof[6] / f, prof[7] / f, prof[9] / f, prof[10] / f,
              prof[11] / f, prof[12] / f);
    }
    return 0;
  }
  std::vector<int32_t> h_idx(frames * N);
  std::vector<float> h_p(frames * N);
  HIP_TRY(hipMemcpyAsync(h_idx.data(), d_idx, h_idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_p.data(), d_p, h_p.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return mi355asr_beam_topn_impl(h_idx.data(), h_p.data(), in_len, B, T, V, N, beam_size, cutoff_prob, cutoff_top_n,
                                 num_threads, max_len, ids, lens, scores, n_hyp);
}

int mi355asr_beam_math_eval(int32_t kind, const float* in_dev, void* out_dev, int32_t n, void* stream) {
  if (!in_dev || !out_dev || n < 0 || kind < 0 || kind > 3) return fail(MI355ASR_EINVAL, "bad argument");
  if (mi355asr_launch_refmath_eval(kind, in_dev, out_dev, n, (hipStream_t)stream) != 0)
    return fail(MI355ASR_EHIP, "refmath kernel launch failed");
  return 0;
}

int mi355asr_ctc_prefix_beam_workspace_bytes(int32_t B, int32_t T, int32_t cutoff_top_n, int32_t beam_size, int32_t max_len,
                                             size_t* bytes) {
  if (!bytes || B <= 0 || T <= 0 || cutoff_top_n <= 0 || beam_size <= 0 || max_len <= 0) return fail(MI355ASR_EINVAL, "bad argument");
  const size_t need = (size_t)B * T * std::min(cutoff_top_n, 128) * 8;
  *bytes = ((need + 255) & ~(size_t)255) + mi355asr_beam_device_ws_bytes(B, T, beam_size, max_len);
  return 0;
}

int mi355asr_recognize(mi355asr_model* m, const float* wav, int32_t B, int32_t L, const int32_t* in_len, int32_t* ids,
                       int32_t* out_len, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_ready(m, true);
  if (rc) return rc;
  if (m->cfg.num_classes <= 0) return fail(MI355ASR_ESTATE, "model was created without a CTC head (num_classes=0)");
  if (!wav || !ids || !out_len || !ws) return fail(MI355ASR_EINVAL, "null device pointer");
  Geometry g;
  rc = geometry(m, B, L, &g);
  if (rc) return rc;
  const Plan p = make_plan(m, g.Bp, g.F, g.T);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  char* w = (char*)ws;
  hipStream_t s = (hipStream_t)stream;
  float* enc = (float*)(w + p.enc);
  rc = encoder_impl(m, wav, g, p, w, enc, s);
  if (rc) return rc;
  const int Ttot = g.T * g.nblk;
  int32_t* amax = (int32_t*)(w + p.amax);
  rc = ctc_impl(m, enc, B, Ttot, p, w, nullptr, amax, s);
  if (rc) return rc;
  CollapseArgs ca{amax, in_len, ids, out_len, B, Ttot, m->cfg.num_classes - 1};
  { PROF(MI355ASR_K_COLLAPSE); LAUNCH_TRY(launch_collapse(ca, s), "ctc collapse"); }
  return 0;
}

int mi355asr_melspectrogram(mi355asr_model* m, const float* wav, int32_t B, int32_t L, float* mel, void* ws,
                            size_t ws_bytes, void* stream) {
  int rc = check_ready(m, true);
  if (rc) return rc;
  if (!wav || !mel || !ws) return fail(MI355ASR_EINVAL, "null device pointer");
  Geometry g;
  rc = geometry(m, B, L, &g);
  if (rc) return rc;
  const Plan p = make_plan(m, g.Bp, g.F, g.T);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  char* w = (char*)ws;
  return run_mel(m, wav, g.Bp, g.Lb, g.F, (float*)(w + p.logp), (float*)(w + p.pmax), (float*)(w + p.umax), mel,
                 (hipStream_t)stream);
}

int mi355asr_conv_subsampling(mi355asr_model* m, const float* mel, int32_t B, int32_t F, float* out, void* ws,
                              size_t ws_bytes, void* stream) {
  int rc = check_ready(m, true);
  if (rc) return rc;
  if (!mel || !out || !ws || B <= 0 || F <= 0) return fail(MI355ASR_EINVAL, "bad argument");
  const int T = ceil_div(ceil_div(F, m->dm.st1), 2);
  const Plan p = make_plan(m, B, F, T);
  if (ws_bytes < p.total) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total);
  return run_subsampling(m, mel, B, F, (float*)((char*)ws + p.sub), out, (hipStream_t)stream, false, nullptr);
}

int mi355asr_conformer_block(mi355asr_model* m, int32_t stack, int32_t index, const float* x, int32_t B, int32_t T,
                             float* y, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_ready(m);
  if (rc) return rc;
  if (!x || !y || !ws || B <= 0 || T <= 0) return fail(MI355ASR_EINVAL, "bad argument");
  const auto& blocks = stack == 0 ? m->enc_blocks : m->ctc_blocks;
  if (stack < 0 || stack > 1 || index < 0 || index >= (int)blocks.size())
    return fail(MI355ASR_EINVAL, "no block %d in stack %d", index, stack);
  const Plan p = make_plan(m, B, 16, T);
  if (ws_bytes < p.logp) return fail(MI355ASR_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.logp);
  char* w = (char*)ws;
  hipStream_t s = (hipStream_t)stream;
  Scratch sc{(float*)(w + p.xa), (float*)(w + p.xb), (float*)(w + p.qkv),
             (float*)(w + p.ctx), (float*)(w + p.u), (float*)(w + p.dw)};
  sc.h4 = (float*)(w + p.h4);
  HIP_TRY(hipMemcpyAsync(sc.xa, x, (size_t)B * T * m->cfg.dmodel * 4, hipMemcpyDeviceToDevice, s));
  BlockOpts bo;
  bo.ksz = stack == 0 ? m->cfg.kernel_size : m->cfg.ctc_kernel_size;
  bo.fc = stack == 0 ? m->cfg.fc_factor : m->cfg.ctc_fc_factor;
  return run_block(m, blocks[index], bo, sc, B, T, y, s);
}

}  // extern "C"
