// The reference's C++ recogniser (Inference/CppInference/onnx/src/core/asr_session.{h,cpp}: class ASR::Session with
// EncoderInference / CTCInference over ONNX Runtime, then ctc_greedy_decoder) on the C ABI of libmi355asr.so -- a
// compiled version of the patch INTEGRATION.md section 2 describes.  Plain C++ host code: the only HIP it touches is
// the runtime API for device buffers (hipMalloc / hipMemcpyAsync / streams); no kernels, no torch.
//
//   hipcc -std=c++17 -Iinclude examples/asr_session.cpp -Ltensorflowasr_amd -lmi355asr -Wl,-rpath,$PWD/tensorflowasr_amd -o asr_session
//   ./asr_session [seconds]          -> random-initialised ConformerCTC(S), a synthetic utterance, the greedy token ids
//
// Weights: every tensor the handle expects is enumerated with mi355asr_num_weights / _weight_name / _weight_shape and
// filled here (DFT kernels and a triangular filterbank analytically, the rest from a fixed LCG); a real deployment
// reads the same names from the trainer's checkpoint (tensorflowasr_amd/checkpoint.py lists the mappings).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "mi355asr.h"

namespace ASR {

static void check(int rc) {
  if (rc != 0) throw std::runtime_error(std::string("mi355asr: ") + mi355asr_last_error());
}
static void hcheck(hipError_t e) {
  if (e != hipSuccess) throw std::runtime_error(std::string("hip: ") + hipGetErrorString(e));
}

class Session {
 public:
  explicit Session(int num_classes) {
    // asr/configs/conformerS.yml + am_data.yml: dmodel 144, 13 + 1 blocks, 4 x 36 heads, kernel 32, 80 mels, 16 kHz, 10 ms
    mi355asr_config cfg{};
    cfg.dmodel = 144; cfg.num_blocks = 13; cfg.head_size = 36; cfg.num_heads = 4; cfg.kernel_size = 32; cfg.fc_factor = 0.5f;
    cfg.reduction_factor = 4; cfg.n_mels = 80; cfg.sample_rate = 16000; cfg.stride_ms = 10; cfg.n_dft = 1024;
    cfg.has_encoder = 1; cfg.num_classes = num_classes; cfg.ctc_num_blocks = 1; cfg.ctc_kernel_size = 32; cfg.ctc_fc_factor = 0.5f;
    check(mi355asr_create(&cfg, &gpu_));
    hcheck(hipStreamCreate(&stream_));
    LoadSyntheticWeights();
    check(mi355asr_finalize_weights(gpu_, stream_));
  }
  ~Session() {
    if (gpu_) mi355asr_destroy(gpu_);
    (void)hipFree(ws_); (void)hipFree(d_wav_); (void)hipFree(d_ids_); (void)hipFree(d_len_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;

  // replaces EncoderInference + CTCInference + ctc_greedy_decoder (asr_session.cpp:77-123, :235-239)
  std::vector<int> Recognize(const std::vector<float>& wav) {
    const int L = (int)wav.size();
    int32_t F = 0, T = 0;
    check(mi355asr_out_frames(gpu_, L, &F, &T));
    size_t need = 0;
    check(mi355asr_workspace_bytes(gpu_, 1, L, &need));
    Grow(&ws_, &ws_bytes_, need);
    Grow((void**)&d_wav_, &wav_bytes_, (size_t)L * sizeof(float));
    Grow((void**)&d_ids_, &ids_bytes_, (size_t)T * sizeof(int32_t));
    if (!d_len_) hcheck(hipMalloc((void**)&d_len_, sizeof(int32_t)));
    hcheck(hipMemcpyAsync(d_wav_, wav.data(), (size_t)L * sizeof(float), hipMemcpyHostToDevice, stream_));
    check(mi355asr_recognize(gpu_, d_wav_, 1, L, nullptr, d_ids_, d_len_, ws_, ws_bytes_, stream_));
    std::vector<int32_t> ids(T);
    int32_t n = 0;
    hcheck(hipMemcpyAsync(ids.data(), d_ids_, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    hcheck(hipMemcpyAsync(&n, d_len_, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    hcheck(hipStreamSynchronize(stream_));
    return std::vector<int>(ids.begin(), ids.begin() + n);   // = ctc_greedy_decoder(probs, blank = num_classes - 1, vocab)
  }
  int frames_for(int L) const { int32_t F = 0, T = 0; check(mi355asr_out_frames(gpu_, L, &F, &T)); return T; }

 private:
  static void Grow(void** p, size_t* have, size_t need) {
    if (need <= *have) return;
    (void)hipFree(*p);
    hcheck(hipMalloc(p, need));
    *have = need;
  }
  void LoadSyntheticWeights() {
    uint32_t lcg = 12345u;
    auto uni = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)(lcg >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; };
    const int n = mi355asr_num_weights(gpu_);
    for (int i = 0; i < n; ++i) {
      const std::string name = mi355asr_weight_name(gpu_, i);
      int32_t rank = 0;
      int64_t dims[8];
      check(mi355asr_weight_shape(gpu_, i, &rank, dims, 8));
      int64_t numel = 1;
      for (int k = 0; k < rank; ++k) numel *= dims[k];
      std::vector<float> w((size_t)numel);
      const std::string leaf = name.substr(name.rfind('/') + 1);
      if (leaf == "real_kernels" || leaf == "imag_kernels") {          // [n_dft, 1, 1, bins]: hann window x cos / -sin
        const int N = (int)dims[0], bins = (int)dims[3];
        for (int t = 0; t < N; ++t) {
          const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * t / N);
          for (int k = 0; k < bins; ++k) {
            const double ph = 2.0 * M_PI * (double)((int64_t)k * t % N) / N;
            w[(size_t)t * bins + k] = (float)(win * (leaf == "real_kernels" ? std::cos(ph) : -std::sin(ph)));
          }
        }
      } else if (leaf == "freq2mel") {                                  // [bins, mels]: a plain triangular bank
        const int bins = (int)dims[0], mels = (int)dims[1];
        for (int b = 0; b < bins; ++b)
          for (int m = 0; m < mels; ++m) {
            const double c = (m + 1.0) * bins / (mels + 1.0), half = (double)bins / (mels + 1.0);
            w[(size_t)b * mels + m] = (float)std::fmax(0.0, 1.0 - std::fabs(b - c) / half) / (float)half;
          }
      } else if (leaf == "gamma" || leaf == "moving_variance") {
        for (auto& v : w) v = 1.0f;
      } else if (leaf == "beta" || leaf == "moving_mean" || leaf == "bias" || leaf == "projection_bias") {
        for (auto& v : w) v = 0.0f;
      } else {                                                          // glorot-like scale from the last two dims
        const double fan = rank >= 2 ? (double)(dims[rank - 2] + dims[rank - 1]) * (double)(numel / (dims[rank - 2] * dims[rank - 1])) : (double)numel;
        const float lim = (float)std::sqrt(6.0 / fan);
        for (auto& v : w) v = lim * uni();
      }
      check(mi355asr_load_weight(gpu_, name.c_str(), w.data(), rank, dims));
    }
  }

  mi355asr_model* gpu_ = nullptr;
  hipStream_t stream_ = nullptr;
  void* ws_ = nullptr;
  size_t ws_bytes_ = 0, wav_bytes_ = 0, ids_bytes_ = 0;
  float* d_wav_ = nullptr;
  int32_t *d_ids_ = nullptr, *d_len_ = nullptr;
};

}  // namespace ASR

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 2.0;
  try {
    ASR::Session session(1332);
    std::vector<float> wav((size_t)(16000 * seconds));
    for (size_t i = 0; i < wav.size(); ++i)
      wav[i] = 0.3f * std::sin(2.0 * M_PI * 220.0 * i / 16000.0) + 0.1f * std::sin(2.0 * M_PI * 1370.0 * i / 16000.0 + 0.001 * i);
    const std::vector<int> ids = session.Recognize(wav);
    const std::vector<int> again = session.Recognize(wav);
    std::printf("%s | %.1f s of audio -> %d encoder frames -> %zu tokens, repeatable: %s\n", mi355asr_version(), seconds,
                session.frames_for((int)wav.size()), ids.size(), ids == again ? "yes" : "NO");
    return ids == again ? 0 : 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
