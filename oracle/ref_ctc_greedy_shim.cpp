// TEST INFRASTRUCTURE ONLY.  C-ABI shim around the reference's own header-only greedy decoder
// (Inference/CppInference/onnx/src/core/ctc_greedy_decoder.h:5-44).  The header is compiled from
// where it lies under /root/reference (see oracle/Makefile, target `ref`); nothing is copied.
#include "ctc_greedy_decoder.h"

extern "C" int ref_ctc_greedy(const float* probs, int T, int V, int blank, int* out) {
  std::vector<float> p(probs, probs + (size_t)T * V);
  std::vector<int> r = ctc_greedy_decoder(p, blank, V);
  for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
  return (int)r.size();
}
