// TEST INFRASTRUCTURE ONLY.  C-ABI shim around the reference's own prefix beam search
// (externals/ctc_decoders.zip: ctc_beam_search_decoder.cpp:18-187, decoder_utils.cpp, path_trie.cpp), compiled from
// the unpacked zip by oracle/Makefile (target ref_beam) against the header stand-ins in oracle/ref_stubs/ for the two
// un-vendored libraries (OpenFST, KenLM) that only the ext_scorer path uses.  ext_scorer is always nullptr here.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ctc_beam_search_decoder.h"

// link-time stand-ins for scorer.cpp (not compiled: needs KenLM); unreachable with ext_scorer == nullptr
double Scorer::get_log_cond_prob(const std::vector<std::string>&) { std::abort(); }
double Scorer::get_sent_log_prob(const std::vector<std::string>&) { std::abort(); }
std::vector<std::string> Scorer::make_ngram(PathTrie*) { std::abort(); }
std::vector<std::string> Scorer::split_labels(const std::vector<int>&) { std::abort(); }

// probs: [T][V] row-major doubles, V includes the blank (= last class).  Tokens are returned as ids by giving the
// reference a vocabulary of fixed-width 4-hex-digit strings.  Returns the number of hypotheses written.
extern "C" int ref_ctc_beam_search(const double* probs, int T, int V, int beam_size, double cutoff_prob,
                                   int cutoff_top_n, int max_len, double* scores, int* ids, int* lens) {
  std::vector<std::vector<double>> seq(T, std::vector<double>(V));
  for (int t = 0; t < T; ++t)
    for (int v = 0; v < V; ++v) seq[t][v] = probs[(size_t)t * V + v];
  std::vector<std::string> vocab;
  char buf[8];
  for (int v = 0; v < V - 1; ++v) {
    std::snprintf(buf, sizeof(buf), "%04x", v);
    vocab.push_back(buf);
  }
  auto res = ctc_beam_search_decoder(seq, vocab, (size_t)beam_size, cutoff_prob, (size_t)cutoff_top_n, nullptr);
  int n = 0;
  for (auto& r : res) {
    scores[n] = r.first;
    const int len = (int)r.second.size() / 4;
    lens[n] = len;
    for (int i = 0; i < len && i < max_len; ++i) ids[(size_t)n * max_len + i] = (int)std::strtol(r.second.substr(4 * i, 4).c_str(), nullptr, 16);
    ++n;
  }
  return n;
}

// ---- stateful BeamDecoder (ctc_beam_search_decoder.cpp:217-405): decode() continues from the kept prefix trie.
// Its vocabulary INCLUDES the blank as the last entry (blank_id = vocabulary.size() - 1, :238-240).
extern "C" void* ref_beam_decoder_new(int V, int beam_size, double cutoff_prob, int cutoff_top_n) {
  std::vector<std::string> vocab;
  char buf[8];
  for (int v = 0; v < V; ++v) {
    std::snprintf(buf, sizeof(buf), "%04x", v);
    vocab.push_back(buf);
  }
  return new BeamDecoder(vocab, (size_t)beam_size, cutoff_prob, (size_t)cutoff_top_n, nullptr);
}
extern "C" void ref_beam_decoder_free(void* h) { delete static_cast<BeamDecoder*>(h); }
extern "C" void ref_beam_decoder_reset(void* h) { static_cast<BeamDecoder*>(h)->reset(); }
extern "C" int ref_beam_decoder_decode(void* h, const double* probs, int T, int V, int max_len, double* scores,
                                       int* ids, int* lens) {
  std::vector<std::vector<double>> seq(T, std::vector<double>(V));
  for (int t = 0; t < T; ++t)
    for (int v = 0; v < V; ++v) seq[t][v] = probs[(size_t)t * V + v];
  auto res = static_cast<BeamDecoder*>(h)->decode(seq);
  int n = 0;
  for (auto& r : res) {
    scores[n] = r.first;
    const int len = (int)r.second.size() / 4;
    lens[n] = len;
    for (int i = 0; i < len && i < max_len; ++i) ids[(size_t)n * max_len + i] = (int)std::strtol(r.second.substr(4 * i, 4).c_str(), nullptr, 16);
    ++n;
  }
  return n;
}
