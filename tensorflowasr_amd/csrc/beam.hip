// CTC prefix beam search (scorer-less), the decode step of BASELINE config 5.
//
// Reference: externals/ctc_decoders.zip -- ctc_beam_search_decoder.cpp:18-187 (search), decoder_utils.cpp:7-38
// (get_pruned_log_probs), decoder_utils.h:41-49 (log_sum_exp), decoder_utils.cpp:137-147 (prefix_compare),
// path_trie.cpp:37-147 (trie), ctc_beam_search_decoder.cpp:426-459 (batch = thread pool over utterances).
//
// MI355X split of the work:
//   GPU  topn_kernel (this file): per frame softmax (when fed logits) + selection of the cutoff_top_n most probable
//        classes in descending order -- the only part of the algorithm that touches all V classes (9160 for the text
//        decoder).  HBM-bound: one read of the [frames, V] matrix.
//   GPU  beam_search_kernel (beam_device.hip): the prefix search, one workgroup per utterance, for beam <= 128 and
//        cutoff_top_n <= 40.
//   CPU  the same search on host threads (this file), one utterance per task like the reference's ThreadPool: the
//        specification of the device search (pinned to the reference decoder's known-answer vectors), the path for
//        host-resident probabilities, un-pruned mode and the stateful streaming decoder, and the fallback for wider
//        beams.  Same arithmetic as the reference: float32 trie scores, log(p + FLT_MIN) taken in double,
//        log_sum_exp<float>.
// The trie is an index-based arena (no per-node new/delete); the beam is advanced from the touched set instead of a
// whole-trie DFS per frame (same resulting set: every existing node is either in the beam or was touched this frame).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "env.h"
#include <vector>

#include "beam.h"

namespace {

constexpr float kNegInf = -FLT_MAX;   // NUM_FLT_INF of the reference (decoder_utils.h:9)

inline float log_sum_exp(float x, float y) {   // decoder_utils.h:41-49, T = float
  if (x <= kNegInf) return y;
  if (y <= kNegInf) return x;
  const float m = std::max(x, y);
  return std::log(std::exp(x - m) + std::exp(y - m)) + m;
}

struct Node {
  int ch, parent, first_child, next_sibling;
  float b_prev, nb_prev, b_cur, nb_cur, score;
  int stamp;      // frame at which the node was last put on the candidate list
  int slot;       // this frame's tie-break word (Search::better)
  bool exists;
};

class Trie {
 public:
  std::vector<Node> nodes;
  std::vector<int> free_list;

  int make(int ch, int parent) {
    int id;
    if (!free_list.empty()) { id = free_list.back(); free_list.pop_back(); }
    else { id = (int)nodes.size(); nodes.emplace_back(); }
    Node& n = nodes[id];
    n.ch = ch; n.parent = parent; n.first_child = -1; n.next_sibling = -1;
    n.b_prev = n.nb_prev = n.b_cur = n.nb_cur = n.score = kNegInf;
    n.stamp = -1; n.slot = 0; n.exists = true;
    return id;
  }
  // PathTrie::get_path_trie (path_trie.cpp:37-91, dictionary-less branch)
  int child(int p, int c) {
    for (int k = nodes[p].first_child; k >= 0; k = nodes[k].next_sibling) {
      if (nodes[k].ch == c) {
        Node& n = nodes[k];
        if (!n.exists) {
          n.exists = true;
          n.b_prev = n.nb_prev = n.b_cur = n.nb_cur = kNegInf;
        }
        return k;
      }
    }
    const int id = make(c, p);
    nodes[id].next_sibling = nodes[p].first_child;
    nodes[p].first_child = id;
    return id;
  }
  // PathTrie::remove (path_trie.cpp:129-147)
  void remove(int id) {
    nodes[id].exists = false;
    while (id > 0 && nodes[id].first_child < 0 && !nodes[id].exists) {
      const int p = nodes[id].parent;
      int* link = &nodes[p].first_child;
      while (*link != id) link = &nodes[*link].next_sibling;
      *link = nodes[id].next_sibling;
      free_list.push_back(id);
      id = p;
    }
  }
};

struct Cand { int c; float lp; };

// decoder_utils.cpp:7-38 on a full probability row
void pruned_from_row(const float* prob, int V, double cutoff_prob, int cutoff_top_n, std::vector<Cand>& out,
                     std::vector<int>& order) {
  out.clear();
  int n = V;
  order.resize(V);
  for (int i = 0; i < V; ++i) order[i] = i;
  if (cutoff_prob < 1.0 || cutoff_top_n < n) {
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return prob[a] > prob[b]; });
    if (cutoff_prob < 1.0) {
      double cum = 0.0;
      n = 0;
      for (int i = 0; i < V; ++i) {
        cum += (double)prob[order[i]];
        ++n;
        if (cum >= cutoff_prob || n >= cutoff_top_n) break;
      }
    }
  }
  for (int i = 0; i < n; ++i) out.push_back({order[i], (float)std::log((double)prob[order[i]] + (double)FLT_MIN)});
}

// same, from the GPU's top-n list (already in descending order); only valid when cutoff_prob < 1
void pruned_from_topn(const int32_t* idx, const float* p, int N, double cutoff_prob, int cutoff_top_n,
                      std::vector<Cand>& out) {
  out.clear();
  double cum = 0.0;
  int n = 0;
  for (int i = 0; i < N; ++i) {
    cum += (double)p[i];
    ++n;
    if (cum >= cutoff_prob || n >= cutoff_top_n) break;
  }
  for (int i = 0; i < n; ++i) out.push_back({idx[i], (float)std::log((double)p[i] + (double)FLT_MIN)});
}

struct Search {
  Trie trie;
  std::vector<int> prefixes, touched;
  int beam;
  int blank;

  void reset(int beam_size, int blank_id) {
    beam = beam_size;
    blank = blank_id;
    trie.nodes.clear();
    trie.free_list.clear();
    const int root = trie.make(-1, -1);
    trie.nodes[root].score = trie.nodes[root].b_prev = 0.0f;
    prefixes.assign(1, root);
  }

  // prefix_compare (decoder_utils.cpp:137-147): score descending, then last character ascending -- and nothing else: the
  // reference leaves prefixes that agree in both to std::nth_element / std::sort.  That is not a corner case: at scores of
  // -1000 ... -5000 a float32 ulp is 1e-4 ... 5e-4, neighbouring hypotheses collide in their scores, and two tied parents
  // extended by the same character give tied children with the same last character (16 x 30 s, utterance 11, frame 138).
  // The order is therefore DEFINED here, in terms both this search and the device search (beam_device.hip) can evaluate:
  // the beam is kept sorted; among equals an entry that was in the beam comes before a new child, beam entries in the
  // order of their beam positions, children in the order of (parent's beam position, candidate's position in the frame's
  // descending list).  slot = that rank: position for beam entries, nbm + parent position * candidates + candidate position.
  bool better(int x, int y) const {
    const Node &a = trie.nodes[x], &b = trie.nodes[y];
    if (a.score != b.score) return a.score > b.score;
    if (a.ch != b.ch) return a.ch < b.ch;
    return a.slot < b.slot;
  }

  void step(int t, const std::vector<Cand>& cands) {
    touched.clear();
    const int nbm = (int)prefixes.size(), nc = (int)cands.size();
    for (int i = 0; i < nbm; ++i) {
      trie.nodes[prefixes[i]].stamp = t;
      trie.nodes[prefixes[i]].slot = i;
    }
    for (int k = 0; k < nc; ++k) {
      const int c = cands[k].c;
      const float lp = cands[k].lp;
      for (int i = 0; i < nbm; ++i) {
        const int pi = prefixes[i];
        if (c == blank) {
          Node& p = trie.nodes[pi];
          p.b_cur = log_sum_exp(p.b_cur, lp + p.score);
          continue;
        }
        if (c == trie.nodes[pi].ch) {
          Node& p = trie.nodes[pi];
          p.nb_cur = log_sum_exp(p.nb_cur, lp + p.nb_prev);
        }
        const int qi = trie.child(pi, c);     // may reallocate trie.nodes
        const Node& p = trie.nodes[pi];
        Node& q = trie.nodes[qi];
        float log_p = kNegInf;
        if (c == p.ch && p.b_prev > kNegInf) log_p = lp + p.b_prev;
        else if (c != p.ch) log_p = lp + p.score;
        q.nb_cur = log_sum_exp(q.nb_cur, log_p);
        if (q.stamp != t) { q.stamp = t; q.slot = nbm + i * nc + k; touched.push_back(qi); }
      }
    }
    // PathTrie::iterate_to_vec (path_trie.cpp:113-127) over every existing node = beam + touched
    prefixes.insert(prefixes.end(), touched.begin(), touched.end());
    for (int id : prefixes) {
      Node& n = trie.nodes[id];
      n.b_prev = n.b_cur;
      n.nb_prev = n.nb_cur;
      n.b_cur = n.nb_cur = kNegInf;
      n.score = log_sum_exp(n.b_prev, n.nb_prev);
    }
    if ((int)prefixes.size() >= beam) {
      if ((int)prefixes.size() > beam)
        std::nth_element(prefixes.begin(), prefixes.begin() + beam, prefixes.end(),
                         [&](int x, int y) { return better(x, y); });
      for (size_t i = beam; i < prefixes.size(); ++i) trie.remove(prefixes[i]);
      prefixes.resize(beam);
    }
    std::sort(prefixes.begin(), prefixes.end(), [&](int x, int y) { return better(x, y); });   // beam positions = ranks
  }

  // -> number of hypotheses written
  int finish(int max_len, int32_t* ids, int32_t* lens, float* scores) {
    std::sort(prefixes.begin(), prefixes.end(), [&](int x, int y) { return better(x, y); });
    const int n = std::min((int)prefixes.size(), beam);
    std::vector<int> path;
    for (int i = 0; i < n; ++i) {
      path.clear();
      for (int id = prefixes[i]; trie.nodes[id].ch != -1; id = trie.nodes[id].parent) path.push_back(trie.nodes[id].ch);
      std::reverse(path.begin(), path.end());
      lens[i] = (int32_t)path.size();
      scores[i] = trie.nodes[prefixes[i]].score;
      for (int j = 0; j < max_len; ++j) ids[(size_t)i * max_len + j] = j < (int)path.size() ? path[j] : -1;
    }
    for (int i = n; i < beam; ++i) {
      lens[i] = 0;
      scores[i] = kNegInf;
      for (int j = 0; j < max_len; ++j) ids[(size_t)i * max_len + j] = -1;
    }
    return n;
  }
};

template <class RowFn>
void run_batch(int B, int num_threads, RowFn&& per_utt) {
  num_threads = std::max(1, std::min(num_threads, B));
  if (num_threads == 1) {
    for (int b = 0; b < B; ++b) per_utt(b);
    return;
  }
  std::vector<std::thread> pool;
  for (int w = 0; w < num_threads; ++w)
    pool.emplace_back([&, w]() {
      for (int b = w; b < B; b += num_threads) per_utt(b);
    });
  for (auto& th : pool) th.join();
}

// ---------------------------------------------------------------------------------------------------------
// GPU: per-frame top-n.  One wave per frame, the frame's V values staged in LDS.  Output order = the host's
// stable_sort: value descending, class ascending among equal values.
//
// Fast path (N <= 64): the classes are partitioned over the 64 lanes.  Exactly N lanes own a value >= T0 := the N-th largest of
// the 64 lane maxima, so at least N classes are >= T0, and for independent values only ~1.5 N are.  One pass collects the
// classes > T0 (all of them) and the first N classes == T0 in class order (ballot compaction keeps class order); the
// final order comes from counting, for every collected class, the collected classes that precede it.  ~3 passes over the
// row instead of N.
// Slow path (N > 64, or more than GT_CAP classes above T0 -- only adversarial rows): N rounds of (per-lane scan, wave
// arg-max with lowest-class tie break, knock-out).
// ---------------------------------------------------------------------------------------------------------
constexpr int GT_CAP = 256, EQ_CAP = 64;

__device__ __forceinline__ void topn_rounds(float* sh, int V, int N, int lane, bool is_logits, float mx, float denom,
                                            int32_t* out_idx, float* out_p) {
  for (int n = 0; n < N; ++n) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < V; i += 64) {
      const float v = sh[i];
      if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float ov = __shfl_xor(best, off);
      const int oi = __shfl_xor(bi, off);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
      out_idx[n] = bi;
      out_p[n] = is_logits ? __expf(best - mx) / denom : best;
    }
    if (bi != 0x7fffffff && (bi & 63) == lane) sh[bi] = -INFINITY;
  }
}

__global__ __launch_bounds__(64) void topn_kernel(const float* __restrict__ x, int V, int N, int is_logits,
                                                  int32_t* __restrict__ out_idx, float* __restrict__ out_p) {
  extern __shared__ __align__(16) float sh[];
  __shared__ float cv[GT_CAP + EQ_CAP];
  __shared__ int ci[GT_CAP + EQ_CAP];
  const int lane = threadIdx.x;
  const size_t frame = blockIdx.x;
  const float* row = x + frame * V;
  out_idx += frame * N;
  out_p += frame * N;
  // stage the row; mx = the maximum of what this lane staged (any partition of the classes over the lanes serves the
  // threshold below).  Several loads in flight per lane: with one wave per frame the pass is latency-bound otherwise.
  float mx = -INFINITY;
  if ((V & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const float4* row4 = reinterpret_cast<const float4*>(row);
    float4* sh4 = reinterpret_cast<float4*>(sh);
    const int V4 = V >> 2;
    for (int i0 = 0; i0 < V4; i0 += 64 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u + lane;
        v[u] = i < V4 ? row4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u + lane;
        if (i < V4) sh4[i] = v[u];
        mx = fmaxf(fmaxf(mx, fmaxf(v[u].x, v[u].y)), fmaxf(v[u].z, v[u].w));
      }
    }
  } else {
    for (int i0 = 0; i0 < V; i0 += 64 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u + lane;
        v[u] = i < V ? row[i] : -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u + lane;
        if (i < V) sh[i] = v[u];
        mx = fmaxf(mx, v[u]);
      }
    }
  }
  const float lane_max = mx;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float denom = 1.f;
  if (is_logits) {
    float s = 0.f;
#pragma unroll 8
    for (int i = lane; i < V; i += 64) s += __expf(sh[i] - mx);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    denom = s;
  }
  if (N > 64) {
    topn_rounds(sh, V, N, lane, is_logits, mx, denom, out_idx, out_p);
    return;
  }
  // T0: the lane maximum of rank N - 1 (ranks made distinct by the lane number)
  int rank = 0;
#pragma unroll
  for (int l = 0; l < 64; ++l) {
    const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_max), l));
    rank += (o > lane_max) || (o == lane_max && l < lane);
  }
  const unsigned long long pick = __ballot(rank == N - 1);
  const float T0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_max), (int)__builtin_ctzll(pick)));
  // collect: classes > T0 into [0, G), the first EQ_CAP classes == T0 into [GT_CAP, GT_CAP + E)
  const unsigned long long lt = (1ull << lane) - 1;
  int G = 0, E = 0;
#pragma unroll 4
  for (int i0 = 0; i0 < V; i0 += 64) {
    const int i = i0 + lane;
    const float v = i < V ? sh[i] : -INFINITY;
    const bool gt = i < V && v > T0, eq = i < V && v == T0;
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    if ((bg | be) == 0) continue;
    const int pg = G + __popcll(bg & lt), pe = E + __popcll(be & lt);
    if (gt && pg < GT_CAP) { cv[pg] = v; ci[pg] = i; }
    if (eq && pe < EQ_CAP) { cv[GT_CAP + pe] = v; ci[GT_CAP + pe] = i; }
    G += __popcll(bg);
    E = min(E + __popcll(be), EQ_CAP);
  }
  if (G > GT_CAP) {
    topn_rounds(sh, V, N, lane, is_logits, mx, denom, out_idx, out_p);
    return;
  }
  // candidates: [0, G) and [GT_CAP, GT_CAP + Eu); position of each in (value descending, class ascending) order
  const int Eu = min(E, N);
  const int M = G + Eu;
  for (int m0 = 0; m0 < M; m0 += 64) {
    const int m = m0 + lane;
    const int slot = m < G ? m : GT_CAP + (m - G);
    const float v = m < M ? cv[slot] : 0.f;
    const int id = m < M ? ci[slot] : 0;
    int before = 0;
#pragma unroll 8
    for (int q = 0; q < G; ++q) {
      const float ov = cv[q];
      const int oi = ci[q];
      before += (ov > v) || (ov == v && oi < id);
    }
#pragma unroll 8
    for (int q = 0; q < Eu; ++q) {
      const float ov = cv[GT_CAP + q];
      const int oi = ci[GT_CAP + q];
      before += (ov > v) || (ov == v && oi < id);
    }
    if (m < M && before < N) {
      out_idx[before] = id;
      out_p[before] = is_logits ? __expf(v - mx) / denom : v;
    }
  }
}


// The same selection with the frame's row in REGISTERS (V <= 64 NR): lane l holds classes l, l + 64, ... -- the order the
// LDS kernel's soft-max sum walks them in, so the denominators are bit-identical.  No staging pass, no 36 KB of LDS per
// wave (four waves per CU, each alone on its SIMD: 0.96 TB/s at V = 9 160), ~1 700 instructions per frame instead of three
// latency-bound sweeps over LDS.  More than GT_CAP classes above T0 (adversarial rows) are finished by rounds over the
// registers.
template <int NR>
__global__ __launch_bounds__(64) void topn_reg_kernel(const float* __restrict__ x, int V, int N, int is_logits,
                                                      int32_t* __restrict__ out_idx, float* __restrict__ out_p) {
  __shared__ float cv[GT_CAP + EQ_CAP];
  __shared__ int ci[GT_CAP + EQ_CAP];
  const int lane = threadIdx.x;
  const size_t frame = blockIdx.x;
  const float* row = x + frame * V;
  out_idx += frame * N;
  out_p += frame * N;
  // The row through buffer loads (per-request 64-bit addresses would double the register count).  The whole offset
  // lane * 4 + m * 256 goes into the bounds-CHECKED operand (voffset + the instruction's immediate; an soffset would be
  // excluded from the check and let the last registers of a short row read up to 6 KB past the tensor): reads past the
  // row return 0 and are turned into -inf.  Classes >= V are then never above T0 and
  // come behind every real class in class order -- with V >= 64 >= N (the launch checks) they are never selected, so
  // nothing below needs an `i < V` test.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, V * 4, 0x00020000);
  float v[NR];
#pragma unroll
  for (int m = 0; m < NR; ++m) v[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + m * 256, 0, 0));
  const int vfull = V >> 6;                 // registers every lane of which holds a class
#pragma unroll
  for (int m = 0; m < NR; ++m)
    if (m >= vfull && lane + 64 * m >= V) v[m] = -INFINITY;
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < NR; ++m) mx = fmaxf(mx, v[m]);
  const float lane_max = mx;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float denom = 1.f;
  if (is_logits) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < NR; ++m) s += __expf(v[m] - mx);        // + 0 for the padding: the LDS kernel's sum, bit for bit
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    denom = s;
  }
  int rank = 0;
#pragma unroll
  for (int l = 0; l < 64; ++l) {
    const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_max), l));
    rank += (o > lane_max) || (o == lane_max && l < lane);
  }
  const unsigned long long pick = __ballot(rank == N - 1);
  const float T0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_max), (int)__builtin_ctzll(pick)));
  const unsigned long long lt = (1ull << lane) - 1;
  int G = 0, E = 0;
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const bool gt = v[m] > T0, eq = v[m] == T0;
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    if ((bg | be) == 0) continue;
    int lane_here = lane;                   // opaque: NR class numbers computed ahead of the branches would cost NR registers
    asm volatile("" : "+v"(lane_here));
    const int i = lane_here + 64 * m;
    const int pg = G + __popcll(bg & lt), pe = E + __popcll(be & lt);
    if (gt && pg < GT_CAP) { cv[pg] = v[m]; ci[pg] = i; }
    if (eq && pe < EQ_CAP) { cv[GT_CAP + pe] = v[m]; ci[GT_CAP + pe] = i; }
    G += __popcll(bg);
    E = min(E + __popcll(be), EQ_CAP);
  }
  if (G > GT_CAP) {                         // rounds of (lane arg-max, wave arg-max with lowest-class tie break, knock-out)
    for (int n = 0; n < N; ++n) {
      float best = -INFINITY;
      int bm = -1;                          // register of the lane's best (no per-register class numbers kept live)
#pragma unroll
      for (int m = 0; m < NR; ++m)
        if (v[m] > best) { best = v[m]; bm = m; }
      int bi = bm >= 0 ? lane + 64 * bm : 0x7fffffff;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) {
        out_idx[n] = bi;
        out_p[n] = is_logits ? __expf(best - mx) / denom : best;
      }
      const int km = __builtin_amdgcn_readfirstlane(bi >> 6);
      const bool me = (bi & 63) == lane;
#pragma unroll
      for (int m = 0; m < NR; ++m)
        if (m == km && me) v[m] = -INFINITY;
    }
    return;
  }
  const int Eu = min(E, N);
  const int M = G + Eu;
  for (int m0 = 0; m0 < M; m0 += 64) {
    const int m = m0 + lane;
    const int slot = m < G ? m : GT_CAP + (m - G);
    const float val = m < M ? cv[slot] : 0.f;
    const int id = m < M ? ci[slot] : 0;
    int before = 0;
#pragma unroll 8
    for (int q = 0; q < G; ++q) {
      const float ov = cv[q];
      const int oi = ci[q];
      before += (ov > val) || (ov == val && oi < id);
    }
#pragma unroll 8
    for (int q = 0; q < Eu; ++q) {
      const float ov = cv[GT_CAP + q];
      const int oi = ci[GT_CAP + q];
      before += (ov > val) || (ov == val && oi < id);
    }
    if (m < M && before < N) {
      out_idx[before] = id;
      out_p[before] = is_logits ? __expf(val - mx) / denom : val;
    }
  }
}

}  // namespace

extern "C" {

int mi355asr_beam_host_impl(const float* probs, const int32_t* in_len, int B, int T, int V, int beam_size,
                            double cutoff_prob, int cutoff_top_n, int num_threads, int max_len, int32_t* ids,
                            int32_t* lens, float* scores, int32_t* n_hyp) {
  run_batch(B, num_threads, [&](int b) {
    Search s;
    s.reset(beam_size, V - 1);
    std::vector<Cand> cands;
    std::vector<int> order;
    const int n = in_len ? std::max(0, std::min(in_len[b], T)) : T;
    for (int t = 0; t < n; ++t) {
      pruned_from_row(probs + ((size_t)b * T + t) * V, V, cutoff_prob, cutoff_top_n, cands, order);
      s.step(t, cands);
    }
    n_hyp[b] = s.finish(max_len, ids + (size_t)b * beam_size * max_len, lens + (size_t)b * beam_size,
                        scores + (size_t)b * beam_size);
  });
  return 0;
}

int mi355asr_beam_topn_impl(const int32_t* top_idx, const float* top_p, const int32_t* in_len, int B, int T, int V,
                            int N, int beam_size, double cutoff_prob, int cutoff_top_n, int num_threads, int max_len,
                            int32_t* ids, int32_t* lens, float* scores, int32_t* n_hyp) {
  run_batch(B, num_threads, [&](int b) {
    Search s;
    s.reset(beam_size, V - 1);
    std::vector<Cand> cands;
    const int n = in_len ? std::max(0, std::min(in_len[b], T)) : T;
    for (int t = 0; t < n; ++t) {
      const size_t o = ((size_t)b * T + t) * N;
      pruned_from_topn(top_idx + o, top_p + o, N, cutoff_prob, cutoff_top_n, cands);
      s.step(t, cands);
    }
    n_hyp[b] = s.finish(max_len, ids + (size_t)b * beam_size * max_len, lens + (size_t)b * beam_size,
                        scores + (size_t)b * beam_size);
  });
  return 0;
}

int mi355asr_launch_topn(const float* x_dev, int frames, int V, int N, int is_logits, int32_t* idx_dev, float* p_dev,
                         hipStream_t s) {
  static const bool reg_off = mi355_env("MI355ASR_TOPN_REG", -1) == 0;
  if (N <= 64 && V >= 64 && V <= 64 * 144 && !reg_off) {
    const int nr = (V + 63) / 64;
#define TOPN_REG(NR) hipLaunchKernelGGL(topn_reg_kernel<NR>, dim3(frames), dim3(64), 0, s, x_dev, V, N, is_logits, idx_dev, p_dev)
    if (nr <= 24) TOPN_REG(24);
    else if (nr <= 64) TOPN_REG(64);
    else if (nr <= 96) TOPN_REG(96);
    else TOPN_REG(144);
#undef TOPN_REG
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  const size_t lds = (size_t)V * sizeof(float);
  if (lds > 160 * 1024) return -1;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)topn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return -2;
  }
  hipLaunchKernelGGL(topn_kernel, dim3(frames), dim3(64), lds, s, x_dev, V, N, is_logits, idx_dev, p_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


// ---- stateful decoder (replaces: class BeamDecoder, ctc_beam_search_decoder.cpp:217-405, ext_scorer == nullptr) ----
struct BeamState {
  Search s;
  int V, beam, cutoff_top_n, frame;
  double cutoff_prob;
  std::vector<Cand> cands;
  std::vector<int> order;
};

void* mi355asr_beam_state_new(int V, int beam_size, double cutoff_prob, int cutoff_top_n) {
  auto* st = new BeamState();
  st->V = V; st->beam = beam_size; st->cutoff_prob = cutoff_prob; st->cutoff_top_n = cutoff_top_n; st->frame = 0;
  st->s.reset(beam_size, V - 1);       // BeamDecoder: blank_id = vocabulary.size() - 1 (:238-240)
  return st;
}
void mi355asr_beam_state_free(void* h) { delete static_cast<BeamState*>(h); }
void mi355asr_beam_state_reset(void* h) {
  auto* st = static_cast<BeamState*>(h);
  st->s.reset(st->beam, st->V - 1);
  st->frame = 0;
}
int mi355asr_beam_state_decode(void* h, const float* probs, int T, int max_len, int32_t* ids, int32_t* lens,
                               float* scores) {
  auto* st = static_cast<BeamState*>(h);
  for (int t = 0; t < T; ++t) {
    pruned_from_row(probs + (size_t)t * st->V, st->V, st->cutoff_prob, st->cutoff_top_n, st->cands, st->order);
    st->s.step(st->frame++, st->cands);          // the stamp must be unique over the decoder's lifetime
  }
  // BeamDecoder::decode sorts the beam before returning (:389-392); the order carries over to the next call
  return st->s.finish(max_len, ids, lens, scores);
}

}  // extern "C"
