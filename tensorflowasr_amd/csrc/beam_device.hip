// CTC prefix beam search ON THE DEVICE (round 2): one workgroup per utterance walks the frames; the host search of
// beam.hip (bit-exact against the reference's own decoder) remains the specification and the fallback.
//
// Reference: externals/ctc_decoders.zip -- ctc_beam_search_decoder.cpp:18-187, decoder_utils.cpp:7-38 (pruning),
// decoder_utils.h:41-49 (log_sum_exp<float>), decoder_utils.cpp:137-147 (prefix_compare), path_trie.cpp:37-147.
//
// Why this parallelises although the reference is a pointer-chasing trie walk.  After a frame's pruning, a trie node is
// either in the beam or it does not exist (removed nodes are reset when they are reached again, path_trie.cpp:54-60), so
// the state is the beam itself: per entry (prefix identity, parent identity, last character, score, p_blank, p_non_blank).
// In one frame a node's non-blank probability receives at most two contributions -- its own repetition and the extension of
// its parent by its last character -- and log_sum_exp of two values is symmetric, so the order in which the reference
// visits (character, prefix) pairs does not matter.  A frame is therefore:
//   1. pruned candidates (top-n list of topn_kernel; cumulative-probability cut in double, log(p + FLT_MIN) in double)
//   2. every beam entry in parallel: blank update, repetition, extension by its parent if that is in the beam (prefix
//      identities are 64-bit hashes of (parent identity, character), compared against the beam)
//   3. every (beam entry, candidate) pair that is not an existing beam entry: a fresh child with p_non_blank = the
//      extension term
//   4. keep the `beam` best of the <= beam x (n + 1) nodes: radix select on 64-bit keys (score descending, then character
//      ascending -- prefix_compare -- then slot), early exit as soon as the selected bin is wanted whole
//   5. compact into the next beam; a kept child appends (parent link, character) to a back-pointer arena in HBM.
// After the last frame the beam is ranked (prefix_compare) and the paths are read back through the arena.
// float32 scores as the reference; expf / logf are evaluated in double and rounded to float (the host's libm float
// functions are correctly rounded in all but ~1e-8 of the cases, so is that), which reproduces the host search bit for bit
// on the known-answer vectors.  Ties of (score, character) between different prefixes -- the reference leaves their order
// to std::nth_element -- are broken by slot (existing entries first, then children in (entry, candidate) order).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "beam.h"

namespace {

constexpr int NT = 256;          // threads per utterance
constexpr int BMAX = 128;        // beam entries
constexpr int NMAX = 40;         // candidates per frame (cutoff_top_n)
constexpr int SMAX = BMAX * (NMAX + 1);
constexpr float kNegInf = -FLT_MAX;

typedef unsigned long long u64;

__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float logf_cr(float x) { return (float)log((double)x); }
__device__ __forceinline__ float lse(float x, float y) {        // decoder_utils.h:41-49 with T = float
  if (x <= kNegInf) return y;
  if (y <= kNegInf) return x;
  const float m = fmaxf(x, y);
  return logf_cr(expf_cr(x - m) + expf_cr(y - m)) + m;
}
__device__ __forceinline__ u64 mix(u64 parent, int c) {        // prefix identity
  u64 z = parent * 0x9E3779B97F4A7C15ull + (u64)(unsigned)(c + 2) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z ? z : 1;
}
// ascending key order = better first: score descending, character ascending, slot ascending
__device__ __forceinline__ u64 make_key(float score, int ch, int slot) {
  unsigned u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);              // larger float -> larger u
  return ((u64)(~u) << 32) | ((u64)(unsigned)(ch + 1) << 16) | (u64)(unsigned)slot;
}

struct Beam {
  u64 id[BMAX], par[BMAX];
  int ch[BMAX], arena[BMAX];
  float score[BMAX], b[BMAX], nb[BMAX];
};

// block-wide exclusive scan of one int per thread (NT = 256 = 4 waves); returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_scan(int v, int* wave_tot, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int t = wave_tot[w];
    if (w < wv) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(NT) void beam_search_kernel(BeamDeviceArgs a) {
  __shared__ Beam beams[2];
  __shared__ u64 keys[SMAX];
  __shared__ u64 exist_mask[BMAX];
  __shared__ float cb[BMAX], cnb[BMAX], cscore[BMAX];
  __shared__ int ccand[NMAX];
  __shared__ float clp[NMAX];
  __shared__ int hist[256];
  __shared__ int wave_tot[NT / 64];
  __shared__ int sh_ncand, sh_blank, sh_digit, sh_need, sh_done;
  __shared__ u64 sh_prefix;

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T, N = a.N, V = a.V, beam = a.beam;
  const int frames = a.in_len ? max(0, min(a.in_len[b], T)) : T;
  int2* arena = a.arena + (size_t)b * ((size_t)T * beam + 1);
  int cur = 0, nbm = 1;                     // current beam buffer, number of entries
  if (tid == 0) {
    Beam& B0 = beams[0];
    B0.id[0] = 1; B0.par[0] = 0; B0.ch[0] = -1; B0.arena[0] = 0;
    B0.score[0] = 0.f; B0.b[0] = 0.f; B0.nb[0] = kNegInf;
    arena[0] = make_int2(-1, -1);
  }
  __syncthreads();

  // the top-n list of frame t + 1 is requested while frame t is processed (a dependent global load per frame would cost
  // more than the rest of the frame)
  __shared__ float praw[NMAX];
  float p_nx = 0.f;
  int c_nx = 0;
  if (tid < N && frames > 0) {
    p_nx = a.top_p[(size_t)b * T * N + tid];
    c_nx = a.top_idx[(size_t)b * T * N + tid];
  }
  for (int t = 0; t < frames; ++t) {
    const Beam& C = beams[cur];
    Beam& Nx = beams[cur ^ 1];
    // ---- 1. pruned candidates (decoder_utils.cpp:7-38 on the descending top-n list)
    if (tid < N) {
      praw[tid] = p_nx;
      ccand[tid] = c_nx;
      if (t + 1 < frames) {
        const size_t fn = ((size_t)b * T + t + 1) * N + tid;
        p_nx = a.top_p[fn];
        c_nx = a.top_idx[fn];
      }
    }
    if (tid < BMAX) exist_mask[tid] = 0;
    __syncthreads();
    if (tid == 0) {
      double cum = 0.0;
      int n = 0;
      for (int i = 0; i < N; ++i) {
        cum += (double)praw[i];
        ++n;
        if (cum >= a.cutoff_prob || n >= a.cutoff_top_n) break;
      }
      sh_ncand = n;
    }
    if (tid == 64) sh_blank = -1;
    __syncthreads();
    const int nc = sh_ncand;
    if (tid < nc) {
      clp[tid] = (float)log((double)praw[tid] + (double)FLT_MIN);
      if (ccand[tid] == V - 1) sh_blank = tid;
    }
    __syncthreads();
    const int kb = sh_blank;
    // ---- 2. existing entries
    if (tid < nbm) {
      const int i = tid;
      const int ci = C.ch[i];
      float bc = kNegInf, nbc = kNegInf;
      if (kb >= 0) bc = lse(bc, clp[kb] + C.score[i]);
      int kc = -1;
      for (int k = 0; k < nc; ++k)
        if (ccand[k] == ci) kc = k;
      if (kc >= 0) {
        nbc = lse(nbc, clp[kc] + C.nb[i]);
        const u64 pid = C.par[i];
        int j = -1;
        for (int jj = 0; jj < nbm; ++jj)
          if (C.id[jj] == pid) j = jj;
        if (j >= 0) {
          float lp = kNegInf;
          if (ci == C.ch[j] && C.b[j] > kNegInf) lp = clp[kc] + C.b[j];
          else if (ci != C.ch[j]) lp = clp[kc] + C.score[j];
          nbc = lse(nbc, lp);
          atomicOr(&exist_mask[j], 1ull << kc);
        }
      }
      cb[i] = bc;
      cnb[i] = nbc;
      cscore[i] = lse(bc, nbc);
    }
    __syncthreads();
    // ---- 3. keys: slots [0, nbm) = existing entries, nbm + j * nc + k = child of entry j by candidate k
    const int S = nbm + nbm * nc;
    int valid = 0;
    for (int s = tid; s < S; s += NT) {
      u64 key = ~0ull;
      if (s < nbm) {
        key = make_key(cscore[s], C.ch[s], s);
        ++valid;
      } else {
        const int j = (s - nbm) / nc, k = (s - nbm) - j * nc;
        const int c = ccand[k];
        if (k != kb && !((exist_mask[j] >> k) & 1ull)) {
          float lp = kNegInf;
          if (c == C.ch[j] && C.b[j] > kNegInf) lp = clp[k] + C.b[j];
          else if (c != C.ch[j]) lp = clp[k] + C.score[j];
          key = make_key(lp, c, s);
          ++valid;
        }
      }
      keys[s] = key;
    }
    int M;
    block_scan(valid, wave_tot, &M);
    // ---- 4. the `beam` best: threshold key by radix select (most significant byte first)
    u64 thr = ~0ull - 1;                      // M <= beam: every valid key (invalid ones are ~0)
    int shift_keep = 0;
    if (M > beam) {
      if (tid == 0) { sh_prefix = 0; sh_need = beam; sh_done = 0; }
      __syncthreads();
      int pass = 7;
      for (; pass >= 0; --pass) {
        const int shift = 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        const u64 pre = sh_prefix;
        for (int s = tid; s < S; s += NT) {
          const u64 key = keys[s];
          if (key != ~0ull && (pass == 7 || (key >> (shift + 8)) == (pre >> (shift + 8))))
            atomicAdd(&hist[(int)((key >> shift) & 255)], 1);
        }
        __syncthreads();
        const int h = hist[tid];
        const int need = sh_need;             // read before the scan's barriers: one thread rewrites it below
        int tot;
        const int ex = block_scan(h, wave_tot, &tot);
        if (ex < need && need <= ex + h) {    // the bin that holds the need-th smallest active key
          sh_digit = tid;
          sh_need = need - ex;
          sh_done = (need - ex == h);         // the whole bin is wanted: no need to resolve lower bytes
        }
        __syncthreads();
        if (tid == 0) sh_prefix = pre | ((u64)sh_digit << shift);
        __syncthreads();
        if (sh_done) break;
      }
      if (pass < 0) pass = 0;
      shift_keep = 8 * pass;
      thr = sh_prefix;
    }
    // ---- 5. compaction into the next beam (slot order: existing entries first)
    int keep_cnt = 0;
    const int per = (S + NT - 1) / NT;
    const int s0 = tid * per, s1 = min(S, s0 + per);
    for (int s = s0; s < s1; ++s) {
      const u64 key = keys[s];
      if (key != ~0ull && (key >> shift_keep) <= (thr >> shift_keep)) ++keep_cnt;
    }
    int newn;
    int pos = block_scan(keep_cnt, wave_tot, &newn);
    for (int s = s0; s < s1; ++s) {
      const u64 key = keys[s];
      if (key == ~0ull || (key >> shift_keep) > (thr >> shift_keep)) continue;
      if (s < nbm) {
        Nx.id[pos] = C.id[s]; Nx.par[pos] = C.par[s]; Nx.ch[pos] = C.ch[s]; Nx.arena[pos] = C.arena[s];
        Nx.score[pos] = cscore[s]; Nx.b[pos] = cb[s]; Nx.nb[pos] = cnb[s];
      } else {
        const int j = (s - nbm) / nc, k = (s - nbm) - j * nc;
        const int c = ccand[k];
        float lp = kNegInf;
        if (c == C.ch[j] && C.b[j] > kNegInf) lp = clp[k] + C.b[j];
        else if (c != C.ch[j]) lp = clp[k] + C.score[j];
        const int ai = 1 + t * beam + pos;
        arena[ai] = make_int2(C.arena[j], c);
        Nx.id[pos] = mix(C.id[j], c); Nx.par[pos] = C.id[j]; Nx.ch[pos] = c; Nx.arena[pos] = ai;
        Nx.score[pos] = lp; Nx.b[pos] = kNegInf; Nx.nb[pos] = lp;
      }
      ++pos;
    }
    __syncthreads();
    cur ^= 1;
    nbm = newn;
  }

  // ---- finish: rank by prefix_compare (+ slot), read the paths back
  const Beam& C = beams[cur];
  const int n = min(nbm, beam);
  int32_t* ids = a.ids + (size_t)b * beam * a.max_len;
  int32_t* lens = a.lens + (size_t)b * beam;
  float* scores = a.scores + (size_t)b * beam;
  if (tid == 0) a.n_hyp[b] = n;
  __threadfence_block();
  for (int i = tid; i < beam; i += NT) {
    if (i >= nbm) continue;
    const u64 ki = make_key(C.score[i], C.ch[i], i);
    int rank = 0;
    for (int j = 0; j < nbm; ++j) rank += make_key(C.score[j], C.ch[j], j) < ki;
    if (rank >= beam) continue;
    int len = 0;
    for (int p = C.arena[i]; p > 0; p = arena[p].x) ++len;
    lens[rank] = len;
    scores[rank] = C.score[i];
    int32_t* row = ids + (size_t)rank * a.max_len;
    for (int q = len; q < a.max_len; ++q) row[q] = -1;
    int q = len - 1;
    for (int p = C.arena[i]; p > 0; p = arena[p].x, --q)
      if (q < a.max_len) row[q] = arena[p].y;
  }
  for (int i = n + tid; i < beam; i += NT) {
    lens[i] = 0;
    scores[i] = kNegInf;
    for (int q = 0; q < a.max_len; ++q) ids[(size_t)i * a.max_len + q] = -1;
  }
}

}  // namespace

extern "C" {

bool mi355asr_beam_device_applicable(int V, int N, int beam) { return beam >= 1 && beam <= BMAX && N >= 1 && N <= NMAX && V <= 65534; }

size_t mi355asr_beam_device_ws_bytes(int B, int T, int beam, int max_len) {
  const size_t arena = (size_t)B * ((size_t)T * beam + 1) * sizeof(int2);
  const size_t out = (size_t)B * beam * ((size_t)max_len * 4 + 8) + (size_t)B * 4;
  return arena + out + 256;
}

int mi355asr_launch_beam_device(const BeamDeviceArgs* a, hipStream_t s) {
  hipLaunchKernelGGL(beam_search_kernel, dim3(a->B), dim3(NT), 0, s, *a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
