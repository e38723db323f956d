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
// float32 scores as the reference; expf / logf / log are the host C library's own evaluation (refmath.h -- glibc's float
// routines are NOT correctly rounded, so "the exact value rounded once" is a different function), which reproduces the
// host search bit for bit: known-answer vectors and the benched 16 x 30 s shape alike.  Ties of (score, character) between
// different prefixes -- the reference leaves their order to std::nth_element -- are broken by slot (existing entries first
// in beam order, then children in (entry, candidate) order) with the beam kept in rank order: the order beam.hip's
// Search::better defines, so both searches resolve them alike (float32 scores of -1000 ... -5000 tie all the time).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>

#include "beam.h"
#include "refmath.h"

namespace {

constexpr int NT = 256;          // threads per utterance: beams up to SMALL_BEAM (the one-key-per-thread path)
constexpr int NTW = 512;         // ... wider beams (round 4): eight waves walk the (entry, candidate) keys of the radix path, two per SIMD
constexpr int BMAX = 128;        // beam entries
constexpr int NMAX = 40;         // candidates per frame (cutoff_top_n)
constexpr int HASH = 1024;       // cells of the parent hash
static_assert(HASH == 4 * NT && 4 * BMAX <= HASH, "one int4 of cells per thread of the first four waves to clear; the set stays sparse");
constexpr int SMALL_BEAM = 16;
__host__ __device__ constexpr int tab_stride(int V) { return (V + 15) & ~15; }   // bytes of one class table   // widest beam of the one-key-per-thread kernel
constexpr float kNegInf = -FLT_MAX;

typedef unsigned long long u64;

// expf / logf / log as the HOST's libm returns them (refmath.h: glibc's own evaluation, restated operation by operation
// and proven equal to the installed library on every argument the search can form).  Rounds 2-3 evaluated the exact
// functions and rounded once; glibc's float routines are not correctly rounded (logf(1 + expf(d)) differs for 0.5 % of
// all d), which made the device's scores drift from the host search's by an ulp every few hundred calls and reordered
// beam entries on long utterances (round-3 verdict: 78 of 88 hypotheses at 16 x 30 s).  The tables (2.5 KB) are copied to
// LDS once per workgroup: the per-lane index makes them a gather, which LDS serves at full rate.
__constant__ uint64_t kExp2fTab[refmath::kExp2fTabWords] = REFMATH_EXP2F_TAB;
__constant__ double kLogfTab[refmath::kLogfTabWords] = REFMATH_LOGF_TAB;
__constant__ double kLogTab[refmath::kLogTabWords] = REFMATH_LOG_TAB;
struct MathTabs {
  uint64_t exp2f[refmath::kExp2fTabWords];
  double logf[refmath::kLogfTabWords];
  double log[refmath::kLogTabWords];
};
__device__ __forceinline__ void load_math_tabs(MathTabs& m, int tid, int nthreads) {   // followed by a barrier at the caller
  for (int i = tid; i < refmath::kExp2fTabWords; i += nthreads) m.exp2f[i] = kExp2fTab[i];
  for (int i = tid; i < refmath::kLogfTabWords; i += nthreads) m.logf[i] = kLogfTab[i];
  for (int i = tid; i < refmath::kLogTabWords; i += nthreads) m.log[i] = kLogTab[i];
}
__device__ __forceinline__ float lse(const MathTabs& m, float x, float y) {        // decoder_utils.h:41-49 with T = float
  if (x <= kNegInf) return y;
  if (y <= kNegInf) return x;
  // logf(expf(x - m) + expf(y - m)) + m: the larger argument contributes expf(0) = 1 exactly; below exp(-17.5) < 2^-25
  // the other one is absorbed by the float addition whatever its last bit, and logf(1) = 0
  const float mx = fmaxf(x, y), d = fminf(x, y) - mx;
  if (d < -17.5f) return 0.f + mx;
  return refmath::ref_logf(1.0f + refmath::ref_expf(d, m.exp2f), m.logf) + mx;
}
// inclusive prefix sum over the wave: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then lane 15 of a row into the next
// row and lane 31 into the upper half (DPP; lanes without a source add 0)
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWS, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_prefix_sum(double v) {
  v += dpp_f64<0x111, 0xf>(v);
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);                                    // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xc>(v);                                    // row_bcast:31 into rows 2 and 3
  return v;
}
// Prefix identity: one 64-bit multiply on the per-frame path (four of them cost ~250 clocks of every frame's install
// step).  The shift-xor in front keeps the ids from being a polynomial in the characters modulo 2^64.
__device__ __forceinline__ u64 mix(u64 parent, int c) {
  const u64 z = (parent ^ (parent >> 29)) * 0x9E3779B97F4A7C15ull + (u64)(unsigned)(c + 2);
  return z ? z : 1;
}
// ascending key order = better first: score descending, character ascending, slot ascending
__device__ __forceinline__ u64 make_key(float score, int ch, int slot) {
  unsigned u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);              // larger float -> larger u
  return ((u64)(~u) << 32) | ((u64)(unsigned)(ch + 1) << 16) | (u64)(unsigned)slot;
}

struct Beam {
  __align__(16) u64 id[BMAX];
  u64 par[BMAX];
  int ch[BMAX], arena[BMAX];
  int kc[BMAX];        // position of ch among the candidates of the frame this beam is extended with (-1: not one)
  float score[BMAX], b[BMAX], nb[BMAX];
};

// block-wide exclusive scan of one int per thread (NW waves); returns the exclusive prefix, *total = sum.
// One barrier: callers alternate `wave_tot` buffers (or have another barrier before the same one is written again).
template <int NW>
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
  for (int w = 0; w < NW; ++w) {
    const int t = wave_tot[w];
    if (w < wv) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

// Frame candidates (decoder_utils.cpp:7-38 on the descending top-n list), prepared by the last wave one frame ahead of the
// search: they do not depend on the beam.
struct Cands {
  __align__(16) int c[NMAX];
  float lp[NMAX];
  int n, blank;        // number of candidates kept, position of the blank among them (-1: pruned)
};

struct Shared {
  Beam beams[2];
  MathTabs math;
  __align__(16) u64 keys[NT];   // SMALL path: one key per thread
  u64 exist_mask[2][BMAX];      // [frame parity][entry]: bit k = the child by candidate k is a beam entry itself
  float cb[BMAX], cnb[BMAX], cscore[BMAX];
  Cands cands[2];
  int hist[2][256];
  int wave_tot[4][NTW / 64];
  int digit, need, done, thr_ok;
  float max_score;
  u64 diff, kmin[2];
  int kept_j[BMAX];
  __align__(16) int hpos[2][HASH];   // [beam buffer]: open-addressed set of 1 + entry position, keyed by id mod HASH (0: free cell)
};
// Two lookups replace list scans on the per-frame path (a thread scanning 40 candidates and up to 128 ids cost 2 500 of
// the 3 400 clocks of the entry phase):
//   class table (dynamic LDS, V bytes per frame parity): tab[c] = 1 + position of class c among the frame's candidates,
//     written by the wave that prepares the candidates (which also clears the entries of the frame two back); an entry's
//     `kc` is read from it when the entry is installed, one frame ahead.
//   parent hash: hpos above, linear probing; an entry claims the first free cell from id mod HASH on when it is installed
//     (atomicCAS), a lookup walks from there until the id matches or a cell is free.  At most 2 x BMAX of the 1 024 cells
//     are ever taken (the small path may install a frame twice: stale cells fail the id test and are walked over).
struct Lookup {
  const unsigned char* tab_next;   // class table of the frame the new beam meets
  int* hnx;                        // parent hash of the new beam
};

// extension of entry j by candidate k (ctc_beam_search_decoder.cpp:99-113)
__device__ __forceinline__ float child_lp(const Beam& C, const Cands& K, int j, int k) {
  const int c = K.c[k];
  if (c == C.ch[j]) return C.b[j] > kNegInf ? K.lp[k] + C.b[j] : kNegInf;
  return K.lp[k] + C.score[j];
}
__device__ __forceinline__ void hash_claim(int* cells, u64 id, int pos) {
  unsigned h = (unsigned)id & (HASH - 1);
  while (atomicCAS(&cells[h], 0, pos + 1) != 0) h = (h + 1) & (HASH - 1);
}
__device__ __forceinline__ int hash_find(const int* cells, const Beam& C, int nbm, u64 id) {
  for (unsigned h = (unsigned)id & (HASH - 1);; h = (h + 1) & (HASH - 1)) {
    const int v = cells[h];
    if (v == 0) return -1;
    if (v <= nbm && C.id[v - 1] == id) return v - 1;
  }
}
__device__ __forceinline__ void keep_entry(Shared& sh, const Beam& C, Beam& Nx, const Lookup& L, int s, int pos) {
  const u64 id = C.id[s];
  const int ch = C.ch[s];
  Nx.id[pos] = id; Nx.par[pos] = C.par[s]; Nx.ch[pos] = ch; Nx.arena[pos] = C.arena[s];
  Nx.score[pos] = sh.cscore[s]; Nx.b[pos] = sh.cb[s]; Nx.nb[pos] = sh.cnb[s];
  Nx.kc[pos] = ch >= 0 ? (int)L.tab_next[ch] - 1 : -1;
  hash_claim(L.hnx, id, pos);
}
__device__ __forceinline__ void keep_child(const Beam& C, Beam& Nx, const Lookup& L, int2* arena, int ai, int j, int c, float lp, int pos) {
  const u64 pid = C.id[j], id = mix(pid, c);
  arena[ai] = make_int2(C.arena[j], c);
  Nx.id[pos] = id; Nx.par[pos] = pid; Nx.ch[pos] = c; Nx.arena[pos] = ai;
  Nx.score[pos] = lp; Nx.b[pos] = kNegInf; Nx.nb[pos] = lp;
  Nx.kc[pos] = (int)L.tab_next[c] - 1;
  hash_claim(L.hnx, id, pos);
}

__device__ __forceinline__ float key_score(u64 key) {             // inverse of make_key's first word
  const unsigned u = ~(unsigned)(key >> 32);
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Steps 3-5 for any beam width.  Keys live in registers: thread (wave w, lane k) owns the children by candidate k of the
// entries w, w + 4, w + 8, ... (what depends on k is loaded once, what depends on the entry is a broadcast read), and
// thread i < nbm also owns entry i's own key.  The threshold key comes from a radix select with digits of up to 8 bits,
// the first digit starting at the most significant bit in which the finite keys differ (children that cannot be
// extended score -FLT_MAX: they only matter while the beam is not full, and would otherwise cost whole passes on the
// exponent bits), early exit as soon as the selected bin is wanted whole.  The kept keys are listed in LDS and installed
// by one thread each.  Ends with a barrier; returns the size of the next beam.
constexpr unsigned kNegHi = 0xFF7FFFFFu;   // first key word of a score of -FLT_MAX: ~(~bits(-FLT_MAX))
struct RadixProf { long long keys, select, install; };
template <int JPT, int NW>                 // entries per wave, waves: nbm <= NW * JPT
__device__ __forceinline__ int select_radix(Shared& sh, const Beam& C, Beam& Nx, const Cands& K, const u64* exist, int nbm,
                                            int beam, int t, int2* arena, const Lookup& L, bool profiling, RadixProf& rp) {
  const int tid = threadIdx.x, k = tid & 63, w = tid >> 6;
  long long c0 = 0;
  if (profiling) c0 = clock64();
  const int nc = K.n, kb = K.blank;
  const bool cand = k < nc && k != kb;
  const int c = cand ? K.c[k] : 0;
  const float clp = cand ? K.lp[k] : 0.f;
  u64 ekey = ~0ull, ck[JPT];
  int counts = 0;                           // valid keys + (finite keys << 16)
  u64 kmin = ~0ull;
  if (tid < nbm) {
    ekey = make_key(sh.cscore[tid], C.ch[tid], tid);
    counts += 1 + (((unsigned)(ekey >> 32) != kNegHi) << 16);
    kmin = ekey;
  }
  // what depends on the entry: lane l fetches entry w + NW l's words once, each use is a v_readlane (scalar operand) --
  // no LDS latency inside the loop over the wave's entries
  static_assert(JPT <= 64, "one lane per entry of the wave");
  const int jl = min(w + NW * k, nbm - 1);
  const u64 ex_l = exist[jl];
  const int ch_l = C.ch[jl];
  const float b_l = C.b[jl], s_l = C.score[jl];
#pragma unroll
  for (int g = 0; g < JPT; ++g) {
    const int j = w + NW * g;
    u64 key = ~0ull;
    if (j < nbm) {                          // wave-uniform
      const u64 ex = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(ex_l >> 32), g) << 32) | (unsigned)__builtin_amdgcn_readlane((int)ex_l, g);
      const int chj = __builtin_amdgcn_readlane(ch_l, g);
      const float bj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), g));
      const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s_l), g));
      if (cand && !((ex >> k) & 1ull)) {
        const float lp = c == chj ? (bj > kNegInf ? clp + bj : kNegInf) : clp + sj;
        key = make_key(lp, c, nbm + j * nc + k);
        counts += 1 + (((unsigned)(key >> 32) != kNegHi) << 16);
        kmin = key < kmin ? key : kmin;
      }
    }
    ck[g] = key;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u64 o = __shfl_xor(kmin, off);
    kmin = o < kmin ? o : kmin;
  }
  if (tid == 0) { sh.diff = 0; }
  if (k == 0) atomicMin(&sh.kmin[t & 1], kmin);
  int Mc;
  block_scan<NW>(counts, sh.wave_tot[3], &Mc);  // its barrier also publishes diff = 0 and the best key
  const int M = Mc & 0xffff, Mfin = Mc >> 16;
  if (profiling) { const long long cc = clock64(); rp.keys += cc - c0; c0 = cc; }
  u64 thr = ~0ull - 1;                      // M <= beam: every valid key (invalid ones are ~0)
  int shift_keep = 0;
  if (M > beam) {
    const u64 key0 = sh.kmin[t & 1];
    const bool fin_only = Mfin > beam;
    u64 diff = 0;
    if (ekey != ~0ull && !(fin_only && (unsigned)(ekey >> 32) == kNegHi)) diff |= ekey ^ key0;
#pragma unroll
    for (int i = 0; i < JPT; ++i)
      if (ck[i] != ~0ull && !(fin_only && (unsigned)(ck[i] >> 32) == kNegHi)) diff |= ck[i] ^ key0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) diff |= __shfl_xor(diff, off);
    if (k == 0) atomicOr(&sh.diff, diff);
    __syncthreads();
    int low = 64 - __builtin_clzll(sh.diff | 1ull);                     // the keys in play agree in bits [low, 64)
    u64 pre = low == 64 ? 0 : (key0 >> low) << low;
    int need = beam;
    for (int pass = 0; low > 0; ++pass) {
      const int width = min(8, low), shift = low - width, hb = pass & 1;
      const u64 hi_mask = low == 64 ? 0 : ~0ull << low;
      const unsigned dmask = (1u << width) - 1;
      if (ekey != ~0ull && ((ekey ^ pre) & hi_mask) == 0) atomicAdd(&sh.hist[hb][(int)((unsigned)(ekey >> shift) & dmask)], 1);
#pragma unroll
      for (int i = 0; i < JPT; ++i)
        if (ck[i] != ~0ull && ((ck[i] ^ pre) & hi_mask) == 0) atomicAdd(&sh.hist[hb][(int)((unsigned)(ck[i] >> shift) & dmask)], 1);
      __syncthreads();
      const int h = tid < 256 ? sh.hist[hb][tid] : 0;     // 256 bins; the threads beyond scan zeros
      if (tid < 256) sh.hist[hb][tid] = 0;  // clean for the pass after the next (and for the next frame)
      int tot;
      const int ex = block_scan<NW>(h, sh.wave_tot[hb], &tot);
      if (ex < need && need <= ex + h) {    // the bin that holds the need-th smallest key in play (h = 0 never qualifies)
        sh.digit = tid;
        sh.need = need - ex;
        sh.done = (need - ex == h);         // the whole bin is wanted: no need to resolve lower bits
      }
      __syncthreads();
      pre |= (u64)sh.digit << shift;
      need = sh.need;
      low = shift;
      if (sh.done) break;
    }
    shift_keep = low;
    thr = pre;
  }
  if (profiling) { const long long cc = clock64(); rp.select += cc - c0; c0 = cc; }
  // the kept keys into a list (thread order), then one thread per kept key installs it in the next beam
  const u64 thr_s = thr >> shift_keep;
  int keep_cnt = (ekey != ~0ull && (ekey >> shift_keep) <= thr_s);
#pragma unroll
  for (int i = 0; i < JPT; ++i) keep_cnt += (ck[i] != ~0ull && (ck[i] >> shift_keep) <= thr_s);
  int newn;
  int pos = block_scan<NW>(keep_cnt, sh.wave_tot[2], &newn);
  if (keep_cnt) {
    if (ekey != ~0ull && (ekey >> shift_keep) <= thr_s) { sh.keys[pos] = ekey; sh.kept_j[pos] = -1 - tid; ++pos; }
#pragma unroll
    for (int i = 0; i < JPT; ++i)
      if (ck[i] != ~0ull && (ck[i] >> shift_keep) <= thr_s) { sh.keys[pos] = ck[i]; sh.kept_j[pos] = w + NW * i; ++pos; }
  }
  __syncthreads();
  {
    // beam positions = ranks (the tie-break word of the next frame's keys is the position: beam.hip Search::better defines
    // the order among prefixes of equal score and last character through it, and both searches have to agree on it).
    // Keys are distinct (their low word is the slot): rank = number of kept keys below this one.  Two threads per kept key
    // (newn <= BMAX = NT / 2), each counting over one half of the list, sixteen keys per round requested before any is
    // compared (one LDS latency per round: the first version walked the list two keys at a time and cost 12 000 of the
    // 34 000 clocks of a beam-100 frame).
    const int kidx = tid >> 1, half = tid & 1;
    const bool mine = kidx < newn;
    const u64 key = mine ? sh.keys[kidx] : 0;
    const int j = mine ? sh.kept_j[kidx] : 0;
    const int hlen = (((newn + 1) >> 1) + 1) & ~1;          // keys per half, even (16-byte reads)
    const int qb = half * hlen, qe = min(newn, qb + hlen);
    int rank = 0;
    if (mine) {
      for (int q0 = qb; q0 < qe; q0 += 16) {
        ulonglong2 kk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kk[q] = *reinterpret_cast<const ulonglong2*>(sh.keys + min(q0 + 2 * q, NT - 2));
#pragma unroll
        for (int q = 0; q < 8; ++q) rank += (q0 + 2 * q < qe && kk[q].x < key) + (q0 + 2 * q + 1 < qe && kk[q].y < key);
      }
    }
    rank += __shfl_xor(rank, 1);
    if (mine && half == 0) {
      if (j < 0) keep_entry(sh, C, Nx, L, -1 - j, rank);
      else keep_child(C, Nx, L, arena, 1 + t * beam + rank, j, (int)((key >> 16) & 0xffff) - 1, key_score(key), rank);
    }
  }
  __syncthreads();
  if (profiling) rp.install += clock64() - c0;
  return newn;
}

// SMALL: beam <= SMALL_BEAM and beam * (min(N, beam + 2) + 1) <= NT -- one key per thread over the first ncap = min(nc, beam + 2) candidates,
// ranks by counting, the next beam written in rank order.  A child by the candidate at position k has k - 2 or more
// siblings that score at least as much (those by the candidates before it, minus the blank and the entry's own
// character; a sibling that is a beam entry itself scores at least its extension term), so children beyond position
// beam + 1 are almost never kept.  "Almost" is made exact: the frame is accepted only when the beam is full and its worst
// kept score is strictly above the best score any skipped child could have (lp[ncap] + the best entry score); otherwise
// (ties at the beam boundary, under-full beams) the frame is redone by select_radix over all candidates.
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? NT : NTW) void beam_search_kernel(BeamDeviceArgs a) {
  constexpr int NTT = SMALL ? NT : NTW, NWV = NTT / 64;      // threads / waves of this instantiation
  __shared__ Shared sh;
  extern __shared__ __align__(16) unsigned char class_tab[];     // [frame parity][tab_stride(V)]
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T, N = a.N, V = a.V, beam = a.beam;
  const int vstride = tab_stride(V);
  for (int i = tid; i < 2 * vstride / 16; i += NTT) reinterpret_cast<int4*>(class_tab)[i] = make_int4(0, 0, 0, 0);
  for (int i = tid; i < 2 * HASH / 4; i += NTT) reinterpret_cast<int4*>(sh.hpos)[i] = make_int4(0, 0, 0, 0);
  if (tid < 2) sh.cands[tid].n = 0;
  load_math_tabs(sh.math, tid, NTT);
  const int frames = a.in_len ? max(0, min(a.in_len[b], T)) : T;
  int2* arena = a.arena + (size_t)b * ((size_t)T * beam + 1);
  int cur = 0, nbm = 1;                     // current beam buffer, number of entries
  if (tid == 0) {
    Beam& B0 = sh.beams[0];
    B0.id[0] = 1; B0.par[0] = 0; B0.ch[0] = -1; B0.arena[0] = 0; B0.kc[0] = -1;
    B0.score[0] = 0.f; B0.b[0] = 0.f; B0.nb[0] = kNegInf;
    arena[0] = make_int2(-1, -1);
  }
  long long p_entries = 0, p_keys = 0, p_keep = 0, p_radix = 0, p_redone = 0;
  RadixProf rprof = {0, 0, 0};
  const bool profiling = a.prof != nullptr && b == 0 && tid == 0;
  const bool profiling3 = a.prof != nullptr && b == 0 && tid == NTT - 64;   // the last wave's share of the entry phase
  long long p_own = 0, p_scan = 0, p_prep = 0, p_cum = 0;

  // wave 3 holds the top-n list of the frame it prepares next in registers (lane k = position k); the list of the frame
  // after that is requested as soon as this one is consumed, so no global load sits on the per-frame path
  const int pl = tid - (NTT - 64);           // the last wave ("wave 3" in the comments: it is with four waves)
  float p_nx = 0.f;
  int c_nx = 0;
  if (pl >= 0 && pl < N && frames > 0) {
    p_nx = a.top_p[(size_t)b * T * N + pl];
    c_nx = a.top_idx[(size_t)b * T * N + pl];
  }
  auto prepare = [&](int tt) {              // wave 3 only: candidates of frame tt into cands[tt & 1]
    Cands& K = sh.cands[tt & 1];
    unsigned char* tab = class_tab + (tt & 1) * vstride;
    long long q0 = 0;
    if (profiling3) q0 = clock64();
    if (pl < K.n) tab[K.c[pl]] = 0;         // the list of frame tt - 2
    // number of candidates = 1 + the first position whose running sum (in doubles, position order:
    // ctc_beam_search_decoder.cpp:44-55) reaches cutoff_prob, capped by cutoff_top_n.  The sums come from a wave scan;
    // its association differs from the reference's left-to-right loop by a few ulps, so when any sum lies within 1e-9
    // of the threshold the loop below decides instead.
    int n;
    {
      const double ps = wave_prefix_sum((double)p_nx);             // lanes >= N hold 0
      const double gap = ps - (double)a.cutoff_prob;
      const u64 reach = __ballot(pl < N && gap >= 0.0);
      const u64 close = __ballot(pl < N && fabs(gap) <= 1e-9);
      n = reach ? (int)__builtin_ctzll(reach) + 1 : N;
      if (close) {
        double cum = 0.0;
        n = N;
        for (int i = 0; i < N; ++i) {
          cum += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_nx), i));
          if (cum >= a.cutoff_prob) { n = i + 1; break; }
        }
      }
      n = max(1, min(n, a.cutoff_top_n));
    }
    const bool mine = pl < n;
    if (profiling3) p_cum += clock64() - q0;
    if (mine) {
      tab[c_nx] = (unsigned char)(pl + 1);
      K.c[pl] = c_nx;
      K.lp[pl] = (float)refmath::ref_log((double)p_nx + (double)FLT_MIN, sh.math.log);
    }
    const u64 bm = __ballot(mine && c_nx == V - 1);
    if (pl == 0) {
      K.n = n;
      K.blank = bm ? (int)__builtin_ctzll(bm) : -1;
    }
    if (pl < N && tt + 1 < frames) {
      const size_t fn = ((size_t)b * T + tt + 1) * N + pl;
      p_nx = a.top_p[fn];
      c_nx = a.top_idx[fn];
    }
    if (profiling3) p_prep += clock64() - q0;
  };
  __syncthreads();                          // the cleared tables, K.n = 0
  if (tid == 0) sh.hpos[0][1 & (HASH - 1)] = 1;      // the root's id is 1
  if (pl >= 0 && frames > 0) prepare(0);
  if (tid < 256) { sh.hist[0][tid] = 0; sh.hist[1][tid] = 0; }
  if (tid < BMAX) sh.exist_mask[0][tid] = 0;
  if (tid < 2) sh.kmin[tid] = ~0ull;
  __syncthreads();

  for (int t = 0; t < frames; ++t) {
    const Beam& C = sh.beams[cur];
    Beam& Nx = sh.beams[cur ^ 1];
    const Cands& K = sh.cands[t & 1];
    u64* exist = sh.exist_mask[t & 1];
    const int nc = K.n, kb = K.blank;
    long long t0 = 0;
    if (profiling) t0 = clock64();
    // ---- 1. (wave 3) the next frame's candidates
    if (pl >= 0 && t + 1 < frames) prepare(t + 1);
    if (tid < BMAX) sh.exist_mask[(t + 1) & 1][tid] = 0;
    if (tid < HASH / 4) reinterpret_cast<int4*>(sh.hpos[cur ^ 1])[tid] = make_int4(0, 0, 0, 0);   // HASH = 1024 cells = 256 int4
    const Lookup L = {class_tab + ((t + 1) & 1) * vstride, sh.hpos[cur ^ 1]};
    if (!SMALL && tid == 0) sh.kmin[(t + 1) & 1] = ~0ull;   // select_radix's best key, per frame parity
    if (SMALL && (tid >> 6) == 2) {         // wave 2: the best entry score, for the acceptance test of the small path
      float best = kNegInf;
      for (int i = tid & 63; i < nbm; i += 64) best = fmaxf(best, C.score[i]);
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) best = fmaxf(best, __shfl_xor(best, off));
      if ((tid & 63) == 0) sh.max_score = best;
    }
    // ---- 2. existing entries: blank update, repetition, extension by the parent if that is in the beam
    if (tid < nbm) {
      const int i = tid;
      const int kc = C.kc[i];
      const u64 pid = C.par[i];
      const float nb_i = C.nb[i];
      const float bc = kb >= 0 ? K.lp[kb] + C.score[i] : kNegInf;  // log_sum_exp(-inf, x) = x
      float nbc = kNegInf;
      if (kc >= 0) {
        nbc = K.lp[kc] + nb_i;
        const int j = hash_find(sh.hpos[cur], C, nbm, pid);
        if (profiling) p_scan += clock64() - t0;
        if (j >= 0) {
          nbc = lse(sh.math, nbc, child_lp(C, K, j, kc));
          atomicOr(&exist[j], 1ull << kc);
        }
      }
      sh.cb[i] = bc;
      sh.cnb[i] = nbc;
      sh.cscore[i] = lse(sh.math, bc, nbc);
    }
    if (profiling) p_own += clock64() - t0;
    __syncthreads();
    long long t1 = 0;
    if (profiling) { t1 = clock64(); p_entries += t1 - t0; }

    int newn = 0;
    bool redo = !SMALL;
    if (SMALL) {
      // ---- 3s. slots [0, nbm) = existing entries, nbm + j * ncap + k = child of entry j by candidate k; one thread
      //          per slot, or a pair of threads when the slots fill at most half of the workgroup (ncap gives one
      //          candidate away for that: the acceptance test below does not care how ncap was chosen)
      int ncap = min(nc, beam + 2);
      if (nbm + nbm * ncap > NT / 2 && ncap > 1 && nbm + nbm * (ncap - 1) <= NT / 2) --ncap;
      const int S = nbm + nbm * ncap;
      const bool pairs = S <= NT / 2;
      const int slot = pairs ? tid >> 1 : tid, half = pairs ? tid & 1 : 0;
      u64 key = ~0ull;
      int j = 0, k = 0;
      float lp = kNegInf;
      if (slot < nbm) {
        lp = sh.cscore[slot];
        key = make_key(lp, C.ch[slot], slot);
      } else if (slot < S) {
        // (slot - nbm) / ncap without the integer-division sequence: (x + 0.5) / n is at least 0.5 / n away from an integer
        j = (int)(((float)(slot - nbm) + 0.5f) * __builtin_amdgcn_rcpf((float)ncap));
        k = (slot - nbm) - j * ncap;
        // every operand is requested before any is looked at: one LDS latency instead of three
        const u64 ex = exist[j];
        const int c = K.c[k], cj = C.ch[j];
        const float lk = K.lp[k], bj = C.b[j], sj = C.score[j];
        if (k != kb && !((ex >> k) & 1ull)) {
          lp = c == cj ? (bj > kNegInf ? lk + bj : kNegInf) : lk + sj;    // child_lp
          key = make_key(lp, c, slot);
        }
      }
      if (half == 0) sh.keys[slot] = key;
      if (pairs && half == 1 && slot + NT / 2 < NT) sh.keys[slot + NT / 2] = ~0ull;
      if (tid == 0) sh.thr_ok = 0;            // every wave has read the previous frame's flag: it is past this frame's first barrier
      __syncthreads();
      // ---- 4s. rank = number of better keys (slots make keys distinct); a pair splits the count
      int rank = 0;
      const int sbeg = pairs ? half * (NT / 4) : 0, send = pairs ? min(S, sbeg + NT / 4) : S;
      for (int s0 = sbeg; s0 < send; s0 += 16) {   // slots [S, NT) hold ~0 and count for nothing
        ulonglong2 kk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kk[q] = reinterpret_cast<const ulonglong2*>(sh.keys + s0)[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) rank += (kk[q].x < key) + (kk[q].y < key);
      }
      if (pairs) rank += __shfl_xor(rank, 1);
      const bool keep = half == 0 && key != ~0ull && rank < beam;
      long long t2 = 0;
      if (profiling) { t2 = clock64(); p_keys += t2 - t1; }
      if (keep) {
        if (slot < nbm) keep_entry(sh, C, Nx, L, slot, rank);
        else keep_child(C, Nx, L, arena, 1 + t * beam + rank, j, K.c[k], lp, rank);
        if (rank == beam - 1 && ncap < nc && lp > K.lp[ncap] + sh.max_score) sh.thr_ok = 1;
      }
      newn = __syncthreads_count(keep);
      redo = ncap < nc && !sh.thr_ok;       // uniform
      if (profiling) p_keep += clock64() - t2;
    }
    if (redo) {
      long long t3 = 0;
      if (profiling) { t3 = clock64(); if (SMALL) ++p_redone; }
      if (SMALL) {                          // a rare frame: reset the best-key cell here rather than in every frame
        if (tid == 0) sh.kmin[t & 1] = ~0ull;
        __syncthreads();
      }
      newn = select_radix<(SMALL ? SMALL_BEAM : BMAX) / NWV, NWV>(sh, C, Nx, K, exist, nbm, beam, t, arena, L, profiling, rprof);
      if (profiling) p_radix += clock64() - t3;
    }
    cur ^= 1;
    nbm = newn;
  }
  if (profiling) {
    a.prof[0] = p_entries; a.prof[1] = p_keys; a.prof[2] = p_keep; a.prof[3] = p_radix; a.prof[4] = p_redone;
    a.prof[5] = rprof.keys; a.prof[6] = rprof.select; a.prof[7] = rprof.install; a.prof[8] = frames;
    a.prof[9] = p_own; a.prof[10] = p_scan;
  }
  if (profiling3) { a.prof[11] = p_prep; a.prof[12] = p_cum; }

  // ---- finish: rank by prefix_compare (+ slot), read the paths back
  const Beam& C = sh.beams[cur];
  const int n = min(nbm, beam);
  int32_t* ids = a.ids + (size_t)b * beam * a.max_len;
  int32_t* lens = a.lens + (size_t)b * beam;
  float* scores = a.scores + (size_t)b * beam;
  if (tid == 0) a.n_hyp[b] = n;
  for (int i = tid; i < beam; i += NTT) {
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
  for (int i = n + tid; i < beam; i += NTT) {
    lens[i] = 0;
    scores[i] = kNegInf;
    for (int q = 0; q < a.max_len; ++q) ids[(size_t)i * a.max_len + q] = -1;
  }
}

// the device's evaluation of refmath.h, for the parity test against the host's libm (tests/test_gpu_parity.py)
__global__ __launch_bounds__(256) void refmath_eval_kernel(int kind, const float* __restrict__ in, void* __restrict__ out, int n) {
  __shared__ MathTabs m;
  load_math_tabs(m, threadIdx.x, 256);
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = in[i];
    if (kind == 0) static_cast<float*>(out)[i] = refmath::ref_expf(x, m.exp2f);
    else if (kind == 1) static_cast<float*>(out)[i] = refmath::ref_logf(x, m.logf);
    else if (kind == 2) static_cast<double*>(out)[i] = refmath::ref_log((double)x + (double)FLT_MIN, m.log);
    else static_cast<float*>(out)[i] = lse(m, x, in[n + i]);   // kind 3: log_sum_exp(in[i], in[n + i])
  }
}

}  // namespace

extern "C" {

int mi355asr_launch_refmath_eval(int kind, const float* in, void* out, int n, hipStream_t s) {
  if (kind < 0 || kind > 3 || n < 0) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(refmath_eval_kernel, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, s, kind, in, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// the class tables (2 V bytes of dynamic LDS) sit next to the kernel's static LDS in the CU's 160 KB
constexpr int kLdsBytes = 160 * 1024;
constexpr int kMaxClasses = std::min(65534, ((kLdsBytes - (int)sizeof(Shared) - 512) / 2) & ~15);
static_assert(kMaxClasses >= 32768, "the static LDS of the search left too little for the class tables");
bool mi355asr_beam_device_applicable(int V, int N, int beam) { return beam >= 1 && beam <= BMAX && N >= 1 && N <= NMAX && V <= kMaxClasses; }

size_t mi355asr_beam_device_carve(char* ws, int B, int T, int beam, int max_len, BeamDeviceArgs* a, int32_t** d_len, long long** prof) {
  size_t off = 0;
  auto seg = [&](size_t bytes) { char* p = ws ? ws + off : nullptr; off += (bytes + 15) & ~(size_t)15; return p; };
  char* arena = seg((size_t)B * ((size_t)T * beam + 1) * sizeof(int2));
  char* ids = seg((size_t)B * beam * max_len * sizeof(int32_t));
  char* lens = seg((size_t)B * beam * sizeof(int32_t));
  char* scores = seg((size_t)B * beam * sizeof(float));
  char* n_hyp = seg((size_t)B * sizeof(int32_t));
  char* len = seg((size_t)B * sizeof(int32_t));
  char* pr = seg(16 * sizeof(long long));
  if (a) { a->arena = (int2*)arena; a->ids = (int32_t*)ids; a->lens = (int32_t*)lens; a->scores = (float*)scores; a->n_hyp = (int32_t*)n_hyp; }
  if (d_len) *d_len = (int32_t*)len;
  if (prof) *prof = (long long*)pr;
  return off;
}
size_t mi355asr_beam_device_ws_bytes(int B, int T, int beam, int max_len) {
  return mi355asr_beam_device_carve(nullptr, B, T, beam, max_len, nullptr, nullptr, nullptr);
}

int mi355asr_launch_beam_device(const BeamDeviceArgs* a, hipStream_t s) {
  const bool small = a->beam <= SMALL_BEAM && a->beam * (std::min(a->N, a->beam + 2) + 1) <= NT;
  const int dyn = 2 * tab_stride(a->V);     // the class tables
  if (a->V > kMaxClasses) return -2;       // mi355asr_beam_device_applicable
  // per DEVICE: the attribute belongs to the kernel's code object on the device that is current (a process that drives
  // several GPUs would otherwise launch large-V searches without it everywhere but on the first)
  static bool allowed_on[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
  if (!allowed_on[dev]) {
    if (hipFuncSetAttribute((const void*)beam_search_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * tab_stride(kMaxClasses)) != hipSuccess ||
        hipFuncSetAttribute((const void*)beam_search_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * tab_stride(kMaxClasses)) != hipSuccess)
      return -2;
    allowed_on[dev] = true;
  }
  if (small) hipLaunchKernelGGL(beam_search_kernel<true>, dim3(a->B), dim3(NT), dyn, s, *a);
  else hipLaunchKernelGGL(beam_search_kernel<false>, dim3(a->B), dim3(NTW), dyn, s, *a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
