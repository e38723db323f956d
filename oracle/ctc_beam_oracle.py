"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's scorer-less CTC prefix beam search
(externals/ctc_decoders.zip: ctc_beam_search_decoder.cpp:18-187, decoder_utils.cpp:7-38,137-147,
decoder_utils.h:41-49, path_trie.cpp:11-147) in plain Python, float32 trie scores as in the reference.

Pinned against the reference itself: oracle/_ref/libref_ctc_beam.so (built by `make -C oracle ref_beam` from the
unpacked zip) produced tests/golden/beam_kat.json; tests/test_oracle.py replays them.

Reference behaviours that are kept on purpose:
  * blank = last class; vocabulary = classes 0..V-2.
  * get_pruned_log_probs: when cutoff_prob == 1.0 NOTHING is pruned, whatever cutoff_top_n says (the top-n bound is
    only applied inside the `cutoff_prob < 1.0` loop, decoder_utils.cpp:18-31); classes are then visited in
    descending-probability order.  log prob = float32(log(p + FLT_MIN)) computed in double.
  * trie scores are float32; log_sum_exp treats <= -FLT_MAX as -inf.
  * beams are kept with nth_element (order among survivors unspecified) and the final result is sorted by
    (score desc, last character asc); exact ties beyond that are unspecified in the reference too.
"""
import math

import numpy as np

F32 = np.float32
NEG = F32(-np.finfo(np.float32).max)
FLT_MIN = float(np.finfo(np.float32).tiny)


def log_sum_exp(x, y):
    """decoder_utils.h:41-49 on float32."""
    if x <= NEG:
        return y
    if y <= NEG:
        return x
    m = max(x, y)
    return F32(F32(math.log(F32(math.exp(F32(x - m))) + F32(math.exp(F32(y - m))))) + m)


def _lse32(x, y):
    # the reference instantiates log_sum_exp<float>: std::exp/std::log on float arguments are the float
    # overloads, i.e. correctly rounded-ish float32 results.  numpy float32 exp/log match to the last bit on
    # glibc for the vast majority of inputs; KAT comparison uses a 1e-5 score tolerance for the rest.
    if x <= NEG:
        return y
    if y <= NEG:
        return x
    m = x if x > y else y
    return F32(np.log(np.exp(F32(x - m), dtype=np.float32) + np.exp(F32(y - m), dtype=np.float32), dtype=np.float32) + m)


class _Node:
    __slots__ = ("ch", "parent", "children", "b_prev", "nb_prev", "b_cur", "nb_cur", "score", "exists")

    def __init__(self, ch=-1, parent=None):
        self.ch, self.parent, self.children = ch, parent, []
        self.b_prev = self.nb_prev = self.b_cur = self.nb_cur = self.score = NEG
        self.exists = True

    def child(self, c):
        for k, n in self.children:
            if k == c:
                if not n.exists:
                    n.exists = True
                    n.b_prev = n.nb_prev = n.b_cur = n.nb_cur = NEG
                return n
        n = _Node(c, self)
        self.children.append((c, n))
        return n

    def collect(self, out):
        if self.exists:
            self.b_prev, self.nb_prev = self.b_cur, self.nb_cur
            self.b_cur = self.nb_cur = NEG
            self.score = _lse32(self.b_prev, self.nb_prev)
            out.append(self)
        for _, n in self.children:
            n.collect(out)

    def remove(self):
        self.exists = False
        if not self.children:
            p = self.parent
            p.children = [(k, n) for k, n in p.children if n is not self]
            if not p.children and not p.exists:
                p.remove()

    def path(self):
        out, n = [], self
        while n.ch != -1:
            out.append(n.ch)
            n = n.parent
        return out[::-1]


def pruned_log_probs(prob, cutoff_prob, cutoff_top_n):
    """decoder_utils.cpp:7-38 -> list of (class, float32 log prob)."""
    V = len(prob)
    idx = list(range(V))
    n = V
    if cutoff_prob < 1.0 or cutoff_top_n < n:
        idx.sort(key=lambda i: -prob[i])          # reference: std::sort (tie order unspecified)
        if cutoff_prob < 1.0:
            cum, n = 0.0, 0
            for i in idx:
                cum += float(prob[i])
                n += 1
                if cum >= cutoff_prob or n >= cutoff_top_n:
                    break
        idx = idx[:n]
    return [(i, F32(math.log(float(prob[i]) + FLT_MIN))) for i in idx]


def _key(n):
    return (-float(n.score), n.ch)              # prefix_compare: score desc, then character asc


def ctc_beam_search(probs, beam_size, cutoff_prob=1.0, cutoff_top_n=40):
    """probs [T, V] (V includes the blank = V-1).  -> list of (score float32, [ids]) best first."""
    probs = np.asarray(probs, dtype=np.float64)
    T, V = probs.shape
    blank = V - 1
    root = _Node()
    root.score = root.b_prev = F32(0.0)
    prefixes = [root]
    for t in range(T):
        for c, lp in pruned_log_probs(probs[t], cutoff_prob, cutoff_top_n):
            for p in prefixes[:beam_size]:
                if c == blank:
                    p.b_cur = _lse32(p.b_cur, F32(lp + p.score))
                    continue
                if c == p.ch:
                    p.nb_cur = _lse32(p.nb_cur, F32(lp + p.nb_prev))
                q = p.child(c)
                log_p = NEG
                if c == p.ch and p.b_prev > NEG:
                    log_p = F32(lp + p.b_prev)
                elif c != p.ch:
                    log_p = F32(lp + p.score)
                q.nb_cur = _lse32(q.nb_cur, log_p)
        prefixes = []
        root.collect(prefixes)
        if len(prefixes) >= beam_size:
            prefixes.sort(key=_key)               # nth_element keeps the same top set (order irrelevant, see above)
            for n in prefixes[beam_size:]:
                n.remove()
            prefixes = prefixes[:beam_size]
    prefixes.sort(key=_key)
    return [(n.score, n.path()) for n in prefixes[:beam_size]]
