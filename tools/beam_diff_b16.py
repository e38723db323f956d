"""Where do the device prefix search and the host search part ways on the benched config-5 shape (16 x 30 s)?  Rebuilds the
logits of tests/test_gpu_baseline_shapes.py::test_config5_batch16_..., finds per utterance the first frame count at which
the two beams differ (the search of the first t frames = in_len t) and prints that frame's candidates and both beams.
Run on a GPU box: python tools/beam_diff_b16.py [beam]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import co, waves
from test_gpu_baseline_shapes import chunk_config_dict
from test_gpu_parity import _pick_bias_for_ragged_counts
from tensorflowasr_amd.models import ChunkConformer, ctc_prefix_beam_decode

beam = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = dict(co.CHUNK_S)
w = co.chunk_weights(cfg, seed=5)
B = 16
x = waves(B, 480000, 200)
w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, x[[1, 13]])
m = ChunkConformer(chunk_config_dict(cfg), cfg["picker_num_classes"], cfg["decoder_num_classes"])
m.load_weights(w, by_name=False)
got = m.predict(x, stages=True)
counts = got["counts"]
probs = torch.softmax(got["text_logits"], -1)
ph = probs.cpu().numpy()


def same(a, b):
    return all(np.array_equal(p, q) for p, q in zip(a, b))


def run(lens):
    lens = np.asarray(lens, np.int32)
    return ctc_prefix_beam_decode(probs, lens, beam, 0.99, 40), ctc_prefix_beam_decode(ph, lens, beam, 0.99, 40)


d, h = run(counts)
for u in range(B):
    if all(np.array_equal(p[u], q[u]) for p, q in zip(d, h)):
        continue
    lo, hi = 0, int(counts[u])          # equal at lo frames, different at hi
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lens = np.zeros(B, np.int32)
        lens[u] = mid
        dd, hh = run(lens)
        if all(np.array_equal(p[u], q[u]) for p, q in zip(dd, hh)):
            lo = mid
        else:
            hi = mid
    print("utterance %d (%d frames): beams equal after %d frames, differ after %d" % (u, counts[u], lo, hi))
    t = hi - 1
    row = ph[u, t]
    order = np.argsort(-row, kind="stable")[:42]
    cum = np.cumsum(row[order].astype(np.float64))
    print("  frame %d candidates (class, p, cum):" % t)
    for k in range(min(42, len(order))):
        print("    %2d  %5d  %.9g  %r  %.17g" % (k, order[k], row[order[k]], row[order[k]].tobytes().hex(), cum[k]))
        if cum[k] >= 0.99 and k > 2 and cum[k - 2] >= 0.99:
            break
    for tag, tt in (("before", lo), ("after", hi)):
        lens = np.zeros(B, np.int32)
        lens[u] = tt
        dd, hh = run(lens)
        print("  beams %s frame %d  (rank: score dev / host, len, last tokens)" % (tag, t))
        for i in range(int(max(dd[3][u], hh[3][u]))):
            ld, lh = dd[1][u, i], hh[1][u, i]
            print("    %2d  %r %r  len %d %d  dev %s host %s %s" % (i, float(dd[2][u, i]), float(hh[2][u, i]), ld, lh,
                  dd[0][u, i, max(0, ld - 4):ld].tolist(), hh[0][u, i, max(0, lh - 4):lh].tolist(),
                  "" if np.array_equal(dd[0][u, i], hh[0][u, i]) and dd[2][u, i] == hh[2][u, i] else "<--"))
