"""One worker process of bench.py's `cpu_baseline` "worker processes" leg (TEST INFRASTRUCTURE: the oracle is the checker and the
reported CPU baseline, never the product).  Runs the fp32 NumPy oracle -- waveform -> Mel frontend -> ConformerEncoder(S) ->
CTCDecoder -> greedy ids, the path of bench.py's headline step -- on utterances `first, first + stride, ...` of the benched
synthetic batch with the BLAS pool limited to one thread, so that N workers use N cores on N independent utterances (the
reference's own batch decode shards utterances the same way).

Protocol on stdio: prints READY when the weights are built and one warm-up utterance has run, waits for a line on stdin, runs its
utterances, prints `DONE <utterances> <seconds>`."""
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import conformer_oracle as co  # noqa: E402


def main():
    first, stride, total, L, V = (int(a) for a in sys.argv[1:6])
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    golden = os.path.join(ROOT, "tests", "golden", "ctc_decoder_weights.npz")
    w.update(dict(np.load(golden)) if os.path.exists(golden) else co.ctc_decoder_weights(cfg, V))

    def one(u):
        x = co.synth_wave(u, L)[None]
        enc = co.conformer_encoder(x, w, cfg, dtype=np.float32)
        logits = co.ctc_decoder(enc, w, cfg, dtype=np.float32)
        return co.ctc_greedy(logits, [logits.shape[1]], V - 1)

    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    one(first)
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    n = 0
    for u in range(first, total, stride):
        one(u)
        n += 1
    print("DONE %d %.4f" % (n, time.perf_counter() - t0), flush=True)
    del limit


if __name__ == "__main__":
    main()
