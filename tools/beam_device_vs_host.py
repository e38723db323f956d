"""Lists every hypothesis in which the device prefix search (beam_device.hip) and the host search (beam.hip) differ, for the
input mix of tests/test_gpu_parity.py::test_prefix_beam_device_search_equals_host_search: score, length, positions that
differ, and whether the sets of hypotheses agree.  Differences are expected only inside groups of equal float score (the
reference leaves their order to std::sort / std::nth_element).  Run on a GPU box: python tools/beam_device_vs_host.py"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from tensorflowasr_amd.models import ctc_prefix_beam_decode
for beam in (10, 14, 15, 40, 100):
    rng = np.random.default_rng(100 + beam)
    B, T, V = 6, 90, 300
    z = rng.standard_normal((B, T, V)).astype(np.float32)
    z[0] *= 6.0; z[1] *= 0.2; z[2] *= 3.0; z[2, :, -1] += 8.0
    z[3] *= rng.uniform(0.1, 6.0, (T, 1)).astype(np.float32)
    z[4] *= 2.0; z[5] *= 4.0; z[5, ::3, -1] += 6.0
    z[5, 1::3] = z[5, ::3][: z[5, 1::3].shape[0]] + 0.01 * z[5, 1::3]
    p = torch.softmax(torch.from_numpy(z), -1).numpy()
    in_len = np.array([T, T, 61, T, 1, 0], np.int32)
    for cutoff_prob, top_n in ((0.99, 40), (0.9999, 25)):
        d = ctc_prefix_beam_decode(torch.from_numpy(p).cuda(), in_len, beam, cutoff_prob, top_n)
        h = ctc_prefix_beam_decode(p, in_len, beam, cutoff_prob, top_n)
        for b in range(B):
            for i in range(beam):
                if not np.array_equal(d[0][b, i], h[0][b, i]) or d[2][b, i] != h[2][b, i]:
                    L = max(d[1][b, i], h[1][b, i])
                    nd = int((d[0][b, i][:L] != h[0][b, i][:L]).sum())
                    print(f"beam {beam} cut {cutoff_prob} utt {b} hyp {i}: score d {d[2][b,i]!r} h {h[2][b,i]!r} len {d[1][b,i]} {h[1][b,i]} ndiff {nd}")
        # set comparison
        for b in range(B):
            sd = {tuple(d[0][b, i][:d[1][b, i]]) for i in range(d[3][b])}
            shh = {tuple(h[0][b, i][:h[1][b, i]]) for i in range(h[3][b])}
            if sd != shh: print(f"  beam {beam} utt {b}: sets differ, common {len(sd & shh)} of {len(shh)}")
