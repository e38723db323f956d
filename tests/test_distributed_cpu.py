"""world_size-2 gloo test of the data-parallel host path (shard -> local decode -> all_gather), on CPU.
The per-rank "decode" here is the host reference of the collapse step so that the exchange logic (ordering,
shapes, broadcast of weights) is what is under test; the GPU path swaps in mi355asr_recognize."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import co


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tensorflowasr_amd.parallel import all_gather_ids, broadcast_weights, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: rank 0 holds the real ones, the others zeros -> broadcast makes them identical
        rng = np.random.default_rng(0)
        ref_w = {"a/kernel": rng.standard_normal((5, 7)).astype(np.float32), "b/bias": rng.standard_normal(3).astype(np.float32)}
        mine = ref_w if rank == 0 else {k: np.zeros_like(v) for k, v in ref_w.items()}
        got = broadcast_weights(mine, src=0)
        assert all(np.array_equal(got[k], ref_w[k]) for k in ref_w)
        # global batch of 6 "utterances" of per-frame argmax ids; contiguous shard per rank
        B, T, blank = 6, 40, 9
        frames = np.random.default_rng(1).integers(0, 10, (B, T)).astype(np.int32)
        in_len = np.array([40, 33, 0, 17, 40, 5], np.int32)
        lo, hi = shard_range(B, rank, world)
        ids, lens = co.ctc_collapse(frames[lo:hi], in_len[lo:hi], blank)
        all_ids, all_lens = all_gather_ids(torch.from_numpy(ids), torch.from_numpy(lens))
        full_ids, full_lens = co.ctc_collapse(frames, in_len, blank)
        assert np.array_equal(all_ids.numpy(), full_ids) and np.array_equal(all_lens.numpy(), full_lens)
        # the pipelined form bench.py uses: several exchanges in flight, waited for at the end
        works = []
        for shift in range(3):
            w_, a_ids, a_lens = all_gather_ids(torch.from_numpy(ids + shift), torch.from_numpy(lens), async_op=True)
            works.append((w_, a_ids, a_lens, shift))
        for w_, a_ids, a_lens, shift in works:
            w_.wait()
            assert np.array_equal(a_ids.numpy(), full_ids + shift) and np.array_equal(a_lens.numpy(), full_lens)
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_dp_shard_broadcast_allgather_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def _beam_worker(rank, world, port, tmpdir):
    """BASELINE config 5's exchange: every rank runs the prefix beam search on its utterance shard (here the host search of
    libmi355asr.so, which needs no GPU; on the GPU box the device search) and all ranks end up with the beams of the whole
    batch, ragged hypothesis lengths padded to the global maximum."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    from tensorflowasr_amd.parallel import all_gather_hypotheses, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, T, V, beam = 6, 24, 12, 5
        rng = np.random.default_rng(3)
        logits = rng.standard_normal((B, T, V)).astype(np.float32) * 2.0
        probs = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
        in_len = np.array([24, 9, 1, 17, 24, 5], np.int32)
        lo, hi = shard_range(B, rank, world)
        # a rank's max_len is data dependent on the real path (frames kept by feature_pick): make them differ here
        mine = ctc_prefix_beam_decode(probs[lo:hi], in_len[lo:hi], beam_width=beam, cutoff_prob=1.0, cutoff_top_n=40,
                                      max_len=int(in_len[lo:hi].max()))
        ids, lens, scores, n_hyp = all_gather_hypotheses(*mine)
        full = ctc_prefix_beam_decode(probs, in_len, beam_width=beam, cutoff_prob=1.0, cutoff_top_n=40, max_len=int(in_len.max()))
        assert ids.shape == (B, beam, int(in_len.max())) and np.array_equal(n_hyp, full[3])
        for b in range(B):
            for k in range(int(n_hyp[b])):
                assert lens[b, k] == full[1][b, k] and scores[b, k] == full[2][b, k]
                assert np.array_equal(ids[b, k, :lens[b, k]], full[0][b, k, :lens[b, k]])
                assert (ids[b, k, lens[b, k]:] == -1).all()
        open(os.path.join(tmpdir, "beam_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_dp_prefix_beam_hypotheses_allgather_world2(tmp_path):
    world = 2
    mp.spawn(_beam_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("beam_ok%d" % r)).exists() for r in range(world))
