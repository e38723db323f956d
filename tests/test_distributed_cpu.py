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
            works.append((all_gather_ids(torch.from_numpy(ids + shift), torch.from_numpy(lens), async_op=True), shift))
        for w_, shift in works:
            a_ids, a_lens = w_.result()
            assert np.array_equal(a_ids.numpy(), full_ids + shift) and np.array_equal(a_lens.numpy(), full_lens)
        # ... with ids and lengths as two views of one buffer (parallel.ids_lens_buffer, bench.py's rotating output sets): ONE
        # collective per batch, the same result
        from tensorflowasr_amd.parallel import _packed_pair, ids_lens_buffer
        works = []
        for shift in range(3):
            b_ids, b_lens = ids_lens_buffer(ids.shape[0], ids.shape[1], "cpu")
            b_ids.copy_(torch.from_numpy(ids + shift)); b_lens.copy_(torch.from_numpy(lens))
            assert _packed_pair(b_ids, b_lens) is not None and _packed_pair(torch.from_numpy(ids), torch.from_numpy(lens)) is None
            w_ = all_gather_ids(b_ids, b_lens, async_op=True)
            assert len(w_._works) == 1
            works.append((w_, shift))
        for w_, shift in works:
            a_ids, a_lens = w_.result()
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


# ---------------------------------------------------------------------------------------------------------------
# world 8: BASELINE config 4's partition (512 utterances -> 8 x 64) and config 5's (128 -> 8 x 16, ragged lengths), and
# batches that do NOT divide by the world size (the reference's batch wrapper takes any batch)
# ---------------------------------------------------------------------------------------------------------------
def _world8_worker(rank, world, port, tmpdir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    from tensorflowasr_amd.parallel import all_gather_hypotheses, all_gather_ids, shard_range, shard_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blank = 9
        # config 4: 512 utterances, 64 per rank (tiny T); then 509 and 5 utterances (uneven, and ranks with EMPTY shards)
        for B in (512, 509, 5):
            T = 12
            frames = np.random.default_rng(B).integers(0, 10, (B, T)).astype(np.int32)
            in_len = np.random.default_rng(B + 1).integers(0, T + 1, B).astype(np.int32)
            lo, hi = shard_range(B, rank, world)
            assert hi - lo == shard_sizes(B, world)[rank] and (B != 512 or hi - lo == 64)
            ids, lens = co.ctc_collapse(frames[lo:hi], in_len[lo:hi], blank) if hi > lo else (np.zeros((0, T), np.int32), np.zeros((0,), np.int32))
            full_ids, full_lens = co.ctc_collapse(frames, in_len, blank)
            a_ids, a_lens = all_gather_ids(torch.from_numpy(ids), torch.from_numpy(lens), n_total=B)
            assert np.array_equal(a_ids.numpy(), full_ids) and np.array_equal(a_lens.numpy(), full_lens)
            work = all_gather_ids(torch.from_numpy(ids), torch.from_numpy(lens), async_op=True, n_total=B)   # same handle, padded or not
            r_ids, r_lens = work.result()
            assert np.array_equal(r_ids.numpy(), full_ids) and np.array_equal(r_lens.numpy(), full_lens)
            if B == 512:                    # equal shards: the un-padded form as well
                a_ids, a_lens = all_gather_ids(torch.from_numpy(ids), torch.from_numpy(lens))
                assert np.array_equal(a_ids.numpy(), full_ids)
        # a rank that does not hold its shard_range slice is told so (instead of a collective that mis-sizes)
        try:
            all_gather_ids(torch.zeros((3, 4), dtype=torch.int32), torch.zeros(3, dtype=torch.int32), n_total=5 * world + 1)
            raise AssertionError("expected ValueError")
        except ValueError as e:
            assert "shard_range" in str(e)
        # config 5: 128 utterances -> 16 per rank, ragged frame counts and hypothesis lengths; then 19 (uneven)
        for B in (128, 19):
            T, V, beam = 14, 11, 4
            rng = np.random.default_rng(7 + B)
            lg = rng.standard_normal((B, T, V)).astype(np.float32) * 2.0
            probs = np.exp(lg) / np.exp(lg).sum(-1, keepdims=True)
            in_len = rng.integers(0, T + 1, B).astype(np.int32)
            lo, hi = shard_range(B, rank, world)
            assert B != 128 or hi - lo == 16
            ml = max(int(in_len[lo:hi].max()) if hi > lo else 0, 1)
            mine = ctc_prefix_beam_decode(probs[lo:hi], in_len[lo:hi], beam_width=beam, cutoff_prob=0.999, cutoff_top_n=8, num_threads=1,
                                          max_len=ml) if hi > lo else \
                (np.zeros((0, beam, 1), np.int32), np.zeros((0, beam), np.int32), np.zeros((0, beam), np.float32), np.zeros((0,), np.int32))
            ids, lens, scores, n_hyp = all_gather_hypotheses(*mine, n_total=B)
            if rank == 0:
                full = ctc_prefix_beam_decode(probs, in_len, beam_width=beam, cutoff_prob=0.999, cutoff_top_n=8, num_threads=1, max_len=T)
                assert ids.shape[0] == B and np.array_equal(n_hyp, full[3]) and np.array_equal(lens, full[1])
                assert np.array_equal(scores.view(np.int32), full[2].view(np.int32))
                L = ids.shape[2]
                assert np.array_equal(ids, full[0][:, :, :L]) and (full[0][:, :, L:] == -1).all()
        open(os.path.join(tmpdir, "w8_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_dp_world8_config4_and_config5_partitions_and_uneven_batches(tmp_path):
    world = 8
    mp.spawn(_world8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("w8_ok%d" % r)).exists() for r in range(world))



def test_bench_starts_its_own_ranks_when_run_the_way_the_driver_runs_it():
    """Round-4 review: `python3 bench.py --gpus N` (no launcher around it) stopped in argparse.  Now it re-runs itself under
    torch.distributed.run, one rank per GPU.  --dry-run-gloo takes the GPU out of that path (gloo, a stand-in recogniser) and
    keeps everything else of main()'s multi-rank control flow: rendezvous on 127.0.0.1, rank-count check through the backend,
    rotating output sets, the asynchronous per-step id exchange, barrier + max-over-ranks timing, repeated timed regions and
    ONE JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for n in (2, 4):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "1", "--batch", "3",
                            "--dry-run-gloo"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout                       # rank 0 only
        line = json.loads(lines[0])
        assert line["n_gpus"] == n and line["config"]["ranks"] == n and line["config"]["parallelism"] == "dp%d" % n
        assert line["gathered_ids_in_rank_order"] is True and line["dry_run"] is True
        assert line["steps"] == 4 and line["config"]["global_batch"] == 3 * n and line["scaling"] == "weak"
        for k in ("metric", "value", "unit", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "timed_region_s",
                  "region_ms_per_step"):
            assert k in line, k
    # a WORLD_SIZE that contradicts --gpus is refused, not silently benchmarked
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-gloo"], capture_output=True, text=True,
                       timeout=120, cwd=root, env=dict(env, WORLD_SIZE="3", RANK="0"))
    assert r.returncode != 0 and "does not match" in r.stderr
