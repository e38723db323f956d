"""Utterance-batch data parallelism: one process per GPU, weights replicated, utterances sharded
contiguously, results exchanged with torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on
CPU for tests).  The reference has no multi-device inference path: this is new design.

Collectives on the path: one broadcast of the flattened weights at start-up; per batch one all_gather of
`int32[B_local, T + 1]` (token ids, -1 padded, plus the length in the last column; ~64 KB per rank at
B_local=64, T=250).  The ChunkConformer + prefix-beam configuration (BASELINE config 5) exchanges whole beams:
`all_gather_hypotheses` -- one tiny all_reduce(MAX) for the common hypothesis length, then one all_gather of the packed
`int32[B_local, beam, max_len + 3]` block (ids | length | score bits | n_hyp); the reference spreads the utterances of a
batch over a thread pool instead (externals/ctc_decoders ctc_beam_search_decoder.cpp:426-459)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous split of n_items over `world` ranks (first n_items % world ranks get one extra)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_weights(weights, src=0, device=None, group=None):
    """Broadcast a name -> float32 array dict from `src`.  Every rank passes a dict with the same keys/shapes
    (non-src values are overwritten).  One flat buffer => one collective."""
    names = sorted(weights)
    sizes = [int(np.prod(weights[n].shape)) for n in names]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if dist.get_rank(group) == src:
        flat.copy_(torch.from_numpy(np.concatenate([np.asarray(weights[n], np.float32).reshape(-1) for n in names])))
    dist.broadcast(flat, src=src, group=group)
    host = flat.cpu().numpy()
    out, o = {}, 0
    for n, s in zip(names, sizes):
        out[n] = host[o:o + s].reshape(weights[n].shape).copy()
        o += s
    return out


def shard_sizes(n_items, world):
    """[rows of rank 0, rows of rank 1, ...] under shard_range"""
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def _pad_rows(x, rows, fill):
    if x.shape[0] == rows:
        return x
    pad = torch.full((rows - x.shape[0],) + tuple(x.shape[1:]), fill, dtype=x.dtype, device=x.device)
    return torch.cat([x, pad], 0)


def _valid_rows(n_total, world, b_pad, device):
    """indices of the real rows in a [world * b_pad, ...] gather of shard_range shards padded to b_pad rows each"""
    idx = [r * b_pad + i for r, n in enumerate(shard_sizes(n_total, world)) for i in range(n)]
    return torch.as_tensor(idx, dtype=torch.long, device=device)


def _check_local_rows(B, n_total, group):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    want = shard_sizes(n_total, world)[rank]
    if B != want:
        raise ValueError("rank %d holds %d utterances, shard_range(%d, %d, %d) gives %d: every rank has to decode exactly its "
                         "shard_range slice of the batch" % (rank, B, n_total, rank, world, want))


class GatherWork:
    """handle of an asynchronous all_gather_ids: wait(), then result() -> (ids_all, lens_all)"""

    def __init__(self, works, outs, keep, inputs, unpack=None):
        self._works, self._outs, self._keep = works, outs, keep
        self._inputs = inputs            # the (possibly padded) send buffers stay alive until the exchange has read them
        self._unpack = unpack            # one packed exchange: (rows per rank, T) of the [world, B T + B] block

    def wait(self):
        for w in self._works:
            w.wait()

    def result(self):
        self.wait()
        if self._unpack is not None:
            B, T = self._unpack
            flat = self._outs[0]
            return flat[:, :B * T].reshape(-1, T), flat[:, B * T:].reshape(-1)
        if self._keep is None:
            return self._outs
        return tuple(o[self._keep] for o in self._outs)


def ids_lens_buffer(B, T, device):
    """(ids int32 [B, T], lens int32 [B]) as two views of ONE flat buffer [ids | lens]: all_gather_ids(async_op=True) sends such a
    pair as a single collective (the exchange is latency-bound at 64 KB per rank: two collectives per batch cost twice one)"""
    buf = torch.empty(B * T + B, dtype=torch.int32, device=device)
    return buf[:B * T].view(B, T), buf[B * T:]


def _packed_pair(ids, lens):
    """the flat [ids | lens] buffer behind a pair made by ids_lens_buffer, or None"""
    if not (ids.is_contiguous() and lens.is_contiguous() and ids.dtype == torch.int32 and lens.dtype == torch.int32):
        return None
    if ids.untyped_storage().data_ptr() != lens.untyped_storage().data_ptr():
        return None
    if lens.storage_offset() != ids.storage_offset() + ids.numel() or lens.numel() != ids.shape[0]:
        return None
    return torch.as_strided(ids, (ids.numel() + lens.numel(),), (1,), ids.storage_offset())


def all_gather_ids(ids, lens, group=None, async_op=False, n_total=None):
    """ids int32 [B_local, T] (-1 padded), lens int32 [B_local] -> ([B_total, T], [B_total]) on every rank, in rank order.

    n_total = None: every rank holds the same number of utterances (world | B_total; bench.py's weak-scaling batches).
    all_gather_into_tensor cannot take ragged inputs -- with unequal B_local it fails or, worse, mis-sizes -- so a batch that
    does not divide (the reference's batch wrapper takes ANY batch: ctc_beam_search_decoder.cpp:426-459) passes n_total =
    the global utterance count: every rank pads its shard_range slice to ceil(n_total / world) rows, the padding rows are
    dropped after the exchange (each rank knows every other rank's count from shard_range: no extra collective).

    async_op=True returns ONE handle at once, the same in every case: the collectives run on RCCL's stream behind the kernels
    already queued, the caller's stream goes on with the next batch; `handle.wait()` blocks until the exchange is done and
    `handle.result()` = wait() + (ids_all, lens_all) with any padding rows dropped -- one batch's exchange then overlaps the
    next batch's recognition.  Without n_total nothing is enqueued on the caller's stream (the two tensors go out as they
    are: the caller keeps them untouched until the work is done -- rotating output buffers, recognize(out=...); a pair made by
    ids_lens_buffer goes out as ONE collective); with n_total the shard is first padded to ceil(n_total / world) rows on the
    caller's stream (a cat of at most one row)."""
    world = dist.get_world_size(group)
    B, T = ids.shape
    b_pad = B
    if n_total is not None:
        _check_local_rows(B, n_total, group)
        b_pad = -(-n_total // world)
        ids, lens = _pad_rows(ids, b_pad, -1), _pad_rows(lens, b_pad, 0)
    if async_op and n_total is None:
        flat = _packed_pair(ids, lens)
        if flat is not None:             # ids_lens_buffer: one collective for both
            out = torch.empty(world * flat.numel(), dtype=torch.int32, device=ids.device)
            w = dist.all_gather_into_tensor(out, flat, group=group, async_op=True)
            return GatherWork((w,), (out.view(world, flat.numel()),), None, (ids, lens), unpack=(B, T))
    if async_op:
        all_ids = torch.empty((world * b_pad, T), dtype=ids.dtype, device=ids.device)
        all_lens = torch.empty((world * b_pad,), dtype=lens.dtype, device=lens.device)
        w1 = dist.all_gather_into_tensor(all_ids, ids, group=group, async_op=True)
        w2 = dist.all_gather_into_tensor(all_lens, lens, group=group, async_op=True)
        keep = _valid_rows(n_total, world, b_pad, ids.device) if n_total is not None and n_total != world * b_pad else None
        return GatherWork((w1, w2), (all_ids, all_lens), keep, (ids, lens))
    # one collective per batch: the lengths ride in an extra column of the id matrix (xGMI is latency-, not
    # bandwidth-bound at 64 KB per rank, so the second all_gather would double the exchange time)
    packed = torch.empty((b_pad, T + 1), dtype=ids.dtype, device=ids.device)
    packed[:, :T] = ids
    packed[:, T] = lens.to(ids.dtype)
    out = torch.empty((world * b_pad, T + 1), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    if n_total is not None and n_total != world * b_pad:
        out = out[_valid_rows(n_total, world, b_pad, out.device)]
    return out[:, :T], out[:, T].to(lens.dtype)


def all_gather_hypotheses(ids, lens, scores, n_hyp, group=None, device=None, n_total=None):
    """Beams of a local utterance shard -> the beams of the whole batch on every rank, in rank order.

    ids int32 [B_local, beam, L_local] (-1 padded), lens int32 [B_local, beam], scores float32 [B_local, beam],
    n_hyp int32 [B_local] -- what `ctc_prefix_beam_decode` returns (NumPy or torch).  L_local is data dependent
    (`feature_pick` keeps a different number of frames per batch), so the ranks first agree on max(L_local) with one
    all_reduce(MAX) and pad; then ONE all_gather moves everything: the float32 scores travel as their int32 bit pattern
    in the same block.  n_total as in all_gather_ids (batches that do not divide by the world size; a rank whose shard is
    EMPTY passes arrays with B_local = 0).  Returns NumPy arrays (ids [B_total, beam, L_max], lens, scores, n_hyp)."""
    world = dist.get_world_size(group)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a.cpu().numpy() if torch.is_tensor(a) else a)).to(dt)
    ids_t, lens_t, n_t = t(ids, torch.int32), t(lens, torch.int32), t(n_hyp, torch.int32)
    sc_bits = t(scores, torch.float32).view(torch.int32)
    B, beam, L = ids_t.shape
    b_pad = B
    if n_total is not None:
        _check_local_rows(B, n_total, group)
        b_pad = -(-n_total // world)
    dev = torch.device(device) if device is not None else ids_t.device
    lmax = torch.tensor([L], dtype=torch.int32, device=dev)
    dist.all_reduce(lmax, op=dist.ReduceOp.MAX, group=group)
    Lm = int(lmax.item())
    packed = torch.full((b_pad, beam, Lm + 3), -1, dtype=torch.int32, device=dev)
    packed[:B, :, :L] = ids_t.to(dev)
    packed[:B, :, Lm] = lens_t.to(dev)
    packed[:B, :, Lm + 1] = sc_bits.to(dev)
    packed[:B, :, Lm + 2] = n_t.to(dev)[:, None]
    out = torch.empty((world * b_pad, beam, Lm + 3), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, packed, group=group)
    if n_total is not None and n_total != world * b_pad:
        out = out[_valid_rows(n_total, world, b_pad, out.device)]
    o = out.cpu()
    return (o[:, :, :Lm].numpy().copy(), o[:, :, Lm].numpy().copy(),
            o[:, :, Lm + 1].contiguous().view(torch.float32).numpy().copy(), o[:, 0, Lm + 2].numpy().copy())
