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


def all_gather_ids(ids, lens, group=None, async_op=False):
    """ids int32 [B_local, T] (-1 padded), lens int32 [B_local] -> ([B_total, T], [B_total]) on every rank,
    in rank order (requires equal B_local on all ranks, which shard_range gives when world | B_total).

    async_op=True returns (work, ids_all, lens_all) at once: the collective runs on RCCL's stream behind the kernels
    already queued, the caller's stream goes on with the next batch and `work.wait()` (or a device synchronise) makes
    the views valid -- one batch's exchange then overlaps the next batch's recognition."""
    world = dist.get_world_size(group)
    B, T = ids.shape
    if async_op:
        # nothing on the caller's stream: the two tensors go out as they are (the caller keeps them untouched until the
        # work is done -- rotating output buffers, recognize(out=...)), both collectives on RCCL's stream
        all_ids = torch.empty((world * B, T), dtype=ids.dtype, device=ids.device)
        all_lens = torch.empty((world * B,), dtype=lens.dtype, device=lens.device)
        w1 = dist.all_gather_into_tensor(all_ids, ids, group=group, async_op=True)
        w2 = dist.all_gather_into_tensor(all_lens, lens, group=group, async_op=True)

        class _Both:
            def wait(self):
                w1.wait()
                w2.wait()
        return _Both(), all_ids, all_lens
    # one collective per batch: the lengths ride in an extra column of the id matrix (xGMI is latency-, not
    # bandwidth-bound at 64 KB per rank, so the second all_gather would double the exchange time)
    packed = torch.empty((B, T + 1), dtype=ids.dtype, device=ids.device)
    packed[:, :T] = ids
    packed[:, T] = lens.to(ids.dtype)
    out = torch.empty((world * B, T + 1), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    return out[:, :T], out[:, T].to(lens.dtype)


def all_gather_hypotheses(ids, lens, scores, n_hyp, group=None, device=None):
    """Beams of a local utterance shard -> the beams of the whole batch on every rank, in rank order.

    ids int32 [B_local, beam, L_local] (-1 padded), lens int32 [B_local, beam], scores float32 [B_local, beam],
    n_hyp int32 [B_local] -- what `ctc_prefix_beam_decode` returns (NumPy or torch).  L_local is data dependent
    (`feature_pick` keeps a different number of frames per batch), so the ranks first agree on max(L_local) with one
    all_reduce(MAX) and pad; then ONE all_gather moves everything: the float32 scores travel as their int32 bit pattern
    in the same block.  Returns NumPy arrays (ids [B_total, beam, L_max], lens, scores, n_hyp)."""
    world = dist.get_world_size(group)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a.cpu().numpy() if torch.is_tensor(a) else a)).to(dt)
    ids_t, lens_t, n_t = t(ids, torch.int32), t(lens, torch.int32), t(n_hyp, torch.int32)
    sc_bits = t(scores, torch.float32).view(torch.int32)
    B, beam, L = ids_t.shape
    dev = torch.device(device) if device is not None else ids_t.device
    lmax = torch.tensor([L], dtype=torch.int32, device=dev)
    dist.all_reduce(lmax, op=dist.ReduceOp.MAX, group=group)
    Lm = int(lmax.item())
    packed = torch.full((B, beam, Lm + 3), -1, dtype=torch.int32, device=dev)
    packed[:, :, :L] = ids_t.to(dev)
    packed[:, :, Lm] = lens_t.to(dev)
    packed[:, :, Lm + 1] = sc_bits.to(dev)
    packed[:, :, Lm + 2] = n_t.to(dev)[:, None]
    out = torch.empty((world * B, beam, Lm + 3), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, packed, group=group)
    o = out.cpu()
    return (o[:, :, :Lm].numpy().copy(), o[:, :, Lm].numpy().copy(),
            o[:, :, Lm + 1].contiguous().view(torch.float32).numpy().copy(), o[:, 0, Lm + 2].numpy().copy())
