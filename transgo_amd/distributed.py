"""Multi-GPU sharding of self-play: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" in CPU tests).  Games are independent, so the data path has no collective; the only exchange is the gather of
finished games' (s, pi, z) material to the rank that owns the replay buffer -- the Ray `mem.append.remote` traffic of
self_play.py:943-965.  What travels is `records.Harvest`'s flat buffer: bit-packed observation planes (10*S*S bits), raw
visit counts (pi = counts/sum is recomputed bit-identically at the owner), z, territory from the mover's side and the
per-game tables.  With RCCL the buffer is the very device tensor `tg_sp_harvest` filled, sent rank-to-rank at its exact
length (grouped send/recv, no padding to the largest rank, no host staging); the 8-fold augmentation happens after the
gather, at the consumer."""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import records


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def gather_harvest(h, S, C, dst=0, device_index=0):
    """Every rank contributes the games it finished this step (a records.Harvest or None); rank `dst` returns the list of
    all non-empty batches in rank order (its own included), every other rank returns [].  One tiny all_gather of
    (games, positions) per call; payloads move only when somebody finished a game."""
    if not _active():
        return [h] if h is not None else []
    world, rank = dist.get_world_size(), dist.get_rank()
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device_index) if nccl else torch.device("cpu")
    mine = torch.tensor([h.n_games, h.n_positions] if h is not None else [0, 0], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [tuple(int(x) for x in s.tolist()) for s in sizes]
    if not any(g for g, _ in sizes):
        return []

    def as_tensor(hv):                                   # the payload where the backend can send it from
        b = hv.buf
        if isinstance(b, np.ndarray):
            b = torch.from_numpy(b)
        return b.to(dev) if b.device != dev else b

    if os.environ.get("TRANSGO_GATHER", "p2p") == "allgather":
        # fallback transport (plain collective only): every rank contributes its buffer padded to the longest one
        mx = max(records.layout(S, C, g, n)[1] if g else 0 for g, n in sizes)
        pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
        if h is not None:
            pad[:h.nbytes] = as_tensor(h)
        bucket = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(bucket, pad)
        if nccl:
            torch.cuda.synchronize(dev)
        if rank != dst:
            return []
        out = []
        for r, (g, n) in enumerate(sizes):
            if g:
                nb = records.layout(S, C, g, n)[1]
                out.append(h if r == dst else records.Harvest(S, C, g, n, bucket[r][:nb].clone() if nccl else bucket[r][:nb].numpy().copy()))
        return out
    ops, recv = [], {}
    if rank == dst:
        for r, (g, n) in enumerate(sizes):
            if r != dst and g:
                recv[r] = torch.empty(records.layout(S, C, g, n)[1], dtype=torch.uint8, device=dev)
                ops.append(dist.P2POp(dist.irecv, recv[r], r))
    elif h is not None:
        ops.append(dist.P2POp(dist.isend, as_tensor(h), dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if nccl:
            # wait() on an RCCL work item only orders torch's current stream behind the transfer; the received buffer is consumed
            # next by libtransgo_hip on ITS OWN stream (tg_replay_append_dev), so the transfer must really be over
            torch.cuda.synchronize(dev)
    if rank != dst:
        return []
    out = []
    for r, (g, n) in enumerate(sizes):
        if not g:
            continue
        if r == dst:
            out.append(h)
        else:
            buf = recv[r] if nccl else recv[r].numpy()
            out.append(records.Harvest(S, C, g, n, buf))
    return out


def broadcast_weights(blob, src=0, device=None):
    """Weight refresh for every rank (the Ray `get_info("weights")` of self_play.py:913): one broadcast of the packed,
    BN-folded float32 blob."""
    if not _active():
        return blob
    dev = device if device is not None else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
