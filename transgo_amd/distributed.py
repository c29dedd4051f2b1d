"""Multi-GPU sharding of self-play: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" in CPU tests).  Games are independent, so the data path has no collective; the only exchange is the gather of
finished games' (s, pi, z) material to the rank that owns the replay buffer -- the Ray `mem.append.remote` traffic of
self_play.py:943-965.  What travels is `records.Harvest`'s flat buffer: bit-packed observation planes (10*S*S bits), raw
visit counts (pi = counts/sum is recomputed bit-identically at the owner), z, territory from the mover's side and the
per-game tables.  With RCCL the buffer is the very device tensor `tg_sp_harvest` filled (no host staging); the 8-fold
augmentation happens after the gather, at the consumer.

Transport of the payloads (TRANSGO_GATHER): "allgather" (default) = ONE plain `all_gather` of the buffers padded to the
longest rank's -- the collective every RCCL build runs on first contact; ~2 MB per rank and move at C2 against a 3.4-s move,
so the padding costs nothing measurable.  "p2p" = grouped isend/irecv to the owner at exact lengths (no padding, nothing
delivered to ranks that do not need it): opt-in until it has run between two RCCL ranks on hardware (it is covered over
gloo).  `transport_name()` is what bench.py prints in `ranks.transport`."""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import records


def _transport():
    t = os.environ.get("TRANSGO_GATHER", "allgather")
    if t not in ("allgather", "p2p"):
        raise ValueError(f"TRANSGO_GATHER={t!r}: expected 'allgather' or 'p2p'")
    return t


def transport_name():
    return {"allgather": "all_gather of harvest buffers padded to the longest rank's (default)",
            "p2p": "grouped isend/irecv to the owner at exact lengths (TRANSGO_GATHER=p2p)"}[_transport()]


def _active():
    """True when there is a process group to talk to.  A group of ONE rank counts only with TRANSGO_DIST_SINGLE_RANK=1: that is how
    the RCCL code paths (device-resident control words, size exchange, weight broadcast, device -> device weight load) are
    exercised on a one-GPU box, where a second RCCL rank cannot exist (tests/test_gpu_rccl.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("TRANSGO_DIST_SINGLE_RANK", "0") == "1"


def init_process_group(backend, rank, world, device_index=None, timeout_s=1800.0, init_method=None):
    """torch.distributed.init_process_group with the two things the actor loop relies on spelt out: an EXPLICIT timeout (every
    collective of the loop is short -- a control word, a size exchange, a payload -- so a rank that waits longer than this has
    lost a peer; gloo then raises in the waiting rank, RCCL's watchdog aborts it: either way the process ends non-zero instead
    of hanging), and, for RCCL, the device bound at creation (eager communicator, no lazy init inside the first collective)."""
    import datetime
    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(timeout_s)))
    if init_method:
        kw["init_method"] = init_method
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's hosts share GPU memory through dmabuf handles only
        dev = torch.device("cuda", 0 if device_index is None else device_index)
        torch.cuda.set_device(dev)
        kw["device_id"] = dev
    dist.init_process_group(backend, **kw)


def control_exchange(values, src=0, device_index=0):
    """One small int64 broadcast from `src`: the per-move control word of the actor loop (wait / weights-follow flags).  Every
    rank passes a list of the same length (only src's content counts) and gets src's values back."""
    if not _active():
        return [int(v) for v in values]
    dev = torch.device("cuda", device_index) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return [int(v) for v in t.tolist()]


def gather_harvest(h, S, C, dst=0, device_index=0, live=None):
    """Every rank contributes the games it finished this step (a records.Harvest or None); rank `dst` returns the list of
    all non-empty batches in rank order (its own included), every other rank returns [].  One tiny all_gather of
    (games, positions, live) per call; payloads move only when somebody finished a game.  `live` (optional) = how many of this
    rank's slots really played a move this step; with it the call returns (batches, sum of live over all ranks) -- what
    now_play_steps advances by (self_play.py:928 counts moves that were played, not slots)."""
    if not _active():
        out = [h] if h is not None else []
        return out if live is None else (out, int(live))
    world, rank = dist.get_world_size(), dist.get_rank()
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device_index) if nccl else torch.device("cpu")
    mine = torch.tensor(([h.n_games, h.n_positions] if h is not None else [0, 0]) + [int(live or 0)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes3 = [tuple(int(x) for x in s.tolist()) for s in sizes]
    live_total = sum(x[2] for x in sizes3)
    sizes = [x[:2] for x in sizes3]
    batches = _gather_payloads(h, S, C, dst, sizes, world, rank, nccl, dev)
    return batches if live is None else (batches, live_total)


def _gather_payloads(h, S, C, dst, sizes, world, rank, nccl, dev):
    if not any(g for g, _ in sizes):
        return []

    def as_tensor(hv):                                   # the payload where the backend can send it from
        b = hv.buf
        if isinstance(b, np.ndarray):
            b = torch.from_numpy(b)
        return b.to(dev) if b.device != dev else b

    if _transport() == "allgather":
        # default transport (plain collective only): every rank contributes its buffer padded to the longest one
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
                # over RCCL a VIEW of the gathered buffer (it lives as long as the Harvest does): the synchronize above covers the
                # collective, and nothing else is queued on torch's stream that the library -- which reads the buffer next on ITS
                # OWN stream (tg_replay_append_dev) -- would have to wait for (a clone made here would be exactly that)
                out.append(h if r == dst else records.Harvest(S, C, g, n, bucket[r][:nb] if nccl else bucket[r][:nb].numpy().copy()))
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


def broadcast_weights(blob, src=0, device=None, n_floats=None):
    """Weight refresh for every rank (the Ray `get_info("weights")` of self_play.py:913): one broadcast of the packed,
    BN-folded float32 blob.  Rank `src` passes the blob (NumPy); the others may pass None with `n_floats`.  Returns the blob
    WHERE THE BACKEND DELIVERED IT: over RCCL a float32 tensor in this rank's GPU memory, complete on return (the stream is
    synchronised), which goes to the network device -> device (tg_net_load_async_dev: no host bounce on the receiving ranks);
    over gloo a NumPy array."""
    if not _active():
        return blob
    dev = device if device is not None else torch.device("cpu")
    if blob is not None:
        t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(dev)
    else:
        t = torch.empty(int(n_floats), dtype=torch.float32, device=dev)
    dist.broadcast(t, src=src)
    if t.is_cuda:
        torch.cuda.synchronize(dev)         # the consumer is libtransgo_hip on its own streams: the payload must really be there
        return t
    return t.numpy()
