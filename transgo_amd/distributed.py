"""Multi-GPU sharding of self-play: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" in CPU tests).  Games are independent, so the data path has no collective; the only exchange is the gather of
finished games' (s, pi, z) material to the rank that owns the replay buffer -- the Ray `mem.append.remote` traffic of
self_play.py:943-965.  What travels is the exact, compact form: bit-packed observation planes (10*S*S bits), the raw
visit counts (pi = counts/sum is recomputed bit-identically at the owner), side to move, and per game the winner and
the territory map; the 8-fold augmentation happens after the gather, at the consumer."""
import numpy as np
import torch
import torch.distributed as dist


def pack_records(records, S, C):
    """-> uint8 vector.  Layout per game: u32 n_moves, u8 winner, i8 terr[P], then per move: obs bits, i32 counts[A], u8 player."""
    P, A = S * S, S * S + 1
    obs_bytes = (C * P + 7) // 8
    chunks = []
    for r in records:
        n = len(r.players)
        chunks.append(np.array([n], np.uint32).view(np.uint8))
        chunks.append(np.array([r.winner], np.uint8))
        chunks.append(np.asarray(r.territory, np.int8).view(np.uint8))
        for ob, vis, pl in zip(r.observations, r.visits, r.players):
            chunks.append(np.packbits(np.asarray(ob, np.uint8).reshape(-1))[:obs_bytes])
            chunks.append(np.asarray(vis, np.int32).view(np.uint8))
            chunks.append(np.array([pl], np.uint8))
    return np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)


def unpack_records(buf, S, C):
    from .self_play import GameRecord
    P, A = S * S, S * S + 1
    obs_bytes = (C * P + 7) // 8
    out, i = [], 0
    buf = np.asarray(buf, np.uint8)
    while i < len(buf):
        n = int(buf[i:i + 4].view(np.uint32)[0]); i += 4
        r = GameRecord(0)
        r.winner = int(buf[i]); i += 1
        r.territory = buf[i:i + P].view(np.int8).astype(np.float32); i += P
        for _ in range(n):
            ob = np.unpackbits(buf[i:i + obs_bytes])[:C * P].reshape(C, S, S).astype(np.float32); i += obs_bytes
            vis = buf[i:i + 4 * A].view(np.int32).copy(); i += 4 * A
            pl = int(buf[i]); i += 1
            counts = np.array([int(c) for c in vis])
            counts = np.where(counts == 1, 0, counts)                     # self_play.py:666-671
            r.observations.append(ob); r.visits.append(vis); r.pis.append(counts / np.sum(counts)); r.players.append(pl)
        out.append(r)
    return out


def gather_records(records, S, C, dst=0, device=None):
    """Every rank contributes its finished games; rank `dst` returns the list of all of them (others return [])."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(records)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    payload = torch.from_numpy(pack_records(records, S, C)).to(dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([payload.numel()], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if mx == 0:
        return []
    padded = torch.zeros(mx, dtype=torch.uint8, device=dev)
    padded[:payload.numel()] = payload
    bucket = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst)
    if rank != dst:
        return []
    out = []
    for r in range(world):
        out += unpack_records(bucket[r][:sizes[r]].cpu().numpy(), S, C)
    return out


def broadcast_weights(blob, src=0, device=None):
    """Weight refresh for every rank (the Ray `get_info("weights")` of self_play.py:913): one broadcast of the packed,
    BN-folded float32 blob."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return blob
    dev = device if device is not None else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
