"""ReplayMemory_Random with the reference's interface (replay_buffer.py:16-94): same method names, argument order and
tuple layout `(obs f32[C,S,S], pi f64[A], z float, own f64[S*S])`, so trainer.py:46-54 (`map(np.stack, zip(*batch))`)
consumes it unchanged.  Host-side container, not a Ray actor: wrap it with `ray.remote(ReplayMemory_Random)` where Ray
exists; transgo_amd.self_play calls `.remote()` when the object offers it."""
import numpy as np


class ReplayMemory_Random:
    def __init__(self, config):
        self.capacity = int(config.buffer_size)
        self.full = False
        self.index = 0
        self.last_save_index = 0
        self.load_index = 0
        # replay_buffer.py:21-27: np.array([blank]*capacity, dtype=object) is a 2-D object array of shape (capacity, 4) -- one
        # column per tuple field, every row referring to the same four blank objects.  Built here by broadcasting one row
        # (the same result without a capacity-long Python list), so save()/load() dicts are interchangeable with the
        # reference's.
        blank = (np.zeros((config.encode_state_channels, config.board_size, config.board_size)),
                 np.zeros((config.board_size ** 2 + 1)), 0.0, np.zeros((config.board_size ** 2)))
        row = np.empty(4, dtype=object)
        for i, v in enumerate(blank):
            row[i] = v
        self.data = np.empty((self.capacity, 4), dtype=object)
        self.data[:] = row

    def append(self, observation, act_prob, win_z, own_z):          # replay_buffer.py:30-34
        row = self.data[self.index]                                 # assigning a tuple would broadcast the arrays' elements
        row[0], row[1], row[2], row[3] = observation, act_prob, win_z, own_z
        self.index = (self.index + 1) % self.capacity
        self.full = self.full or self.index == 0

    def sample(self, batch_size):                                   # replay_buffer.py:36-47
        buffer_len = self.capacity if self.full else self.index
        if buffer_len < batch_size:
            idx = np.random.choice(buffer_len, batch_size)
        else:
            idx = np.random.choice(buffer_len, batch_size, replace=False)
        return self.data[idx]

    def save(self):                                                 # replay_buffer.py:49-72
        if self.capacity <= 1500000:
            return {"buffer_capacity": self.capacity, "index": self.index, "full": self.full, "data": self.data,
                    "save_len": self.capacity}
        if 0 < self.index - self.last_save_index < 300000:
            return False
        if self.index - self.last_save_index < 0 and self.index < 300000:
            return False
        save_len = min(1000000, self.index)
        out = {"buffer_capacity": self.capacity, "data": self.data[self.index - save_len:self.index],
               "save_len": save_len, "index": self.index}
        self.last_save_index = self.index
        return out

    def load(self, infos):                                          # replay_buffer.py:74-87
        save_len = infos["save_len"]
        if self.load_index + save_len < self.capacity:
            self.data[self.load_index:self.load_index + save_len] = infos["data"]
            self.load_index += save_len
        else:
            end_len = self.capacity - self.load_index
            self.data[self.load_index:] = infos["data"][:end_len]
            self.load_index = 0
            self.full = True
        self.index = self.load_index
        return self.full

    def info(self):                                                 # replay_buffer.py:89-94
        return {"capacity": self.capacity, "index": self.index, "full": self.full}


def ReplayMemory(config):                                           # replay_buffer.py:7-10
    return ReplayMemory_Random(config)


class DeviceReplayMemory:
    """Replay store resident in HBM with the sampler on the GPU (libtransgo_hip tg_replay_*).

    Holds each position once, un-augmented; `sample(batch_size)` returns what trainer.py:46-54 builds from the reference
    buffer -- `state f32[B,C,S,S], pi f32[B,A], z f32[B], own f32[B,S*S]` -- for entries drawn exactly as
    ReplayMemory_Random.sample draws them (replay_buffer.py:36-47), where entry e is (position e//8, symmetry e%8) in the
    reference's append order.  `append_game` takes the GameRecord objects transgo_amd.self_play produces."""

    def __init__(self, config, capacity_positions=None, device=0):
        import ctypes
        from . import _lib
        self._ct, self._lib = ctypes, _lib
        cfg = _lib.default_config()
        cfg.board_size, cfg.encode_dim, cfg.n_games, cfg.device = config.board_size, config.encode_state_channels, 0, device
        self.ctx = _lib.Context(cfg)
        self.S, self.C = config.board_size, config.encode_state_channels
        self.P, self.A = self.S ** 2, self.S ** 2 + 1
        self.capacity = int(capacity_positions or max(1, int(config.buffer_size) // 8))
        h = ctypes.c_void_p()
        _lib.check(self.ctx.lib, self.ctx.h, self.ctx.lib.tg_replay_create(self.ctx.h, self.capacity, ctypes.byref(h)))
        self.h = h
        self.words = (self.C * self.P + 31) // 32

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.tg_replay_destroy(self.h); self.h = None
            self.ctx.close()

    def append_game(self, rec):
        n = len(rec.players)
        bits = np.zeros((n, self.words * 32), np.uint8)
        bits[:, :self.C * self.P] = np.stack([np.asarray(o, np.uint8).reshape(-1) for o in rec.observations])
        packed = np.packbits(bits, axis=1, bitorder="little").view(np.uint32)
        counts = np.ascontiguousarray(np.stack(rec.visits), np.int32)
        players = np.asarray(rec.players)
        z = np.where(players == rec.winner, 1.0, -1.0).astype(np.float32)                       # self_play.py:931-934
        terr = np.asarray(rec.territory, np.float32)
        own = np.where((players == 1)[:, None], terr[None, :], -terr[None, :]).astype(np.int8)  # self_play.py:938-940
        p = lambda a: a.ctypes.data_as(self._ct.c_void_p)
        self._lib.check(self.ctx.lib, self.ctx.h, self.ctx.lib.tg_replay_append(self.h, p(np.ascontiguousarray(packed)), p(counts), p(z),
                                                                                p(np.ascontiguousarray(own)), n))

    def append_harvest(self, h):
        """Finished games as a records.Harvest batch: straight from HBM when the batch lives on this GPU (tg_sp_harvest or an
        RCCL gather wrote it there), from host arrays otherwise."""
        fn = self.ctx.lib.tg_replay_append_dev if h.on_device else self.ctx.lib.tg_replay_append
        if not h.on_device:
            h = h.to_host()
        self._lib.check(self.ctx.lib, self.ctx.h, fn(self.h, h.ptr("obs_bits"), h.ptr("counts"), h.ptr("z"), h.ptr("own"),
                                                     h.n_positions))

    def info(self):                                                  # replay_buffer.py:89-94, in units of reference entries
        e, i, f = self._ct.c_longlong(), self._ct.c_longlong(), self._ct.c_int()
        self.ctx.lib.tg_replay_info(self.h, self._ct.byref(e), self._ct.byref(i), self._ct.byref(f))
        return {"capacity": self.capacity * 8, "index": i.value, "full": bool(f.value), "entries": e.value}

    def sample_entries(self, entries):
        entries = np.ascontiguousarray(entries, np.int64)
        B = len(entries)
        state = np.empty((B, self.C, self.S, self.S), np.float32); pi = np.empty((B, self.A), np.float32)
        z = np.empty(B, np.float32); own = np.empty((B, self.P), np.float32)
        p = lambda a: a.ctypes.data_as(self._ct.c_void_p)
        self._lib.check(self.ctx.lib, self.ctx.h, self.ctx.lib.tg_replay_sample(self.h, p(entries), B, p(state), p(pi), p(z), p(own), 0))
        return state, pi, z, own

    def sample_entries_device(self, entries):
        """The same four arrays as torch float32 tensors in THIS GPU's memory (tg_replay_sample with device_out): what
        trainer.py:50-54 builds with torch.FloatTensor(...).to(device), without the host round trip."""
        import torch
        entries = np.ascontiguousarray(entries, np.int64)
        B = len(entries)
        dev = torch.device("cuda", self.ctx.cfg.device)
        state = torch.empty((B, self.C, self.S, self.S), dtype=torch.float32, device=dev); pi = torch.empty((B, self.A), dtype=torch.float32, device=dev)
        z = torch.empty(B, dtype=torch.float32, device=dev); own = torch.empty((B, self.P), dtype=torch.float32, device=dev)
        vp = self._ct.c_void_p
        self._lib.check(self.ctx.lib, self.ctx.h, self.ctx.lib.tg_replay_sample(
            self.h, entries.ctypes.data_as(vp), B, vp(state.data_ptr()), vp(pi.data_ptr()), vp(z.data_ptr()), vp(own.data_ptr()), 1))
        return state, pi, z, own

    def sample_device(self, batch_size):
        return self.sample_entries_device(self._draw(batch_size))

    def _draw(self, batch_size):                                     # replay_buffer.py:36-47: the index draw
        buffer_len = self.info()["entries"]
        if buffer_len < batch_size:
            return np.random.choice(buffer_len, batch_size)
        return np.random.choice(buffer_len, batch_size, replace=False)

    def sample(self, batch_size):                                    # replay_buffer.py:36-47
        buffer_len = self.info()["entries"]
        if buffer_len < batch_size:
            idx = np.random.choice(buffer_len, batch_size)
        else:
            idx = np.random.choice(buffer_len, batch_size, replace=False)
        return self.sample_entries(idx)


# ---- packed on-disk / wire format (replaces the pickled object arrays of replay_buffer.py:49-87) -----------
# One file = one batch of finished games in the exact, compact form that also travels between GPUs and that tg_sp_harvest
# writes (transgo_amd.records): bit-packed planes (10*S*S bits), raw visit counts, z, territory from the mover's side, side
# to move, and per game the winner / territory / seed.  ~0.52 KB per position instead of ~36 KB for its 8 pickled float
# tuples; pi and the 8 symmetries are regenerated bit-identically by the loader.
_MAGIC = b"TGRP2\0"


def save_packed(path, games, board_size, encode_dim):
    """games: a records.Harvest or a list of GameRecord."""
    from . import records
    h = games if isinstance(games, records.Harvest) else records.from_records(games, board_size, encode_dim)
    h = h.to_host()
    with open(path, "wb") as f:
        f.write(_MAGIC)
        f.write(np.array([board_size, encode_dim, h.n_games, h.n_positions], np.int32).tobytes())
        f.write(h.buf.tobytes())


def load_packed_batch(path):
    """-> records.Harvest (host)."""
    from . import records
    with open(path, "rb") as f:
        if f.read(len(_MAGIC)) != _MAGIC:
            raise ValueError("not a packed replay file")
        S, C, g, n = (int(x) for x in np.frombuffer(f.read(16), np.int32))
        buf = np.frombuffer(f.read(), np.uint8).copy()
    if len(buf) != records.layout(S, C, g, n)[1]:
        raise ValueError("truncated packed replay file")
    return records.Harvest(S, C, g, n, buf)


def load_packed(path):
    """-> list of GameRecord (observations, visits, pis, players, winner, territory)."""
    return load_packed_batch(path).records()


def load_packed_into(path, mem):
    """Feed a packed file into any buffer with the reference's append(obs, pi, z, own) (replay_buffer.py:30-34), in the
    reference's order (or into a DeviceReplayMemory as one batch).  Returns the number of reference tuples that makes."""
    h = load_packed_batch(path)
    if hasattr(mem, "append_harvest"):
        mem.append_harvest(h)
        return 8 * h.n_positions
    k = 0
    for t in h.targets():
        mem.append(*t); k += 1
    return k
