"""Finished games as one position-major batch -- what `tg_sp_harvest` writes, what travels between GPUs, what
`tg_replay_append(_dev)` stores, and (unpacked) what the reference's `mem.append` receives (self_play.py:929-967).

One flat byte buffer per batch, either a NumPy array (host) or a torch uint8 tensor in HBM (the RCCL gather sends it as it
is and the device replay store copies out of it, so finished games never visit the host on that route):

    n_moves i32[g] | winner i32[g] | slot i32[g] | seed u32[g] | score f32[g] | terr i8[g][P] (padded to 4 B)
    obs_bits u32[n][W] | counts i32[n][A] | z f32[n] | own i8[n][P] (padded) | player u8[n] (padded)

g = games, n = positions (sum of n_moves), W = ceil(C*P/32); positions are game after game, plies ascending.  obs_bits is
env.encode(root) with bit i = plane-major flat index i; counts are raw root visit counts (pi = counts with 1 -> 0, / sum,
self_play.py:666-671); z = +1 when the mover is the winner else -1 (:931-934); own = territory from the mover's side
(:938-940); terr = getTerritory of the final position (+1 black, 0, -1 white).
"""
import ctypes

import numpy as np


def _pad4(n):
    return (n + 3) & ~3


def layout(S, C, n_games, n_positions):
    """-> ({name: (byte offset, dtype, shape)}, total bytes)"""
    P, A = S * S, S * S + 1
    W = (C * P + 31) // 32
    g, n = n_games, n_positions
    sec, o = {}, 0
    for name, dt, shape in (("n_moves", np.int32, (g,)), ("winner", np.int32, (g,)), ("slot", np.int32, (g,)),
                            ("seed", np.uint32, (g,)), ("score", np.float32, (g,)), ("terr", np.int8, (g, P)),
                            ("obs_bits", np.uint32, (n, W)), ("counts", np.int32, (n, A)), ("z", np.float32, (n,)),
                            ("own", np.int8, (n, P)), ("player", np.uint8, (n,))):
        sec[name] = (o, np.dtype(dt), shape)
        o += _pad4(int(np.prod(shape)) * np.dtype(dt).itemsize)
    return sec, o


def header_bytes(S, C, n_games):
    sec, _ = layout(S, C, n_games, 0)
    return sec["obs_bits"][0]


class Harvest:
    """A batch of finished games.  `buf` is a NumPy uint8 array or a torch uint8 tensor (any device)."""

    def __init__(self, S, C, n_games, n_positions, buf):
        self.S, self.C, self.n_games, self.n_positions, self.buf = S, C, int(n_games), int(n_positions), buf
        self.sec, self.nbytes = layout(S, C, self.n_games, self.n_positions)
        assert len(buf) == self.nbytes, (len(buf), self.nbytes)

    @property
    def on_device(self):
        return not isinstance(self.buf, np.ndarray) and self.buf.is_cuda

    def to_host(self):
        if isinstance(self.buf, np.ndarray):
            return self
        return Harvest(self.S, self.C, self.n_games, self.n_positions, self.buf.cpu().numpy())

    def view(self, name):
        """NumPy view of a section (host batches only)."""
        o, dt, shape = self.sec[name]
        n = int(np.prod(shape))
        return self.buf[o:o + n * dt.itemsize].view(dt).reshape(shape)

    def ptr(self, name):
        """Raw address of a section (host or device)."""
        base = self.buf.ctypes.data if isinstance(self.buf, np.ndarray) else self.buf.data_ptr()
        return ctypes.c_void_p(base + self.sec[name][0])

    # ---- host-side consumers ----------------------------------------------------------------------------------------------
    def observations(self):
        """f32 [n][C][S][S] = env.encode of every recorded root."""
        h = self.to_host()
        bits = np.unpackbits(np.ascontiguousarray(h.view("obs_bits")).view(np.uint8), axis=1, bitorder="little")
        return bits[:, :self.C * self.S * self.S].reshape(-1, self.C, self.S, self.S).astype(np.float32)

    def pis(self):
        """f64 [n][A]: counts == 1 -> 0, counts / sum (self_play.py:666-671); row-wise identical to the per-game form."""
        counts = self.to_host().view("counts").astype(np.int64)
        counts = np.where(counts == 1, 0, counts)
        return counts / np.sum(counts, axis=1)[:, None]

    def records(self):
        """-> list of GameRecord (the per-game Python lists the reference keeps, self_play.py:917-926)."""
        from .self_play import GameRecord
        h = self.to_host()
        obs, pis, cnt, pl = h.observations(), h.pis(), h.view("counts"), h.view("player")
        nm, win, terr, score, seed = h.view("n_moves"), h.view("winner"), h.view("terr"), h.view("score"), h.view("seed")
        out, o = [], 0
        for i in range(h.n_games):
            n = int(nm[i])
            r = GameRecord(int(seed[i]))
            r.observations = list(obs[o:o + n]); r.pis = list(pis[o:o + n]); r.visits = list(cnt[o:o + n].copy())
            r.players = [int(x) for x in pl[o:o + n]]
            r.winner, r.territory, r.score = int(win[i]), terr[i].astype(np.float32), float(score[i])
            out.append(r)
            o += n
        return out

    def targets(self):
        """Every tuple the reference appends for these games, in its order (self_play.py:943-965): per game, per move,
        for i in 1..4: rot90(i) then fliplr of that.  Built with whole-batch rot90/flip; element for element equal to
        transgo_amd.self_play.game_targets (tests/test_host_logic.py)."""
        h = self.to_host()
        S, P = self.S, self.S * self.S
        n = h.n_positions
        obs, pis = h.observations(), h.pis()
        z = h.view("z").astype(np.float64)
        own = h.view("own").astype(np.float64).reshape(n, S, S)
        board_p, pass_p = pis[:, :P].reshape(n, S, S), pis[:, P:]
        syms = []
        for i in (1, 2, 3, 4):
            ro, rp, rw = np.rot90(obs, i, axes=(2, 3)), np.rot90(board_p, i, axes=(1, 2)), np.rot90(own, i, axes=(1, 2))
            syms.append((ro, rp, rw))
            syms.append((ro[..., ::-1], rp[..., ::-1], rw[..., ::-1]))
        so = np.stack([np.ascontiguousarray(s[0]) for s in syms], 1)                                   # [n][8][C][S][S]
        sp = np.stack([np.concatenate([s[1].reshape(n, P), pass_p], 1) for s in syms], 1)              # [n][8][A]
        sw = np.stack([s[2].reshape(n, P) for s in syms], 1)                                           # [n][8][P]
        out = []
        for t in range(n):
            for k in range(8):
                out.append((so[t, k], sp[t, k], z[t], sw[t, k]))
        return out


def from_arrays(S, C, n_moves, winner, terr, obs_bits, counts, z, own, player, slot=None, seed=None, score=None):
    """Assemble a host batch from separate arrays (tests, file loaders)."""
    g, n = len(n_moves), int(np.sum(n_moves))
    sec, total = layout(S, C, g, n)
    h = Harvest(S, C, g, n, np.zeros(total, np.uint8))
    vals = dict(n_moves=n_moves, winner=winner, terr=terr, obs_bits=obs_bits, counts=counts, z=z, own=own, player=player,
                slot=np.zeros(g) if slot is None else slot, seed=np.zeros(g) if seed is None else seed,
                score=np.zeros(g) if score is None else score)
    for k, v in vals.items():
        h.view(k)[...] = np.asarray(v).reshape(sec[k][2])
    return h


def from_records(recs, S, C):
    """GameRecord objects -> a host batch (the inverse of Harvest.records())."""
    P = S * S
    W = (C * P + 31) // 32
    n_moves = [len(r.players) for r in recs]
    n = int(np.sum(n_moves)) if recs else 0
    bits = np.zeros((n, W * 32), np.uint8)
    if n:
        bits[:, :C * P] = np.concatenate([np.asarray(r.observations, np.uint8).reshape(len(r.players), -1) for r in recs])
    obs_bits = np.packbits(bits, axis=1, bitorder="little").view(np.uint32).reshape(n, W)
    counts = np.concatenate([np.asarray(r.visits, np.int32).reshape(-1, P + 1) for r in recs]) if n else np.zeros((0, P + 1), np.int32)
    player = np.concatenate([np.asarray(r.players, np.uint8) for r in recs]) if n else np.zeros(0, np.uint8)
    winner = np.array([r.winner for r in recs], np.int32)
    terr = np.array([np.asarray(r.territory) for r in recs], np.int8).reshape(len(recs), P)
    wpos = np.repeat(winner, n_moves)
    z = np.where(player == wpos, 1.0, -1.0).astype(np.float32)                                   # self_play.py:931-934
    tpos = np.repeat(terr, n_moves, axis=0)
    own = np.where((player == 1)[:, None], tpos, -tpos).astype(np.int8)                          # self_play.py:938-940
    return from_arrays(S, C, np.asarray(n_moves, np.int32), winner, terr, obs_bits, counts, z, own, player,
                       seed=np.array([r.seed for r in recs], np.uint32),
                       score=np.array([0.0 if r.score is None else r.score for r in recs], np.float32))
