"""SelfPlayEngine -- host driver of the batched WP_MCTS engine in libtransgo_hip.so.

Mirrors, for G games at once, the per-game objects of the reference: `WP_MCTS.reset_root / get_action_probs /
update_with_action` (self_play.py:595-605, :657-687, :857-872).  Tree search, rules and the network run on the GPU;
this class only sequences the phases and does the one part the reference itself does in NumPy float64 on the host --
turning visit counts into pi and the sampled move (self_play.py:666-683) -- with the same NumPy calls, so those are
bit-identical by construction.  The random numbers for it come from the game's own MT19937 stream inside the library.
"""
import ctypes
import math
import time

import numpy as np

from . import _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def temperature(game_step):
    """Config.epsilon_by_frame (configure.py:75-79)."""
    return 0.65 + (1.0 - 0.65) * math.exp(-1. * game_step / 10)


def choose_moves_reference(visits, steps, u, live, selfplay=True):
    """Literal per-game form of self_play.py:666-683 (the definition choose_moves_batch is tested against)."""
    G, A = visits.shape
    actions = np.zeros(G, np.int32)
    pis = np.zeros((G, A), np.float64)
    for g in np.flatnonzero(live):
        counts = np.array([int(c) for c in visits[g]])
        counts = np.where(counts == 1, 0, counts)
        pis[g] = counts / np.sum(counts)
        tau = temperature(int(steps[g])) if selfplay else 0.12
        powed = np.power(counts, 1.0 / tau)
        probs = np.array(powed) / np.sum(powed)
        cdf = probs.cumsum(); cdf /= cdf[-1]
        actions[g] = cdf.searchsorted(u[g], side="right")
    return actions, pis


def _choose_rows(counts_raw, steps, u, selfplay):
    """self_play.py:666-683 for a block of live rows (see choose_moves_batch)."""
    counts = counts_raw.astype(np.int64)
    counts = np.where(counts == 1, 0, counts)
    # No child visited twice (only possible when num_simulation is far below the number of legal moves): the reference
    # divides 0 by 0 here and np.random.choice raises on the NaN probabilities (self_play.py:671-683).  A batched engine
    # cannot afford one degenerate board stopping thousands, so such a row keeps its raw counts instead.
    dead = np.sum(counts, axis=1) == 0
    if dead.any():
        counts[dead] = counts_raw[dead]
    tot = np.sum(counts, axis=1)
    empty = tot == 0                                     # a slot that was never searched (parked / already over): pi = 0, action 0
    if empty.any():
        counts = counts.copy(); counts[empty, 0] = 1
        tot = np.sum(counts, axis=1)
    pis = counts / tot[:, None]
    if selfplay:
        inv_tau = 1.0 / (0.65 + (1.0 - 0.65) * np.array([math.exp(-1. * int(s) / 10) for s in steps]))     # configure.py:75-79, per game
    else:
        inv_tau = np.full(len(steps), 1.0 / 0.12)
    powed = np.power(counts, inv_tau[:, None])
    probs = powed / np.sum(powed, axis=1)[:, None]
    cdf = np.cumsum(probs, axis=1)
    cdf /= cdf[:, -1][:, None]
    acts = (cdf <= u[:, None]).sum(axis=1)                      # searchsorted(u, 'right') on a non-decreasing row
    if empty.any():
        acts[empty] = 0; pis[empty] = 0.0
    return acts, pis


_POOL = None


def choose_moves_batch(visits, steps, u, live, selfplay=True):
    """Every operation is element-wise or a last-axis reduction, so each row equals the per-game 1-D computation bit for bit;
    large batches are cut into row blocks handled by a few threads (NumPy releases the GIL inside its loops)."""
    global _POOL
    G, A = visits.shape
    actions = np.zeros(G, np.int32)
    pis = np.zeros((G, A), np.float64)
    idx = np.flatnonzero(live)
    if len(idx) == 0:
        return actions, pis
    nblk = min(8, len(idx) // 256)
    if nblk <= 1:
        actions[idx], pis[idx] = _choose_rows(visits[idx], steps[idx], u[idx], selfplay)
        return actions, pis
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=8)
    parts = np.array_split(idx, nblk)
    for part, (a, p) in zip(parts, _POOL.map(lambda ix: _choose_rows(visits[ix], steps[ix], u[ix], selfplay), parts)):
        actions[part] = a; pis[part] = p
    return actions, pis


class SelfPlayEngine:
    def __init__(self, n_games, board_size=9, num_simulation=210, parallel_readouts=4, c_puct1=3, c_puct2=0.05,
                 wu_loss=2, komi=7.5, max_step=120, encode_dim=10, net_blocks=6, net_filters=128, arena_slots=0,
                 device=0, evaluator=None, net_precision="f32", record_games=True, pool_slots=0):
        cfg = _lib.default_config()
        cfg.record_games = 1 if record_games else 0       # per-move records in HBM for tg_sp_harvest (self-play); evaluation matches need none
        cfg.net_precision = {"f32": 0, "f16": 1, "f16r": 2, "f32x3": 3}[net_precision]
        cfg.board_size, cfg.encode_dim, cfg.max_step, cfg.komi = board_size, encode_dim, max_step, komi
        cfg.n_games, cfg.num_simulation, cfg.parallel_readouts, cfg.wu_loss = n_games, num_simulation, parallel_readouts, wu_loss
        cfg.c_puct1, cfg.c_puct2, cfg.arena_slots = c_puct1, c_puct2, arena_slots
        cfg.pool_slots = pool_slots                       # tree memory: slots per game ON AVERAGE in the pool all games share (0 = default)
        cfg.net_blocks, cfg.net_filters, cfg.device = net_blocks, net_filters, device
        self.ctx = _lib.Context(cfg)
        self.G, self.S, self.P, self.A, self.C = n_games, board_size, board_size ** 2, board_size ** 2 + 1, encode_dim
        self.num_simulation = num_simulation
        self.evaluator = evaluator            # None -> network on the GPU; callable(obs)->(policy, value) -> injected
        self.finished = np.zeros(n_games, bool)      # slots that take no part in the next search (game over, parked, in error)
        self.errored = np.zeros(n_games, bool)       # slots parked by a tree-arena overflow (restart them with reset(mask))
        self.device = device
        self.begin_move_s = 0.0                      # wall clock spent in tg_sp_begin_move (root noise on the host), for bench.py

    def close(self):
        self.ctx.close()

    # ---- evaluation of the pending batch --------------------------------------------------------------------------------
    def _evaluate(self, rows=None):
        if self.evaluator is None:
            self.ctx.call("tg_sp_eval")
            return
        if rows is None:
            n = ctypes.c_int32()
            self.ctx.call("tg_sp_batch_rows", ctypes.byref(n))
            rows = n.value
        if rows == 0:
            return
        obs = np.empty((rows, self.C, self.S, self.S), np.float32)
        self.ctx.call("tg_sp_batch_obs", _ptr(obs), rows)
        policy, value = self.evaluator(obs)
        policy = np.ascontiguousarray(policy, np.float32); value = np.ascontiguousarray(value, np.float32).reshape(-1)
        assert policy.shape == (rows, self.A) and value.shape == (rows,)
        self.ctx.call("tg_sp_set_eval", _ptr(policy), _ptr(value), rows)

    # ---- reference-shaped operations, G games at a time -------------------------------------------------------------------
    def reset(self, seeds, mask=None):
        """reset_root (self_play.py:595-605) + np.random.seed(seed) per game."""
        seeds = np.ascontiguousarray(seeds, np.uint32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.ctx.call("tg_sp_reset", _ptr(seeds), _ptr(m))
        self._evaluate()
        self.ctx.call("tg_sp_expand_roots")
        if mask is None:
            self.finished[:] = False; self.errored[:] = False
        else:
            self.finished[np.asarray(mask, bool)] = False; self.errored[np.asarray(mask, bool)] = False

    def reset_from(self, states, mask=None):
        """Fresh roots at given positions: the first half of select_action (self_play.py:689-700).  states: [G, state_size]
        uint8 blobs (transgo_amd.environment.GoEnv states); unmasked slots are parked."""
        st = np.ascontiguousarray(states, np.uint8)
        assert st.shape[0] == self.G
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.ctx.call("tg_sp_reset_from", _ptr(st), _ptr(m))
        self._evaluate()
        self.ctx.call("tg_sp_expand_roots")
        self.finished = np.zeros(self.G, bool) if mask is None else ~np.asarray(mask, bool)

    def root_states(self):
        st = np.zeros((self.G, self.ctx.state_size), np.uint8)
        self.ctx.call("tg_sp_root_states", _ptr(st))
        return st

    def select_action(self, states, mask=None):
        """WP_MCTS.select_action (self_play.py:689-703) for G positions at once: fresh tree, no root noise, temperature
        0.12.  Returns the chosen actions (0 for parked slots)."""
        self.reset_from(states, mask)
        self.search(selfplay=False)
        vis, _, _, steps, _ = self.root_info(obs=False)
        actions, _ = self.choose_moves(vis, steps, selfplay=False)
        return actions

    def search(self, selfplay=True, num_simulation=0):
        """The search part of get_action_probs (self_play.py:659-664)."""
        t0 = time.perf_counter()
        self.ctx.call("tg_sp_begin_move", 1 if selfplay else 0, num_simulation)
        self.begin_move_s += time.perf_counter() - t0
        if self.evaluator is None:
            w = ctypes.c_int32()
            self.ctx.call("tg_sp_search", ctypes.byref(w))
            return w.value
        waves = 0
        while True:
            act, rows = ctypes.c_int32(), ctypes.c_int32()
            self.ctx.call("tg_sp_collect", ctypes.byref(act), ctypes.byref(rows))
            self._evaluate(rows.value)
            self.ctx.call("tg_sp_absorb")
            if act.value == 0:
                return waves
            waves += 1

    def root_info(self, obs=True):
        vis = np.zeros((self.G, self.A), np.int32)
        rn = np.zeros(self.G, np.int32); pl = np.zeros(self.G, np.int32); st = np.zeros(self.G, np.int32)
        ob = np.zeros((self.G, self.C, self.S, self.S), np.float32) if obs else None
        self.ctx.call("tg_sp_root_info", _ptr(vis), _ptr(rn), _ptr(pl), _ptr(st), _ptr(ob))
        return vis, rn, pl, st, ob

    def root_visits(self):
        """Visit counts and ply counters only (what move selection needs): 4*(A+1) bytes per game across PCIe."""
        vis = np.zeros((self.G, self.A), np.int32); st = np.zeros(self.G, np.int32)
        self.ctx.call("tg_sp_root_info", _ptr(vis), None, None, _ptr(st), None)
        return vis, st

    def choose_moves(self, visits, steps, selfplay=True):
        """self_play.py:666-683 for every live game, in NumPy float64 with the reference's own operations:
        counts==1 -> 0, pi = counts/sum, p ~ counts**(1/tau), and np.random.choice(A, p) spelt out as NumPy implements it
        (mtrand.pyx: cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(random_sample(), 'right')) with the uniform
        drawn from the game's own stream.  Vectorised over games; every operation is element-wise or a last-axis
        reduction, so each row equals the per-game 1-D computation bit for bit (tests/test_host_logic.py checks it
        against the literal per-game form, choose_moves_reference)."""
        live = ~self.finished
        u = np.zeros(self.G, np.float64)
        self.ctx.call("tg_sp_draw_uniform", _ptr(u), _ptr(live.astype(np.uint8)))
        return choose_moves_batch(visits, steps, u, live, selfplay)

    def play(self, actions):
        """update_with_action (self_play.py:857-872); the move's record entry is written on the device first.  Returns the
        mask of games that are over; `self.errored` marks slots parked by an arena overflow (neither over nor playable)."""
        done = np.zeros(self.G, np.uint8)
        self.ctx.call("tg_sp_play", _ptr(np.ascontiguousarray(actions, np.int32)), _ptr(done))
        self._evaluate()
        self.ctx.call("tg_sp_expand_roots")
        self.finished = done != 0
        self.errored = done == 2
        return done == 1

    def game_errors(self):
        n = ctypes.c_int32(); err = np.zeros(self.G, np.int32)
        self.ctx.call("tg_sp_game_errors", ctypes.byref(n), _ptr(err))
        return err

    def harvest(self, device=False, seeds=None):
        """The games the last play() finished, as one position-major batch (transgo_amd.records.Harvest), or None.
        device=True: the batch is a torch uint8 tensor in this GPU's memory and only the per-game tables cross PCIe."""
        from . import records
        ng, npos = ctypes.c_int32(), ctypes.c_int32()
        self.ctx.call("tg_sp_finished", ctypes.byref(ng), ctypes.byref(npos))
        ng, npos = ng.value, npos.value
        if ng == 0:
            return None
        sec, total = records.layout(self.S, self.C, ng, npos)
        hdr = records.header_bytes(self.S, self.C, ng)
        if device:
            import torch
            # zeros, not empty: the layout has alignment padding nobody writes, and the batch is compared / hashed / sent as bytes
            buf = torch.zeros(total, dtype=torch.uint8, device=torch.device("cuda", self.device))
            host = records.Harvest(self.S, self.C, ng, 0, np.zeros(hdr, np.uint8))        # per-game tables are written on the host
        else:
            buf = np.zeros(total, np.uint8)
        h = records.Harvest(self.S, self.C, ng, npos, buf)
        t = host if device else h
        self.ctx.call("tg_sp_harvest", h.ptr("obs_bits"), h.ptr("counts"), h.ptr("z"), h.ptr("own"), h.ptr("player"),
                      1 if device else 0, t.ptr("slot"), t.ptr("n_moves"), t.ptr("winner"), t.ptr("score"), t.ptr("terr"))
        if seeds is not None:
            t.view("seed")[:] = np.asarray(seeds, np.uint32)[t.view("slot")]
        if device:
            buf[:hdr].copy_(torch.from_numpy(host.buf))
        return h

    def final(self):
        score = np.zeros(self.G, np.float32); terr = np.zeros((self.G, self.P), np.float32); win = np.zeros(self.G, np.int32)
        self.ctx.call("tg_sp_final", _ptr(score), _ptr(terr), _ptr(win))
        return score, terr, win

    def rng_state(self, game):
        s = _lib.TgMt19937()
        self.ctx.call("tg_sp_rng_state", int(game), ctypes.byref(s))
        return np.frombuffer(s.key, np.uint32).copy(), int(s.pos)

    def rng_streams(self):
        """All G MT19937 streams as a uint8 array [G][sizeof(tg_mt19937)] (opaque; hand it to set_rng_streams)."""
        out = np.zeros((self.G, ctypes.sizeof(_lib.TgMt19937)), np.uint8)
        self.ctx.call("tg_sp_rng_get", _ptr(out))
        return out

    def set_rng_streams(self, streams, mask=None):
        st = np.ascontiguousarray(streams, np.uint8)
        assert st.shape == (self.G, ctypes.sizeof(_lib.TgMt19937))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.ctx.call("tg_sp_rng_set", _ptr(st), _ptr(m))

    def stats(self):
        v = [ctypes.c_uint64() for _ in range(4)]; e = ctypes.c_int32(); m = ctypes.c_int32()
        self.ctx.call("tg_sp_stats", *[ctypes.byref(x) for x in v], ctypes.byref(e), ctypes.byref(m))
        tr = ctypes.c_uint64()
        self.ctx.call("tg_sp_tree_truncations", ctypes.byref(tr))
        out = dict(sims=v[0].value, evals=v[1].value, depth_sum=v[2].value, tie_draws=v[3].value, errors=e.value,
                   max_slots=m.value, truncated_blocks=tr.value, fp16_overflows=0)
        out.update(self.pool_stats())
        if self.evaluator is None:
            out["fp16_overflows"] = self.net_range()["fp16_overflows"]
        return out

    def pool_stats(self):
        """tg_sp_pool_stats: the tree pool all games share -- its size, the most that was in use at once, what is in use now (32-byte
        slots) and how often a game found it empty."""
        v = [ctypes.c_uint64() for _ in range(4)]
        self.ctx.call("tg_sp_pool_stats", *[ctypes.byref(x) for x in v])
        return dict(pool_slots=v[0].value, pool_high_water=v[1].value, pool_in_use=v[2].value, pool_exhausted=v[3].value)

    def net_range(self):
        """tg_net_range: sticky count of output tiles in which a value beyond +-65504 was rounded to fp16 (net_precision 1-3; 0 = every
        forward pass so far stayed in range) and the largest |w| of the live BN-folded weight blob."""
        n = ctypes.c_uint64(); w = ctypes.c_float()
        if self.ctx.lib.tg_net_range(self.ctx.h, ctypes.byref(n), ctypes.byref(w)) != 0:
            return {"fp16_overflows": 0, "weight_absmax": None}              # no network loaded in this context (yet)
        return {"fp16_overflows": int(n.value), "weight_absmax": float(w.value)}
