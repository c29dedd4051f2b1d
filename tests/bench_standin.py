"""Stand-in engine for CPU rehearsals of bench.py's N-rank control flow (TRANSGO_BENCH_STANDIN=1): the launcher, the process group,
the per-move finished-game exchange, the reductions and the JSON line run exactly as in a real job, with every rank "playing"
through this object instead of the HIP engine -- no GPU, no library.  Test infrastructure: bench.py imports it only under that
environment variable, the line it prints says `"data": "stand-in engine (CPU rehearsal)"`, and nothing it reports is a measurement."""
import numpy as np

from transgo_amd import records
from transgo_amd.self_play import GameRecord, default_seed


class _Cfg:
    arena_slots = 0


class _Ctx:
    cfg = _Cfg()

    def call(self, name, *args):            # tg_prof_*: the ctypes outputs stay at their zero initialisers
        return None


class StandInSelfPlay:
    """advance() finishes `1 + rank % 3` fake games of 2 + rank % 4 plies per step (ragged across ranks, deterministic)."""

    def __init__(self, config, n_games, rank=0, world=1):
        self.G, self.rank, self.world, self.S = n_games, rank, world, config.board_size
        self.seeds = np.array([default_seed(rank, world, n_games, g, 0) for g in range(n_games)], np.uint32)
        self.games_finished = self.games_dropped = 0
        self.phase_s, self.begin_move_s = {}, 0.0
        self.parts, self.engine, self.ctx = [self], self, _Ctx()
        self._sims = self._k = 0

    def set_weights(self, sd):
        return {"weight_absmax": 1.0, "finite": True}

    def start(self, stagger=0):
        pass

    def seed_of(self, g):
        return int(self.seeds[g])

    def advance(self, device=False, **kw):
        rng = np.random.RandomState(1000 * self.rank + self._k)
        self._k += 1
        self._sims += 4 * self.G
        recs = []
        for j in range(1 + self.rank % 3):
            r = GameRecord(int(self.seeds[j % self.G]))
            for m in range(2 + self.rank % 4):
                r.observations.append((rng.rand(10, self.S, self.S) < 0.3).astype(np.float32))
                v = rng.randint(0, 9, self.S * self.S + 1).astype(np.int32); v[5] = 7
                c = np.where(v == 1, 0, v); r.visits.append(v); r.pis.append(c / c.sum()); r.players.append(1 + m % 2)
            r.winner = 1 + j % 2; r.territory = rng.randint(-1, 2, self.S * self.S).astype(np.float32); r.score = 0.5
            recs.append(r)
            self.seeds[j % self.G] = default_seed(self.rank, self.world, self.G, j % self.G, self._k)   # the slot restarts with a fresh seed
        self.games_finished += len(recs)
        return records.from_records(recs, self.S, 10)

    def stats(self):
        return dict(sims=self._sims, evals=self._sims, depth_sum=2 * self._sims, tie_draws=0, errors=0, max_slots=0, truncated_blocks=0,
                    fp16_overflows=0, pool_slots=0, pool_high_water=0, pool_in_use=0, pool_exhausted=0)


class StandInReplay:
    def __init__(self):
        self.entries = 0

    def append_harvest(self, hb):
        self.entries += 8 * hb.n_positions

    def info(self):
        return {"entries": self.entries}
