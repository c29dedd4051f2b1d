"""SelfPlay -- the reference's self-play actor (self_play.py:881-983) over the batched HIP engine.

`continuous_self_play(shared_storage_worker, mem)` keeps the reference's contract: weights are pulled with
`get_info("weights")`, every move bumps `set_info("now_play_steps")`, every finished game bumps
`set_info("now_play_games")` and appends its 8-fold augmented `(obs, pi, z, own)` tuples with
`mem.append(obs, pi, z, own)` in the reference's order (self_play.py:929-967), so trainer.py consumes the buffer
unchanged.  The difference is inside: G games advance together on the GPU instead of one game per actor.
Storage objects may be Ray actors (methods called through `.remote()`) or plain objects.
"""
import time

import numpy as np

from .engine import SelfPlayEngine
from . import model as _model


def _call(method, *args):
    """obj.method(*args), through .remote() when the object is a Ray actor handle."""
    if hasattr(method, "remote"):
        import ray
        return method.remote(*args)
    return method(*args)


def _get(x):
    try:
        import ray
        return ray.get(x) if isinstance(x, ray.ObjectRef) else x
    except ImportError:
        return x


def game_targets(observations, pis, players, winner, territory, board_size):
    """z / ownership targets and the 8 symmetries, in the reference's append order (self_play.py:931-965):
    for i in 1..4: rot90(i), then fliplr of that."""
    S = board_size
    players = np.array(players)
    z = np.zeros(len(players))
    z[players == winner] = 1
    z[players != winner] = -1
    own = np.zeros((len(players), S * S))
    own[players == 1] = territory
    own[players != 1] = -1 * territory
    out = []
    for ob, pi, zz, ow in zip(observations, pis, z, own):
        board_p, pass_p = pi[:-1], pi[-1]
        for i in (1, 2, 3, 4):
            rp = np.rot90(board_p.reshape(S, S), i)
            ro = np.array([np.rot90(pl, i) for pl in ob])
            rw = np.rot90(ow.reshape(S, S), i)
            out.append((ro, np.append(rp.flatten(), pass_p), zz, rw.flatten()))
            fo = np.array([np.fliplr(pl) for pl in ro])
            out.append((fo, np.append(np.fliplr(rp).flatten(), pass_p), zz, np.fliplr(rw).flatten()))
    return out


class GameRecord:
    __slots__ = ("observations", "pis", "visits", "players", "winner", "territory", "score", "seed")

    def __init__(self, seed):
        self.observations, self.pis, self.visits, self.players = [], [], [], []
        self.winner, self.territory, self.score, self.seed = None, None, None, seed


class BatchedSelfPlay:
    """G concurrent self-play games in lock step (one engine, one GPU)."""

    def __init__(self, config, n_games, device=0, rank=0, world=1, evaluator=None, keep_obs=True, arena_slots=0,
                 seed_fn=None):
        self.config, self.G, self.rank, self.world = config, n_games, rank, world
        self.S = config.board_size
        self.filters = getattr(config, "num_features", 128)
        self.blocks = getattr(config, "num_blocks", 6)
        self.device = device
        # network layout: "tower" = BASELINE.json's N-block x F-filter net; "transgo" = the shipped MainNetwork (model.py:49-76)
        self.arch = _model.transgo_arch() if getattr(config, "network", "tower") == "transgo" else _model.tower_arch(self.blocks)
        self.engine = SelfPlayEngine(
            n_games, board_size=self.S, num_simulation=config.num_simulation,
            parallel_readouts=config.parallel_readouts, c_puct1=config.c_puct1, c_puct2=config.c_puct2,
            wu_loss=config.wu_loss, komi=config.komi, max_step=config.max_step,
            encode_dim=config.encode_state_channels, net_blocks=self.blocks, net_filters=self.filters,
            arena_slots=arena_slots, device=device, evaluator=evaluator,
            net_precision=getattr(config, "inference_dtype", "f32"))
        self.keep_obs = keep_obs
        self.seed_fn = seed_fn
        self.games_started = np.zeros(n_games, np.int64)
        self.records = [None] * n_games
        self.moves_played = 0
        self.games_finished = 0
        self._started = False
        self._hist, self._hist_base, self._move_idx = [], 0, 0
        self._game_start = np.zeros(n_games, np.int64)

    def seed_of(self, g):
        """Game seeds s = 1000*rank + g for the first game of a slot, then a fixed stride per restart."""
        if self.seed_fn is not None:
            return int(self.seed_fn(g, int(self.games_started[g]))) % (2 ** 32)
        return (1000 * self.rank + g + 1000003 * int(self.games_started[g]) * max(1, self.world)) % (2 ** 32)

    def set_weights(self, state_dict):
        _model.load_into(self.engine.ctx, state_dict, self.S, self.config.encode_state_channels, self.filters, arch=self.arch)

    def set_weights_blob(self, blob):
        import ctypes
        blob = np.ascontiguousarray(blob, np.float32)
        self.engine.ctx.call("tg_net_load_arch", self.arch.code.encode(), blob.ctypes.data_as(ctypes.c_void_p), blob.size, 0)

    def _reset(self, mask=None):
        idx = range(self.G) if mask is None else np.flatnonzero(mask)
        seeds = np.zeros(self.G, np.uint32)
        for g in idx:
            seeds[g] = self.seed_of(g)
            self.records[g] = GameRecord(int(seeds[g]))
            self.games_started[g] += 1
        self.engine.reset(seeds, mask)

    def start(self):
        self._reset(None)
        self._started = True

    def step(self, selfplay=True):
        """One move of every game: get_action_probs + update_with_action (self_play.py:917-926).  Returns the records
        of the games that ended with this move; their slots are restarted.  Per-move material is kept as whole-batch
        arrays (observations bit-packed) and only sliced per game when a game ends."""
        if not self._started:
            self.start()
        eng = self.engine
        eng.search(selfplay)
        vis, rn, players, steps, obs = eng.root_info(obs=self.keep_obs)
        actions, pis = eng.choose_moves(vis, steps, selfplay)
        packed = np.packbits(obs.reshape(self.G, -1).astype(np.uint8), axis=1) if self.keep_obs else None
        self._hist.append((packed, vis, players.astype(np.int8)))
        done = eng.play(actions)
        self.moves_played += self.G
        finished = []
        if done.any():
            score, terr, win = eng.final()
            C, S = self.config.encode_state_channels, self.S
            for g in np.flatnonzero(done):
                r = self.records[g]
                first = int(self._game_start[g]) - self._hist_base
                for pk, vi, pl in self._hist[first:]:
                    if pk is not None:
                        r.observations.append(np.unpackbits(pk[g])[:C * S * S].reshape(C, S, S).astype(np.float32))
                    counts = np.array([int(c) for c in vi[g]])
                    counts = np.where(counts == 1, 0, counts)                  # self_play.py:666-671
                    r.visits.append(vi[g].copy()); r.pis.append(counts / np.sum(counts)); r.players.append(int(pl[g]))
                r.winner, r.territory, r.score = int(win[g]), terr[g].copy(), float(score[g])
                finished.append(r)
            self.games_finished += len(finished)
            self._reset(done)
        self._move_idx += 1
        self._game_start[done] = self._move_idx
        drop = int(self._game_start.min()) - self._hist_base        # history older than every live game
        if drop > 0:
            del self._hist[:drop]
            self._hist_base += drop
        return finished

    def targets(self, record):
        return game_targets(record.observations, record.pis, record.players, record.winner, record.territory, self.S)


class SelfPlay:
    """Reference actor surface (self_play.py:881-983)."""

    def __init__(self, config, n_games=None, device=0, rank=0, world=1):
        self.config = config
        self.n_games = n_games or getattr(config, "concurrent_games", 1024)
        self.worker = BatchedSelfPlay(config, self.n_games, device=device, rank=rank, world=world)
        self._weights_version = None

    def _refresh_weights(self, shared_storage_worker):
        """self_play.py:913.  With several ranks, rank 0 asks the storage actor and every other rank receives the packed
        blob by one RCCL broadcast (transgo_amd.distributed.broadcast_weights)."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        w = None
        if not multi or dist.get_rank() == 0:
            w = _get(_call(shared_storage_worker.get_info, "weights"))
        if not multi:
            if w is not None and id(w) != self._weights_version:
                self.worker.set_weights(w)
                self._weights_version = id(w)
            return
        import torch
        from .distributed import broadcast_weights
        wk = self.worker
        dev = torch.device("cuda", wk.device) if dist.get_backend() == "nccl" else torch.device("cpu")
        changed = torch.tensor([int(w is not None and id(w) != self._weights_version)], device=dev)
        dist.broadcast(changed, src=0)
        if not int(changed.item()):
            return
        n = _model._lib.load().tg_net_blob_floats_arch(wk.S, wk.config.encode_state_channels, wk.filters, wk.arch.code.encode())
        blob = _model.pack_weights(w, wk.S, wk.config.encode_state_channels, wk.filters, arch=wk.arch) if w is not None \
            else np.zeros(n, np.float32)
        blob = broadcast_weights(blob, src=0, device=dev)
        wk.set_weights_blob(blob)
        self._weights_version = id(w) if w is not None else object()

    def policy_evaluate(self, n_games=10, shared_storage_worker=None, seed=0):
        """New-vs-old evaluation matches (self_play.py:986-1040): the train model ("weights") against the evaluation model
        ("evaluate_weights"), colours alternating game by game, every move by select_action (fresh tree, no noise,
        temperature 0.12).  All n_games run concurrently: two engines (one per weight set) each move half of the games per
        ply.  Returns (win_ratio, info2, info3) and promotes the weights on a clean sweep exactly as the reference does."""
        from .environment import GoEnv
        cfg, w = self.config, self.worker
        arch = w.arch
        half = (n_games + 1) // 2
        mk = lambda: SelfPlayEngine(
            half, board_size=cfg.board_size, num_simulation=cfg.num_simulation, parallel_readouts=cfg.parallel_readouts,
            c_puct1=cfg.c_puct1, c_puct2=cfg.c_puct2, wu_loss=cfg.wu_loss, komi=cfg.komi, max_step=cfg.max_step,
            encode_dim=cfg.encode_state_channels, net_blocks=w.blocks, net_filters=w.filters, device=w.device,
            net_precision=getattr(cfg, "inference_dtype", "f32"))
        eng = {"train": mk(), "eval": mk()}
        _model.load_into(eng["train"].ctx, _get(_call(shared_storage_worker.get_info, "weights")), cfg.board_size,
                         cfg.encode_state_channels, w.filters, arch=arch)
        _model.load_into(eng["eval"].ctx, _get(_call(shared_storage_worker.get_info, "evaluate_weights")), cfg.board_size,
                         cfg.encode_state_channels, w.filters, arch=arch)
        for k, e in enumerate(eng.values()):
            e.reset(np.arange(half) + seed + 7919 * k)            # seeds the per-game RNG streams
        env = GoEnv(cfg, device=w.device)
        # game i: train model plays BLACK when i is even (self_play.py:1000,1026)
        groups = {"A": np.arange(0, n_games, 2), "B": np.arange(1, n_games, 2)}
        states = {g: env.reset_batch(half) for g in groups}
        done = {g: np.arange(half) >= len(idx) for g, idx in groups.items()}
        ply = 0
        while not (done["A"].all() and done["B"].all()):
            black_is_train_group = "A" if ply % 2 == 0 else "B"      # whose train model is to move at this ply
            for who, grp in (("train", black_is_train_group), ("eval", "B" if black_is_train_group == "A" else "A")):
                live = ~done[grp]
                if not live.any():
                    continue
                acts = eng[who].select_action(states[grp], live)
                nxt, d, _ = env.step_batch(states[grp], np.where(live, acts, cfg.board_size ** 2))
                states[grp][live] = nxt[live]
                done[grp] |= d & live
            ply += 1
        win_num = 0
        info2 = None
        for grp, colour in (("A", 1), ("B", 2)):
            n = len(groups[grp])
            score = env.query_batch(states[grp][:n], score=True)["score"] if n else []
            for k in range(n):
                winner = 1 if score[k] > 0 else 2                     # environment.py:118-119
                win_num += int(winner == colour)
                info2 = "simulate round: {},  winer is : {},  model player is : {}\n".format(int(groups[grp][k]) + 1, winner, colour)
        lose_num = n_games - win_num
        evaluate_score = _get(_call(shared_storage_worker.get_info, "evaluate_score"))
        info3 = "evaluate_score:{}, win: {}, lose: {}\n".format(evaluate_score, win_num, lose_num)
        win_ratio = win_num / n_games
        if win_ratio == 1:                                             # self_play.py:1035-1038
            _call(shared_storage_worker.set_info, "evaluate_score", evaluate_score + 100)
            _call(shared_storage_worker.set_info, "evaluate_weights", _get(_call(shared_storage_worker.get_info, "weights")))
        for e in eng.values():
            e.close()
        return win_ratio, info2, info3

    def continuous_self_play(self, shared_storage_worker, mem, max_moves=None):
        moves = 0
        while max_moves is None or moves < max_moves:
            start = time.time()
            self._refresh_weights(shared_storage_worker)
            finished = self.worker.step()
            for _ in range(self.worker.G):
                _call(shared_storage_worker.set_info, "now_play_steps")           # self_play.py:928
            for rec in finished:
                for tup in self.worker.targets(rec):
                    _call(mem.append, *tup)                                       # self_play.py:956, :965
                _call(shared_storage_worker.set_info, "now_play_games")           # self_play.py:967
            moves += 1
            while (finished and                                                   # self_play.py:970-980
                   _get(_call(shared_storage_worker.get_info, "now_train_steps"))
                   / max(1, _get(_call(shared_storage_worker.get_info, "now_play_steps")))
                   < _get(_call(shared_storage_worker.get_info, "train_play_ratio"))
                   and _get(_call(shared_storage_worker.get_info, "adjust_train_play_ratio"))
                   and _get(_call(shared_storage_worker.get_info, "now_play_games"))
                   < _get(_call(shared_storage_worker.get_info, "game_total_num"))):
                time.sleep(0.5)
            if finished:
                print("run time:%.4fs" % (time.time() - start))
