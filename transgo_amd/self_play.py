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


def default_seed(rank, world, n_games, g, k):
    """Seed of the k-th game played in slot g of rank `rank`: distinct for every (rank, slot, restart) by construction --
    consecutive blocks of n_games seeds per (restart, rank) -- so no two games of a job share an RNG stream.  First games of
    rank 0 get s_g = g."""
    s = (k * max(1, world) + rank) * n_games + g
    if s >= 2 ** 32:
        raise OverflowError("game seed space exhausted (2**32 games)")
    return s


class BatchedSelfPlay:
    """G concurrent self-play games in lock step (one engine, one GPU).  The per-game material the reference keeps in Python
    lists (observations, visit counts, players; self_play.py:917-926) lives in HBM inside the engine; a finished game leaves
    it as a `records.Harvest` batch -- in device memory when the consumer is on the GPU too."""

    def __init__(self, config, n_games, device=0, rank=0, world=1, evaluator=None, arena_slots=0, seed_fn=None, pool_slots=0):
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
            pool_slots=pool_slots or getattr(config, "tree_pool_slots", 0),
            net_precision=getattr(config, "inference_dtype", "f32"))
        self.seed_fn = seed_fn
        self.games_started = np.zeros(n_games, np.int64)
        self.seeds = np.zeros(n_games, np.uint32)          # seed of the game now running in each slot
        self.moves_played = 0
        self.last_live = 0
        self.games_finished = 0
        self.games_dropped = 0                             # games abandoned because their tree outgrew the arena
        self.phase_s = {}
        self._started = False
        self._parked = None                                # slots waiting for their staggered first move (start(stagger))

    def seed_of(self, g):
        k = int(self.games_started[g])
        if self.seed_fn is not None:
            return int(self.seed_fn(g, k)) % (2 ** 32)
        return default_seed(self.rank, self.world, self.G, g, k)

    def set_weights(self, state_dict):
        """state_dict of the configured layout; a MainNetwork state_dict (the reference trainer's, model.py:49-76) switches a
        default-configured worker to that architecture instead of failing on the key names."""
        if "main_network.res_conv2.conv_1.weight" in state_dict and self.arch.code != _model.transgo_arch().code:
            self.arch = _model.transgo_arch()
        elif "main_network.res_blocks.0.conv_1.weight" in state_dict and self.arch.policy_attention:
            self.arch = _model.tower_arch(self.blocks)
        self.weight_range = _model.load_into(self.engine.ctx, state_dict, self.S, self.config.encode_state_channels, self.filters,
                                             arch=self.arch)
        return self.weight_range

    def set_weights_blob(self, blob, background=False):
        """Packed blob -> GPU.  background=True: upload into the idle weight set on a side stream (tg_net_load_async) while
        searches continue on the live one; the switch happens at the next move boundary (tg_sp_begin_move) after the upload
        completed, so one move's search never mixes two weight sets.  A float32 torch tensor in this GPU's memory (what an RCCL
        broadcast delivered) is taken device -> device (tg_net_load_async_dev)."""
        import ctypes
        if hasattr(blob, "is_cuda") and blob.is_cuda:
            assert blob.dtype.is_floating_point and blob.element_size() == 4 and blob.is_contiguous()
            self.engine.ctx.call("tg_net_load_async_dev", self.arch.code.encode(), ctypes.c_void_p(blob.data_ptr()), blob.numel())
            return
        blob = np.ascontiguousarray(blob, np.float32)
        if background:
            self.engine.ctx.call("tg_net_load_async", self.arch.code.encode(), blob.ctypes.data_as(ctypes.c_void_p), blob.size)
        else:
            self.engine.ctx.call("tg_net_load_arch", self.arch.code.encode(), blob.ctypes.data_as(ctypes.c_void_p), blob.size, 0)

    def _reset(self, mask=None):
        idx = range(self.G) if mask is None else np.flatnonzero(mask)
        for g in idx:
            self.seeds[g] = self.seed_of(g)
            self.games_started[g] += 1
        self.engine.reset(self.seeds, mask)

    def start(self, stagger=0):
        """stagger = T > 1: slot g plays its first move at step g mod T and is parked until then, so that games end spread over T
        consecutive steps instead of all on the same one (with a ply limit every game of a generation has the same length) -- a
        steady flow of finished games to the replay store from the first generation on, paid with a partly idle engine during the
        first T steps.  No position is played differently: a delayed game is the same game.  (The reference's actors drift apart by
        themselves, one game per process.)"""
        T = int(stagger or 0)
        if T > 1 and self.G > 1:
            self._start_at = np.arange(self.G) % T
            self._step_no = 0
            self._parked = np.ones(self.G, bool)
            eng = self.engine
            eng.reset(np.zeros(self.G, np.uint32))                      # brings the engine up (reset_from needs seeded streams) ...
            eng.reset_from(np.zeros((self.G, eng.ctx.state_size), np.uint8), np.zeros(self.G, bool))     # ... then every slot is parked
        else:
            self._reset(None)
        self._started = True

    def _begin_due(self):
        due = self._parked & (self._start_at == self._step_no)
        if due.any():
            self._reset(due)
            self._parked &= ~due
        self._step_no += 1

    def advance(self, selfplay=True, device=False, num_simulation=0):
        """One move of every game: get_action_probs + update_with_action (self_play.py:917-926).  Returns the games this
        move finished as a `records.Harvest` (None if none did); their slots are restarted with fresh seeds, and so are slots
        whose tree outgrew its arena (counted in games_dropped)."""
        if not self._started:
            self.start(getattr(self.config, "stagger_games", 0))
        if self._parked is not None and self._parked.any():
            self._begin_due()
        eng = self.engine
        live = int((~eng.finished).sum())
        t0 = time.perf_counter()
        eng.search(selfplay, num_simulation)
        t1 = time.perf_counter()
        vis, steps = eng.root_visits()
        actions, _ = eng.choose_moves(vis, steps, selfplay)
        t2 = time.perf_counter()
        done = eng.play(actions)
        t3 = time.perf_counter()
        self.moves_played += live
        self.last_live = live                                           # slots that really played this move (not parked / over / in error)
        h = None
        if self._parked is not None:
            done = done & ~self._parked                                 # a parked slot reports "over" without having played
        if done.any():
            h = eng.harvest(device=device, seeds=self.seeds)
            self.games_finished += h.n_games
        restart = done | eng.errored
        if restart.any():
            self.games_dropped += int(eng.errored.sum())
            self._reset(restart)
        t4 = time.perf_counter()
        for k, v in (("search", t1 - t0), ("select", t2 - t1), ("play", t3 - t2), ("game_end", t4 - t3)):
            self.phase_s[k] = self.phase_s.get(k, 0.0) + v                        # wall-clock shares of a step (bench.py reports them)
        return h

    def step(self, selfplay=True):
        """advance() with the finished games unpacked into GameRecord objects (host lists, as the reference keeps them)."""
        h = self.advance(selfplay)
        return h.records() if h is not None else []

    def targets(self, record):
        return game_targets(record.observations, record.pis, record.players, record.winner, record.territory, self.S)


class GroupedSelfPlay:
    """The same G concurrent games as K independent groups: every group is a BatchedSelfPlay of G/K games with its own context and
    HIP stream, and K host threads advance them together.  The GPU then runs the groups' kernels concurrently: while one group's
    conv launch drains its partial last round, or its tree stage / stem / heads run (latency- or HBM-bound, a few workgroups'
    worth), the other groups' MFMA work fills the chip -- separate streams instead of one serial chain (+2.5 % at C2 with four
    groups, +2.9 % for the split-precision network with two).  No game changes: slot g of group k is slot k*G/K + g of the job (same
    seeds as the ungrouped engine), and a network row does not depend on which rows share its batch
    (tests/test_gpu_selfplay.py::test_grouped_selfplay_plays_the_same_games)."""

    def __init__(self, config, n_games, groups=2, device=0, rank=0, world=1, arena_slots=0, pool_slots=0):
        assert groups >= 1 and n_games % groups == 0, "games must divide evenly over the groups"
        self.config, self.G, self.K, self.rank, self.world = config, n_games, groups, rank, world
        per = n_games // groups
        self.parts = [BatchedSelfPlay(config, per, device=device, rank=rank, world=world, arena_slots=arena_slots, pool_slots=pool_slots,
                                      seed_fn=(lambda g, r, k=k: default_seed(rank, world, n_games, k * per + g, r)))
                      for k in range(groups)]
        self.S, self.device = config.board_size, device
        self._pool = None

    def _each(self, fn):
        """fn(part) for every group on its own host thread (the library releases the GIL inside its calls).  The K threads are a
        persistent pool (one worker per group, created once: a move is one submit per group, not K thread starts).  A group that
        fails must not leave the job half alive: in a multi-rank actor loop the other ranks would sit in the next collective until
        the process-group timeout, so once every group's call has returned, a failure aborts the process group (peers then fail
        fast instead of waiting) before the first exception is re-raised."""
        if self.K == 1:
            return [fn(self.parts[0])]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.K, thread_name_prefix="transgo-group")
        futs = [self._pool.submit(fn, p) for p in self.parts]
        out, first = [], None
        for f in futs:
            try:
                out.append(f.result())
            except BaseException as e:          # re-raised in the caller's thread, after every group has returned
                out.append(None)
                first = first or e
        if first is not None:
            self._abort_peers()
            raise first
        return out

    @staticmethod
    def _abort_peers():
        """A failed group on this rank: tear the process group down so that peers blocked in a collective error out now."""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.destroy_process_group()
        except Exception:
            pass

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def set_weights(self, state_dict):
        return self._each(lambda p: p.set_weights(state_dict))[0]

    def set_weights_blob(self, blob, background=False):
        # the K uploads go out together, one per group thread (each device -> device load ends with a host-blocking stream wait:
        # issued one after the other they would serialise K waits)
        self._each(lambda p: p.set_weights_blob(blob, background))

    # what the actor loop reads off its worker (SelfPlay.continuous_self_play)
    @property
    def arch(self):
        return self.parts[0].arch

    @arch.setter
    def arch(self, a):
        for p in self.parts:
            p.arch = a

    @property
    def filters(self):
        return self.parts[0].filters

    @property
    def blocks(self):
        return self.parts[0].blocks

    @property
    def last_live(self):
        return sum(p.last_live for p in self.parts)

    def start(self, stagger=0):
        self._each(lambda p: p.start(stagger))

    def advance(self, selfplay=True, device=False, num_simulation=0):
        """One move of every game of every group; returns the groups' finished-game batches (records.Harvest or None), in group order."""
        return self._each(lambda p: p.advance(selfplay, device, num_simulation))

    @property
    def games_finished(self):
        return sum(p.games_finished for p in self.parts)

    @property
    def games_dropped(self):
        return sum(p.games_dropped for p in self.parts)

    def stats(self):
        tot = {}
        for p in self.parts:
            for k, v in p.engine.stats().items():
                tot[k] = max(tot.get(k, 0), v) if k == "max_slots" else tot.get(k, 0) + v
        return tot


class _Root:
    """What scripts read off `mcts.root` in the reference (self_play.py:596-605): the position and the children's visit counts."""

    def __init__(self, owner):
        self._o = owner

    @property
    def state(self):
        return self._o.engine.root_states()[0]

    @property
    def visit_counts(self):
        return self._o.engine.root_visits()[0][0]

    def visit_count(self, action):                     # Node_V.visit_count, self_play.py:79-82
        return int(self.visit_counts[action])


class WP_MCTS:
    """The reference's per-game search object (self_play.py:575-881) over ONE slot of the batched engine, for code that drives a
    single tree the way the reference does -- `get_action_probs()` / `update_with_action(a)` in a game loop, `select_action(state)`
    against another player -- and does not want to batch games itself (throughput lives in BatchedSelfPlay / SelfPlay).
    Same calls, same order => the actions, pis and root observations of the reference run after `np.random.seed(seed)`: the
    reference draws from NumPy's global stream, this object owns one MT19937 stream seeded with `seed` (tests/test_gpu_search.py
    checks both entry points against the oracle).  `model`: a state_dict, or anything with `get_weights()` (TransGoNetwork's
    method, model.py:23-24); `evaluator` replaces the network by a host callable obs -> (policy, value) (parity tests)."""

    def __init__(self, config, env=None, model=None, sub_model=None, seed=0, device=0, evaluator=None):
        self.config, self.env, self.model, self.sub_model = config, env, model, sub_model
        self.board_size, self.komi = config.board_size, config.komi
        self.parallel_readouts, self.num_simulations = config.parallel_readouts, config.num_simulation
        self.c1, self.c2, self.wu_loss = config.c_puct1, config.c_puct2, config.wu_loss
        self.seed = int(seed) % (2 ** 32)
        self._sp = BatchedSelfPlay(config, 1, device=device, evaluator=evaluator, seed_fn=lambda g, k: self.seed)
        self.engine = self._sp.engine
        if evaluator is None:
            if model is None:
                raise ValueError("WP_MCTS needs a model (state_dict or object with get_weights()) or an evaluator")
            self._sp.set_weights(model.get_weights() if hasattr(model, "get_weights") else model)
        self.root = _Root(self)
        self.engine.reset(np.array([self.seed], np.uint32))       # np.random.seed(seed), ONCE + reset_root (self_play.py:592-593)

    def reset_root(self):
        """self_play.py:595-605: empty board, root expanded with raw priors.  The random stream is NOT touched -- the reference
        draws from NumPy's global stream, which keeps advancing from one game to the next (continuous_self_play calls reset_root
        at the top of every game): a second game draws different Dirichlet noise and moves than the first."""
        streams = self.engine.rng_streams()
        self.engine.reset(np.array([self.seed], np.uint32))       # tg_sp_reset re-seeds the slot's stream ...
        self.engine.set_rng_streams(streams)                      # ... which continues where the last game left it

    def get_action_probs(self, is_selfplay=True, now_train_step=0):
        """self_play.py:657-687 -> (action, pi, encode(root)): root noise if is_selfplay, num_simulation more visits at the root,
        counts == 1 -> 0, temperature schedule (0.12 when not self-play), np.random.choice."""
        self.engine.search(selfplay=bool(is_selfplay))
        vis, _, _, steps, obs = self.engine.root_info()
        actions, pis = self.engine.choose_moves(vis, steps, selfplay=bool(is_selfplay))
        return int(actions[0]), pis[0], obs[0]

    def select_action(self, gamestate):                # self_play.py:689-703: fresh tree at `gamestate`, no noise, tau 0.12
        st = np.frombuffer(bytes(gamestate), np.uint8) if not isinstance(gamestate, np.ndarray) else gamestate.view(np.uint8)
        return int(self.engine.select_action(st.reshape(1, -1))[0])

    def update_with_action(self, fall_action):         # self_play.py:857-872: re-root on the child, keep its sub-tree
        """Returns True when the move ended the game (the reference returns nothing; its caller steps the env itself)."""
        return bool(self.engine.play(np.array([int(fall_action)], np.int32))[0])

    def close(self):
        self.engine.close()

    def __str__(self):                                 # self_play.py:874-875
        return "WP_MCTS"


def _bump(storage, key, n):
    """n times the increment form set_info(key) (shared_storage.py:27-28 plus its schedules), as ONE call when the storage
    object offers add_info (transgo_amd.shared_storage does; the result is identical), else n calls."""
    if n <= 0:
        return
    if hasattr(storage, "add_info"):
        _call(storage.add_info, key, int(n))
    else:
        for _ in range(int(n)):
            _call(storage.set_info, key)


class SelfPlay:
    """Reference actor surface (self_play.py:881-983).  With torch.distributed initialised every rank runs one of these over
    its own shard of the games; rank 0 is the one that talks to the storage / replay actors (the others pass None)."""

    def __init__(self, config, n_games=None, device=0, rank=0, world=1, evaluator=None):
        self.config = config
        self.n_games = n_games or getattr(config, "concurrent_games", 1024)
        groups = int(getattr(config, "game_groups", 1) or 1)
        if groups > 1 and evaluator is None:
            # the boards as independent groups on their own HIP streams (GroupedSelfPlay): kernels of different groups overlap
            self.worker = GroupedSelfPlay(config, self.n_games, groups=groups, device=device, rank=rank, world=world)
        else:
            self.worker = BatchedSelfPlay(config, self.n_games, device=device, rank=rank, world=world, evaluator=evaluator)
        self._train_steps_seen = None      # storage's now_train_steps when the weights were last fetched
        self._blob_digest = None           # content digest of the packed weights now on the GPU

    @staticmethod
    def _dist():
        import torch.distributed as dist
        from .distributed import _active
        multi = _active()
        return dist, multi, (dist.get_rank() if multi else 0), (dist.get_world_size() if multi else 1)

    def _fetch_blob(self, shared_storage_worker):
        """Rank-0 side of the weight refresh (self_play.py:913): ask the storage for its train-step counter first and fetch
        + pack the weights only when it moved (the trainer publishes weights inside train steps only, trainer.py:74-89, so
        the counter is a monotonic version); what then decides an upload is the content digest of the packed blob.  A storage
        without that counter is asked for its weights every time.  Returns the new blob or None."""
        import hashlib
        wk = self.worker
        try:
            steps = _get(_call(shared_storage_worker.get_info, "now_train_steps"))
        except KeyError:
            steps = None
        if steps is not None and steps == self._train_steps_seen and self._blob_digest is not None:
            return None
        self._train_steps_seen = steps
        w = _get(_call(shared_storage_worker.get_info, "weights"))
        if w is None:
            return None
        if "main_network.res_conv2.conv_1.weight" in w and not wk.arch.policy_attention:
            wk.arch = _model.transgo_arch()
        blob = _model.pack_weights(w, wk.S, wk.config.encode_state_channels, wk.filters, arch=wk.arch)
        digest = hashlib.blake2b(blob.tobytes(), digest_size=16).digest()
        if digest == self._blob_digest:
            return None
        self._blob_digest = digest
        return blob

    def _throttled(self, st):
        """The reference's train/play throttle (self_play.py:970-980), asked of the storage once."""
        return bool(_get(_call(st.get_info, "now_train_steps")) / max(1, _get(_call(st.get_info, "now_play_steps")))
                    < _get(_call(st.get_info, "train_play_ratio"))
                    and _get(_call(st.get_info, "adjust_train_play_ratio"))
                    and _get(_call(st.get_info, "now_play_games")) < _get(_call(st.get_info, "game_total_num")))

    def _move_prologue(self, shared_storage_worker, throttle):
        """Top of a move for EVERY rank: the throttle wait of self_play.py:970-980 and the weight refresh of :913, decided by
        rank 0 (the rank that talks to the storage) and shared through ONE small broadcast per round (distributed.
        control_exchange): word 0 = "the trainer is behind, everybody sleeps 0.5 s and asks again", word 1 = "a new packed
        blob follows" (2: of the MainNetwork layout).  While rank 0 waits for the trainer every rank sleeps on the host between
        two short collectives -- nobody sits inside a pending broadcast for as long as the trainer is slow (over RCCL that is
        a watchdog abort once the process-group timeout passes)."""
        from .distributed import broadcast_weights, control_exchange
        dist, multi, rank, _ = self._dist()
        wk = self.worker
        waited = 0
        while True:
            wait = bool(rank == 0 and throttle and self._throttled(shared_storage_worker))
            blob = None
            if rank == 0 and not wait:
                blob = self._fetch_blob(shared_storage_worker)
            flag = 0 if blob is None else (2 if wk.arch.policy_attention else 1)
            if multi:
                wait, flag = control_exchange([int(wait), flag], src=0, device_index=wk.device)
            if not wait:
                break
            waited += 1
            time.sleep(0.5)
        self.throttle_rounds = getattr(self, "throttle_rounds", 0) + waited
        if not flag:
            return
        if not multi:
            wk.set_weights_blob(blob, background=True)
            return
        import torch
        dev = torch.device("cuda", wk.device) if dist.get_backend() == "nccl" else torch.device("cpu")
        n = None
        if rank != 0:
            if flag == 2 and not wk.arch.policy_attention:
                wk.arch = _model.transgo_arch()
            n = _model._lib.load().tg_net_blob_floats_arch(wk.S, wk.config.encode_state_channels, wk.filters, wk.arch.code.encode())
        got = broadcast_weights(blob, src=0, device=dev, n_floats=n)
        # over RCCL every rank (the sender included) loads what the broadcast left in its GPU memory, device -> device
        # (tg_net_load_async_dev: no host bounce); over gloo the NumPy blob goes through the pinned staging buffer
        wk.set_weights_blob(got if hasattr(got, "is_cuda") else (blob if rank == 0 else got), background=True)

    def _refresh_weights(self, shared_storage_worker):
        """self_play.py:913 alone (no throttle): see _move_prologue."""
        self._move_prologue(shared_storage_worker, throttle=False)

    def policy_evaluate(self, n_games=10, shared_storage_worker=None, seed=0, evaluators=None):
        """New-vs-old evaluation matches (self_play.py:986-1040): the train model ("weights") against the evaluation model
        ("evaluate_weights"), colours alternating game by game, every move by select_action (fresh tree, no noise,
        temperature 0.12).  All n_games run concurrently: two engines (one per weight set) each move half of the games per
        ply.  In the reference both agents of a game draw from the one global NumPy stream; here game i owns a stream seeded
        `seed + i` that travels with the game between the two engines, so a game equals the reference's game i played after
        np.random.seed(seed + i) (tests/test_gpu_search.py checks that against the oracle).  Returns (win_ratio, info2, info3)
        and promotes the weights on a clean sweep exactly as the reference does.  `evaluators` = {"train": fn, "eval": fn}
        replaces the two networks by host evaluators (parity tests)."""
        from . import _lib
        from .environment import GoEnv
        cfg, w = self.config, self.worker
        arch = w.arch
        half = (n_games + 1) // 2
        mk = lambda who: SelfPlayEngine(
            half, board_size=cfg.board_size, num_simulation=cfg.num_simulation, parallel_readouts=cfg.parallel_readouts,
            c_puct1=cfg.c_puct1, c_puct2=cfg.c_puct2, wu_loss=cfg.wu_loss, komi=cfg.komi, max_step=cfg.max_step,
            encode_dim=cfg.encode_state_channels, net_blocks=w.blocks, net_filters=w.filters, device=w.device,
            net_precision=getattr(cfg, "inference_dtype", "f32"), evaluator=evaluators[who] if evaluators else None,
            record_games=False)
        eng = {"train": mk("train"), "eval": mk("eval")}
        if not evaluators:
            for who, key in (("train", "weights"), ("eval", "evaluate_weights")):
                _model.load_into(eng[who].ctx, _get(_call(shared_storage_worker.get_info, key)), cfg.board_size,
                                 cfg.encode_state_channels, w.filters, arch=arch)
        for e in eng.values():
            e.reset(np.zeros(half, np.uint32))                     # brings the engine up; the streams that count are set per ply
        env = GoEnv(cfg, device=w.device)
        # game i: train model plays BLACK when i is even (self_play.py:1000,1026); group A = even games, B = odd games
        groups = {"A": np.arange(0, n_games, 2), "B": np.arange(1, n_games, 2)}
        states = {g: env.reset_batch(half) for g in groups}
        done = {g: np.arange(half) >= len(idx) for g, idx in groups.items()}
        streams = {}
        lib = _lib.load()
        for g, idx in groups.items():                              # np.random.seed(seed + i) for game i
            st = (_lib.TgMt19937 * half)()
            for k, i in enumerate(idx):
                lib.tg_host_mt_seed(st[k], int(seed + i) % (2 ** 32))
            streams[g] = np.frombuffer(st, np.uint8).reshape(half, -1).copy()
        ply = 0
        while not (done["A"].all() and done["B"].all()):
            black_is_train_group = "A" if ply % 2 == 0 else "B"      # whose train model is to move at this ply
            for who, grp in (("train", black_is_train_group), ("eval", "B" if black_is_train_group == "A" else "A")):
                live = ~done[grp]
                if not live.any():
                    continue
                eng[who].set_rng_streams(streams[grp])
                acts = eng[who].select_action(states[grp], live)
                streams[grp] = eng[who].rng_streams()
                nxt, d, _ = env.step_batch(states[grp], np.where(live, acts, cfg.board_size ** 2))
                states[grp][live] = nxt[live]
                done[grp] |= d & live
            ply += 1
        winners = np.zeros(n_games, np.int64)
        colours = np.where(np.arange(n_games) % 2 == 0, 1, 2)         # the train model's colour in game i
        for grp in groups:
            n = len(groups[grp])
            if n:
                score = env.query_batch(states[grp][:n], score=True)["score"]
                winners[groups[grp]] = np.where(score > 0, 1, 2)      # environment.py:118-119
        win_num = int((winners == colours).sum())
        lose_num = n_games - win_num
        self.last_evaluation = {"winners": winners, "colours": colours}
        info2 = "simulate round: {},  winer is : {},  model player is : {}\n".format(n_games, int(winners[-1]), int(colours[-1])) \
            if n_games else None                                      # the reference keeps the last game's line only
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
        """self_play.py:902-983 for G games per rank at once.  Finished games are gathered to rank 0
        (transgo_amd.distributed.gather_harvest: the RCCL exchange that replaces the per-tuple Ray RPCs of :956,:965) and only
        rank 0 appends.  `mem` is either a device store (DeviceReplayMemory: the batch goes HBM -> HBM) or anything with the
        reference's append(obs, pi, z, own) (replay_buffer.py:30-34), which receives the reference's 8 tuples per move in
        the reference's order.  Counters: now_play_steps += 1 per move actually played (parked, finished and errored slots do
        not count; summed over the ranks inside the gather's size exchange), now_play_games += 1 per finished game.
        Collectives per move, all short and all entered by every rank together: the control word (_move_prologue), the
        weight blob when one follows, the (games, positions, live) exchange and the payloads of gather_harvest."""
        from .distributed import gather_harvest
        dist, multi, rank, world = self._dist()
        owner = rank == 0
        device_mem = hasattr(mem, "append_harvest")
        on_gpu = device_mem or (multi and dist.get_backend() == "nccl")
        wk = self.worker
        moves = 0
        throttle = False                                  # the reference checks its ratio after a finished game only (:968-980)
        while max_moves is None or moves < max_moves:
            start = time.time()
            self._move_prologue(shared_storage_worker, throttle)
            hs = wk.advance(device=on_gpu)
            batches, live = [], 0
            for k, h in enumerate(hs if isinstance(hs, list) else [hs]):     # one gather per group: the same number on every rank
                b, lv = gather_harvest(h, wk.S, wk.config.encode_state_channels, dst=0, device_index=wk.device,
                                       live=wk.last_live if k == 0 else 0)
                batches += b; live += lv
            moves += 1
            throttle = False
            if not owner:
                continue
            _bump(shared_storage_worker, "now_play_steps", live)                      # self_play.py:928
            finished = 0
            for hb in batches:
                if device_mem:
                    mem.append_harvest(hb)
                else:
                    for tup in hb.targets():
                        _call(mem.append, *tup)                                       # self_play.py:956, :965
                finished += hb.n_games
            _bump(shared_storage_worker, "now_play_games", finished)                  # self_play.py:967
            throttle = finished > 0                       # waited for at the top of the next move, by all ranks together
            if finished:
                print("run time:%.4fs" % (time.time() - start))
