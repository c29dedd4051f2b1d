"""Host-side pieces of the hot path that need no GPU: target generation + augmentation order against tuples recorded
from the reference's own continuous_self_play (tests/golden/targets_game.npz), weight packing, storage interfaces, and the
2-rank gather over gloo."""
import os
import socket

import numpy as np
import pytest

from transgo_amd import model
from transgo_amd.configure import Config
from transgo_amd.replay_buffer import ReplayMemory_Random
from transgo_amd.self_play import GameRecord, game_targets
from transgo_amd.shared_storage import SharedStorage


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name)) as z:
        return {k: z[k] for k in z.files}


def test_targets_match_reference_appends(golden_dir):
    b = _load(golden_dir, "targets_game.npz")
    n = len(b["z"]) // 8
    # un-augmented material = the 8th tuple of each position (rot90 k=4 is the identity)
    obs = [b["obs"][8 * i + 6] for i in range(n)]; pis = [b["pi"][8 * i + 6] for i in range(n)]
    players = [1 + (i % 2) for i in range(n)]
    z0 = b["z"][6]; winner = 1 if z0 == 1 else 2
    terr = b["own"][6]                                   # position 0 is black to move: own = territory
    out = game_targets(obs, pis, players, winner, terr, 9)
    assert len(out) == 8 * n
    for i, (o, p, z, w) in enumerate(out):
        assert o.dtype == np.float32 and o.shape == (10, 9, 9) and p.dtype == np.float64 and w.dtype == np.float64
        assert (o == b["obs"][i]).all() and (p == b["pi"][i]).all() and z == b["z"][i] and (w == b["own"][i]).all(), i


def test_targets_equal_oracle_restatement():
    from oracle.go_oracle import OracleGoEnv
    from oracle.wp_mcts import targets_for_game
    env = OracleGoEnv(max_step=16)
    rng = np.random.RandomState(3)
    s, done = env.reset()
    obs, pis, players = [], [], []
    while not done:
        la = env.getLegalAction(s)
        obs.append(env.encode(s)); players.append(env.getPlayer(s))
        c = rng.randint(0, 5, 82).astype(np.float64); c[la[0]] += 1; pis.append(c / c.sum())
        s, done = env.step(s, int(la[rng.randint(len(la))]))
    ref = targets_for_game(env, s, obs, pis, players)
    _, terr = env.getScoreAndTerritory(s)
    mine = game_targets(obs, pis, players, env.getWinner(s), terr, 9)
    assert len(ref) == len(mine) == 8 * len(obs)
    for a, b in zip(ref, mine):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_pack_weights_layout_roundtrip():
    sd = model.random_weights(9, 10, 32, 2, seed=3)
    blob = model.pack_weights(sd, 9, 10, 32, 2)
    assert blob.dtype == np.float32 and blob.size == 9 * 32 * 16 + 32 + 2 * (64 + 2 * (9 * 32 * 32 + 32)) + 64 + 9 * 16 * 32 + 16 \
        + 162 * 64 + 64 + 64 + 1 + 64 * 81 + 81 + 324 * 82 + 82
    # stem: folded BN, [tap][cout][cin16]
    s = sd["main_network.conv1.conv.1.weight"] / np.sqrt(sd["main_network.conv1.conv.1.running_var"] + 1e-5)
    w = sd["main_network.conv1.conv.0.weight"]
    assert np.allclose(blob[(4 * 32 + 5) * 16 + 3], w[5, 3, 1, 1] * s[5], rtol=1e-6)
    assert blob[(4 * 32 + 5) * 16 + 12] == 0


def test_storage_interfaces():
    cfg = Config(buffer_size=64)
    mem = ReplayMemory_Random(cfg)
    for i in range(70):
        mem.append(np.full((10, 9, 9), i, np.float32), np.full(82, 1 / 82), float(i % 2), np.zeros(81))
    assert mem.info() == {"capacity": 64, "index": 6, "full": True}
    batch = mem.sample(16)
    s, p, z, o = map(np.stack, zip(*batch))              # trainer.py:49
    assert s.shape == (16, 10, 9, 9) and p.shape == (16, 82) and z.shape == (16,) and o.shape == (16, 81)
    st = SharedStorage({"weights": None, "now_play_steps": 0, "now_play_games": 0, "learn_rate": 6.5e-5,
                        "adjust_lr": True, "train_play_ratio": 0.075, "adjust_train_play_ratio": True}, cfg)
    st.set_info("now_play_games")
    for _ in range(6):
        st.set_info("now_play_steps")
    assert st.get_info("now_play_steps") == 6 and st.get_info(["now_play_games"]) == {"now_play_games": 1}
    assert st.get_info("train_play_ratio") == (0.075 * 100000 + 1) / 100000     # configure.py:97-103
    st.set_info({"weights": 1}); st.set_info("learn_rate", 1e-3)
    assert st.get_info("weights") == 1 and st.get_info("learn_rate") == 1e-3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_games(rng, sizes, seed0=0):
    recs = []
    for k, n in enumerate(sizes):
        r = GameRecord(seed0 + k)
        for m in range(n):
            r.observations.append((rng.rand(10, 9, 9) < 0.3).astype(np.float32))
            v = rng.randint(0, 9, 82).astype(np.int32); v[5] = 7; v[3] = 1; r.visits.append(v)
            c = np.where(v == 1, 0, v); r.pis.append(c / c.sum()); r.players.append(1 + m % 2)
        r.winner = 1 + k % 2; r.territory = rng.randint(-1, 2, 81).astype(np.float32); r.score = float(k) - 0.5
        recs.append(r)
    return recs


def _summary(recs):
    return [(len(g.players), g.winner, g.seed, float(np.sum(g.territory)), float(sum(p.sum() for p in g.pis)),
             float(sum(o.sum() for o in g.observations)), int(sum(int(v.sum()) for v in g.visits))) for g in recs]


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    from transgo_amd import records
    from transgo_amd.distributed import broadcast_weights, gather_harvest
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.RandomState(rank)
    recs = _fake_games(rng, [3 + k for k in range(rank + 1)], seed0=100 * rank)     # rank 0: 1 game, rank 1: 2 games (ragged)
    h = records.from_records(recs, 9, 10)
    got = gather_harvest(h, 9, 10, dst=0)
    empty = gather_harvest(None, 9, 10, dst=0)             # nothing finished anywhere: no payload exchange
    only1 = gather_harvest(h if rank == 1 else None, 9, 10, dst=0)                  # the owner itself has nothing to add
    blob = np.arange(10, dtype=np.float32) * (rank + 1)
    blob = broadcast_weights(blob, src=0)
    out = [x for hb in got for x in _summary(hb.records())]
    out1 = [x for hb in only1 for x in _summary(hb.records())]
    nbytes = [hb.nbytes for hb in got]
    os.environ["TRANSGO_GATHER"] = "p2p"                   # the opt-in exact-length send/recv delivers the same batches
    alt = [x for hb in gather_harvest(h, 9, 10, dst=0) for x in _summary(hb.records())]
    os.environ.pop("TRANSGO_GATHER")
    assert alt == out
    q.put((rank, out, _summary(recs), len(empty), blob.tolist(), out1, nbytes, h.nbytes))
    dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    (r0, out0, mine0, e0, b0, o10, nb0, my0), (r1, out1, mine1, e1, b1, o11, nb1, my1) = res
    assert out1 == [] and e0 == 0 and e1 == 0 and o11 == []
    assert out0 == mine0 + mine1                          # rank order, exact payloads (pi recomputed from counts)
    assert o10 == mine1
    assert nb0 == [my0, my1] and my1 > my0                # every payload travels at its own length, not padded to the largest
    assert b0 == b1 == list(np.arange(10, dtype=np.float32))


def _gather8_worker(rank, world, port, q):
    """What one rank of an 8-rank job does per move, over gloo on the CPU: control word, size exchange, payloads (both transports),
    weight broadcast -- rank 3 and rank 6 finish nothing, the others a ragged number of games."""
    import torch.distributed as dist
    from transgo_amd import distributed, records
    distributed.init_process_group("gloo", rank, world, timeout_s=120.0, init_method=f"tcp://127.0.0.1:{port}")
    rng = np.random.RandomState(50 + rank)
    recs = [] if rank in (3, 6) else _fake_games(rng, [2 + (rank + k) % 4 for k in range(1 + rank % 3)], seed0=1000 * rank)
    h = records.from_records(recs, 9, 10) if recs else None
    ctl = distributed.control_exchange([rank == 0, 7 if rank == 0 else -1], src=0)
    res = {}
    for tr in ("allgather", "p2p"):
        os.environ["TRANSGO_GATHER"] = tr
        got, live = distributed.gather_harvest(h, 9, 10, dst=0, live=10 + rank)
        res[tr] = ([x for hb in got for x in _summary(hb.records())], live, [hb.nbytes for hb in got])
    os.environ.pop("TRANSGO_GATHER")
    assert distributed.transport_name().startswith("all_gather")            # the default when nothing is set
    blob = distributed.broadcast_weights(np.arange(64, dtype=np.float32) if rank == 0 else None, src=0, n_floats=64)
    q.put((rank, ctl, res, _summary(recs), h.nbytes if h else 0, float(np.asarray(blob).sum())))
    dist.destroy_process_group()


def test_gather_eight_ranks_gloo_both_transports():
    """BASELINE configs[2]/[4] are world 8: the per-move exchange with eight ranks (two of them with nothing to send), the padded
    all_gather (default) and the exact-length send/recv deliver the same batches in rank order; only rank 0 receives."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    ps = [ctx.Process(target=_gather8_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(60) for p in ps]
    assert [p.exitcode for p in ps] == [0] * world
    want = [x for r in res for x in r[3]]
    sizes = [r[4] for r in res if r[4]]
    for rank, ctl, got, _, _, bsum in res:
        assert ctl == [1, 7] and bsum == float(np.arange(64).sum())
        for tr in ("allgather", "p2p"):
            out, live, nbytes = got[tr]
            assert live == sum(10 + r for r in range(world))
            assert (out == want and nbytes == sizes) if rank == 0 else (out == [] and nbytes == [])


def test_harvest_batch_equals_per_game_targets(tmp_path):
    """records.Harvest.targets() (whole-batch rot90/flip) == game_targets (the literal per-game form, pinned above against the
    reference's own appends), tuple for tuple; records() is the inverse of from_records()."""
    from transgo_amd import records
    rng = np.random.RandomState(4)
    recs = _fake_games(rng, [4, 1, 7], seed0=9)
    h = records.from_records(recs, 9, 10)
    assert h.n_games == 3 and h.n_positions == 12 and not h.on_device
    back = h.records()
    assert _summary(back) == _summary(recs)
    for a, b in zip(back, recs):
        assert all(np.array_equal(x, y) for x, y in zip(a.observations, b.observations))
        assert all(np.array_equal(x, y) for x, y in zip(a.pis, b.pis)) and a.players == b.players
    want = [t for r in recs for t in game_targets(r.observations, r.pis, r.players, r.winner, r.territory, 9)]
    got = h.targets()
    assert len(got) == len(want) == 8 * 12
    for x, y in zip(got, want):
        assert x[0].dtype == y[0].dtype == np.float32 and x[1].dtype == y[1].dtype == np.float64 and x[3].dtype == y[3].dtype
        assert all(np.array_equal(p, q) for p, q in zip(x, y))


def test_default_game_seeds_never_collide():
    """with 1000*rank + g two ranks shared seeds as soon as a rank held more than 1000 boards."""
    from transgo_amd.self_play import default_seed
    world, G = 8, 4096
    seen = set()
    for k in range(3):
        for rank in range(world):
            s = {default_seed(rank, world, G, g, k) for g in range(G)}
            assert len(s) == G and not (s & seen)
            seen |= s
    assert default_seed(0, 8, 4096, 17, 0) == 17 and max(seen) < 2 ** 32
    with pytest.raises(OverflowError):
        default_seed(7, 8, 4096, 0, 2 ** 20)


def test_replay_buffer_pickled_layout_is_the_references(golden_dir):
    """ReplayMemory_Random.data is the reference's 2-D (capacity, 4) object array (replay_buffer.py:21-27), so save() / load()
    dicts are interchangeable with the reference's: the layout facts below were recorded from the reference class itself
    (tests/golden/gen_storage.py), including a part-filled buffer (blank rows) and a hand-built (N, 4) dict loaded back."""
    want = _load(golden_dir, "storage.npz")
    cfg = Config(buffer_size=50)
    mem = ReplayMemory_Random(cfg)
    assert mem.data.shape == (50, 4) and mem.data.dtype == object
    assert list(want["save_data_shape"]) == [50, 4] and list(want["partial_shape"]) == [10, 4] and list(want["hand_shape"]) == [3, 4]
    # a dict in the reference's layout loads; rows come back as the four fields
    tup = lambda i: (np.full((10, 9, 9), i, np.float32), np.full(82, 1.0 / 82), float(i), np.full(81, -float(i)))
    hand = {"buffer_capacity": 10, "index": 3, "full": False, "save_len": 3, "data": np.array([tup(20), tup(21), tup(22)], dtype=object)}
    assert hand["data"].shape == (3, 4)
    mem.load(hand)
    assert [float(r[2]) for r in mem.data[:3]] == list(want["hand_loaded_z"]) and mem.index == 3
    s, p, z, o = map(np.stack, zip(*mem.sample(2)))      # trainer.py:49 still stacks rows
    assert s.shape == (2, 10, 9, 9) and p.shape == (2, 82) and o.shape == (2, 81)


def test_add_info_equals_repeated_increments():
    """SharedStorage.add_info(key, n) == n times set_info(key) (shared_storage.py:21-43 with both schedules)."""
    rng = np.random.RandomState(1)
    for trial in range(30):
        base = {"now_play_steps": int(rng.randint(0, 20)), "now_play_games": int(rng.randint(0, 3)), "learn_rate": 6.5e-5,
                "adjust_lr": bool(trial % 3), "train_play_ratio": 0.075, "adjust_train_play_ratio": bool(trial % 2)}
        a, b = SharedStorage(base, Config()), SharedStorage(base, Config())
        for key in ("now_play_steps", "now_play_games", "now_play_steps"):
            n = int(rng.randint(0, 5000))
            for _ in range(n):
                a.set_info(key)
            b.add_info(key, n)
            assert a.current_checkpoint == b.current_checkpoint, (trial, key, n)


def test_vectorised_move_selection_equals_per_game_form():
    """transgo_amd.engine.choose_moves_batch == the literal self_play.py:666-683 computation, row by row, bit for bit."""
    from transgo_amd.engine import choose_moves_batch, choose_moves_reference
    rng = np.random.RandomState(0)
    G, A = 3000, 82
    visits = np.zeros((G, A), np.int32)
    for g in range(G):
        k = rng.randint(2, 40)
        a = rng.choice(A, k, replace=False)
        visits[g, a] = rng.multinomial(rng.randint(30, 900), rng.dirichlet([0.3] * k)) + rng.randint(0, 3, k)
        visits[g, a[0]] += 2
    steps = rng.randint(1, 121, G).astype(np.int32)
    u = rng.random_sample(G)
    u[:5] = [0.0, 1.0 - 2 ** -53, 0.5, 0.25, 0.75]
    live = rng.rand(G) < 0.9
    for selfplay in (True, False):
        a1, p1 = choose_moves_reference(visits, steps, u, live, selfplay)
        a2, p2 = choose_moves_batch(visits, steps, u, live, selfplay)
        assert np.array_equal(a1, a2) and np.array_equal(p1, p2)


def test_packed_replay_file_roundtrip(tmp_path):
    """Packed on-disk format -> loader -> the reference tuple layout, identical to feeding the buffer directly."""
    from transgo_amd.replay_buffer import load_packed, load_packed_into, save_packed
    rng = np.random.RandomState(9)
    recs = []
    for g in range(3):
        r = GameRecord(g)
        for m in range(5 + g):
            r.observations.append((rng.rand(10, 9, 9) < 0.3).astype(np.float32))
            v = rng.randint(0, 50, 82).astype(np.int32); v[3] = 1; v[7] += 2
            r.visits.append(v); c = np.where(v == 1, 0, v); r.pis.append(c / np.sum(c)); r.players.append(1 + m % 2)
        r.winner = 1 + g % 2; r.territory = rng.randint(-1, 2, 81).astype(np.float32)
        recs.append(r)
    path = str(tmp_path / "replay.tgrp")
    save_packed(path, recs, 9, 10)
    assert os.path.getsize(path) < 18 * 560 + 3 * 120 + 64           # ~0.52 KB per position + the per-game tables
    back = load_packed(path)
    assert len(back) == 3 and [len(r.players) for r in back] == [5, 6, 7]
    cfg = Config(buffer_size=1024)
    a, b = ReplayMemory_Random(cfg), ReplayMemory_Random(cfg)
    for r in recs:
        for t in game_targets(r.observations, r.pis, r.players, r.winner, r.territory, 9):
            a.append(*t)
    n = load_packed_into(path, b)
    assert n == a.index == b.index == 8 * 18
    for x, y in zip(a.data[:n], b.data[:n]):
        assert all(np.array_equal(p, q) for p, q in zip(x, y))


def test_pack_weights_matches_a_float64_reference_fold():
    """pack_weights writes straight into a float32 blob; the folded tensors must still be evaluated in float64 and rounded once
    (the fp16 oracle emulation and the HIP path both assume exactly that), and unfolded tensors must be copied bit-for-bit."""
    sd = model.random_weights(9, 10, 32, 2, seed=5)
    blob = model.pack_weights(sd, 9, 10, 32, 2)
    F = 32
    p = "main_network.res_blocks.0."
    s2 = sd[p + "batchnormlize_2.weight"].astype(np.float64) / np.sqrt(sd[p + "batchnormlize_2.running_var"].astype(np.float64) + 1e-5)
    w1 = (sd[p + "conv_1.weight"].astype(np.float64) * s2[:, None, None, None]).reshape(F, F, 9).transpose(2, 0, 1).astype(np.float32)
    o = 9 * F * 16 + F + 2 * F                                    # stem W, stem b, s1, t1
    assert np.array_equal(blob[o:o + 9 * F * F], w1.reshape(-1))
    o += 9 * F * F + F
    w2 = sd[p + "conv_2.weight"].reshape(F, F, 9).transpose(2, 0, 1)
    assert np.array_equal(blob[o:o + 9 * F * F], w2.reshape(-1))


def test_bench_rooflines_are_computed_from_measured_inputs():
    import bench
    assert bench.flops_per_leaf(9, 10, 128, 6) == pytest.approx(289.7e6, rel=1e-3)
    assert bench.flops_per_leaf(19, 10, 256, 20) == pytest.approx(17.06e9, rel=1e-3)
    t = bench.tree_roofline(9, 10, sims=1000, evals=1000, depth_sum=2000, children_scored=160000, tree_ms=10.0, waves=5)
    d, a = 2.0, 80.0
    per_eval = 26 * 4 + 83 * 4 + a * 32 + 2 * 11                 # 810 plane bits = 26 words
    want = d * (a + 1) * 32 + 2 * (d + 1) * 32 + 96 + per_eval
    assert t["bytes_per_sim"] == pytest.approx(want, abs=0.1) and t["mean_fanout"] == 80.0 and t["mean_depth"] == 2.0
    assert t["achieved"] == pytest.approx(1000 * want / 10e-3 / 1e9, abs=0.06) and t["bound"] == "hbm"      # reported to 0.1 GB/s
    assert bench.tree_roofline(9, 10, 0, 0, 0, 0, 0.0, 0) is None
    assert "k_conv3x3_sg<9,128>" in bench.kernel_name(9, 128, "f32") and "k_conv3x3_h2<19,256>" in bench.kernel_name(19, 256, "f16")
    assert bench.kernel_name(9, 64, "f32").startswith("k_conv3x3<9,64,64>")


def test_storage_classes_reproduce_the_reference_scenario(golden_dir):
    """tests/golden/storage.npz was recorded by running tests/golden/gen_storage.py::scenario on the reference's own
    ReplayMemory_Random / SharedStorage; the same function on this package's classes must give the same data."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_storage_scn", os.path.join(golden_dir, "gen_storage.py"))
    src = open(spec.origin).read().replace("import ref_harness  # noqa: E402", "")     # the scenario itself needs no reference
    ns = {"__name__": "gen_storage_scn", "__file__": spec.origin}
    exec(compile(src, spec.origin, "exec"), ns)
    got = ns["scenario"](ReplayMemory_Random, SharedStorage, Config())
    want = _load(golden_dir, "storage.npz")
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(np.asarray(got[k]), want[k]), k


def test_config_defaults_equal_the_reference(golden_dir):
    """Every scalar default this package's Config shares with the reference's (recorded by tests/golden/gen_config.py), and
    the temperature schedule of configure.py:75-79 at a few plies."""
    import json
    from transgo_amd.engine import temperature
    ref = json.load(open(os.path.join(golden_dir, "config_defaults.json")))
    mine = vars(Config())
    shared = [k for k in ref if k in mine]
    assert len(shared) >= 20
    for k in shared:
        assert mine[k] == ref[k], k
    for step, t in ref["temperature_at"].items():
        assert temperature(int(step)) == t


def test_weight_refresh_is_keyed_on_train_steps_and_content():
    """`id(weights)` is not a version.  The actor asks for now_train_steps first, fetches + packs the weights only when
    that counter moved, and uploads only when the packed content differs (SelfPlay._fetch_blob; no GPU needed: the engine is never
    touched here)."""
    import types
    from transgo_amd.self_play import SelfPlay
    sd1, sd2 = model.random_weights(9, 10, 32, 2, seed=1), model.random_weights(9, 10, 32, 2, seed=2)
    fetched = []

    class Storage:
        def __init__(self):
            self.d = {"weights": sd1, "now_train_steps": 5}

        def get_info(self, k):
            fetched.append(k)
            return self.d[k]
    st = Storage()
    sp = SelfPlay.__new__(SelfPlay)
    sp._train_steps_seen, sp._blob_digest = None, None
    sp.worker = types.SimpleNamespace(S=9, filters=32, config=Config(), arch=model.tower_arch(2))
    b = sp._fetch_blob(st)
    assert b is not None and np.array_equal(b, model.pack_weights(sd1, 9, 10, 32, 2))
    fetched.clear()
    assert sp._fetch_blob(st) is None and fetched == ["now_train_steps"]          # counter unchanged: the weights are not even fetched
    st.d["weights"] = dict(sd1)                                                   # a NEW dict object with the same content ...
    st.d["now_train_steps"] = 6                                                   # ... published by a later train step
    assert sp._fetch_blob(st) is None                                             # fetched, packed, same digest: no upload
    st.d["weights"] = sd2; st.d["now_train_steps"] = 7
    b2 = sp._fetch_blob(st)
    assert b2 is not None and np.array_equal(b2, model.pack_weights(sd2, 9, 10, 32, 2))
    st.d["weights"] = sd1                                                         # A -> B -> A again with a moving counter is an update too
    st.d["now_train_steps"] = 8
    assert sp._fetch_blob(st) is not None

    class Bare:                                                                   # a storage without the counter is asked every time
        def get_info(self, k):
            if k == "now_train_steps":
                raise KeyError(k)
            return sd2
    sp2 = SelfPlay.__new__(SelfPlay)
    sp2._train_steps_seen, sp2._blob_digest = None, None
    sp2.worker = sp.worker
    assert sp2._fetch_blob(Bare()) is not None and sp2._fetch_blob(Bare()) is None


def test_wp_mcts_mirror_has_the_reference_call_surface():
    """self_play.py:577, :595, :657, :689, :857, :874 -- names, argument names and defaults (construction needs a GPU; the behaviour is
    checked against the oracle in tests/test_gpu_search.py)."""
    import inspect
    from transgo_amd.self_play import WP_MCTS
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()][1:]
    assert sig(WP_MCTS.__init__)[:4] == [("config", inspect._empty), ("env", None), ("model", None), ("sub_model", None)]
    assert sig(WP_MCTS.reset_root) == []
    assert sig(WP_MCTS.get_action_probs) == [("is_selfplay", True), ("now_train_step", 0)]
    assert sig(WP_MCTS.select_action) == [("gamestate", inspect._empty)]
    assert sig(WP_MCTS.update_with_action) == [("fall_action", inspect._empty)]


# ---- the multi-rank ACTOR loop's control flow on CPU (gloo): fake engine, real SelfPlay.continuous_self_play -------------------
class _FakeWorker:
    """What SelfPlay.continuous_self_play uses of BatchedSelfPlay, without a GPU: every `period`-th move finishes one fake game."""

    def __init__(self, rank, G, period=2, die_at=None):
        self.rank, self.G, self.S, self.device, self.filters, self.blocks = rank, G, 9, 0, 32, 2
        self.config = Config(num_features=32, num_blocks=2)
        self.arch = model.tower_arch(2)
        self.moves, self.last_live, self.period, self.die_at = 0, 0, period, die_at
        self.blobs, self.games_finished = [], 0

    def advance(self, device=False):
        from transgo_amd import records
        self.moves += 1
        if self.die_at is not None and self.moves == self.die_at:
            os._exit(3)                                   # a rank that dies mid-run (no destroy_process_group, no goodbye)
        self.last_live = self.G - (1 if self.moves == 1 else 0)      # one parked slot on the first move
        if self.moves % self.period:
            return None
        self.games_finished += 1
        return records.from_records(_fake_games(np.random.RandomState(self.moves + 10 * self.rank), [2], seed0=self.moves), 9, 10)

    def set_weights_blob(self, blob, background=False):
        self.blobs.append(np.asarray(blob, np.float32).copy())


def _loop_worker(rank, world, port, q, scenario):
    import threading
    import time
    from transgo_amd import distributed
    from transgo_amd.self_play import SelfPlay
    distributed.init_process_group("gloo", rank, world, timeout_s=20.0, init_method=f"tcp://127.0.0.1:{port}")
    actor = SelfPlay.__new__(SelfPlay)
    actor.config = Config(num_features=32, num_blocks=2)
    actor.worker = _FakeWorker(rank, 3 + rank, die_at=2 if (scenario == "death" and rank == 1) else None)
    actor._train_steps_seen = actor._blob_digest = None
    st = mem = None
    if rank == 0:
        st = SharedStorage({"weights": model.random_weights(9, 10, 32, 2, seed=1), "now_play_steps": 0, "now_play_games": 0,
                            "now_train_steps": 0, "train_play_ratio": 0.5, "adjust_train_play_ratio": True,
                            "game_total_num": 1e8, "adjust_lr": False, "learn_rate": 1e-4}, actor.config)
        mem = ReplayMemory_Random(Config(buffer_size=4096))
        if scenario == "stall":
            def trainer():                                 # the trainer catches up (and publishes new weights) after 3 s
                time.sleep(3.0)
                st.set_info({"weights": model.random_weights(9, 10, 32, 2, seed=2), "now_train_steps": 10 ** 9})
            threading.Thread(target=trainer, daemon=True).start()
    t0 = time.time()
    actor.continuous_self_play(st, mem, max_moves=5)
    out = {"rank": rank, "seconds": time.time() - t0, "rounds": actor.throttle_rounds, "moves": actor.worker.moves,
           "blobs": [float(b.sum()) for b in actor.worker.blobs]}
    if rank == 0:
        out.update(steps=st.get_info("now_play_steps"), games=st.get_info("now_play_games"), entries=mem.info()["index"])
    q.put(out)
    import torch.distributed as dist
    dist.destroy_process_group()


def _run_loop(scenario):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_loop_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    [p.start() for p in ps]
    return ps, q


def test_actor_loop_stalled_trainer_two_ranks_gloo():
    """rank 0 used to sleep in the train/play throttle while rank 1 sat inside the next broadcast.  Now the
    wait is a collective decision (control word per round): with the trainer stalled for 3 s BOTH ranks idle on the host in
    0.5-s rounds, the same number of them, and both resume when now_train_steps moves -- with the weights the trainer published
    meanwhile delivered to every rank."""
    ps, q = _run_loop("stall")
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda d: d["rank"])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    r0, r1 = res
    assert r0["moves"] == r1["moves"] == 5
    assert r0["rounds"] == r1["rounds"] >= 4 and min(r0["seconds"], r1["seconds"]) >= 2.5
    want = [float(model.pack_weights(model.random_weights(9, 10, 32, 2, seed=s), 9, 10, 32, 2).sum()) for s in (1, 2)]
    assert r0["blobs"] == r1["blobs"] == want             # first version at move 1, second after the stall, on both ranks
    assert r0["games"] == 4 and r0["entries"] == 4 * 2 * 8                      # moves 2 and 4 finish one 2-ply game per rank
    assert r0["steps"] == 5 * 7 - 2                       # moves actually played: 3 + 4 slots, one parked each on the first move


def test_actor_loop_peer_death_is_a_nonzero_exit_not_a_hang():
    """A rank that dies leaves the others inside a collective: with the explicit process-group timeout (20 s here) the survivor
    ends with an error instead of waiting forever."""
    import time
    ps, q = _run_loop("death")
    t0 = time.time()
    [p.join(90) for p in ps]
    assert all(not p.is_alive() for p in ps), "a rank is still hanging"
    assert ps[1].exitcode == 3 and ps[0].exitcode not in (0, None)
    assert time.time() - t0 < 80


def test_bench_launcher_world8_rehearsal_with_the_stand_in_engine():
    """BASELINE configs[2]/[4] are world 8, and this pool cannot run eight GPU processes on one card: the whole N-rank control flow of
    `bench.py --gpus 8` -- the launcher's eight fresh ranks, the gloo group, the per-move finished-game exchange (default transport),
    the reductions, per-rank figures, ONE line -- runs on the CPU with tests/bench_standin.py in place of the HIP engine."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TRANSGO_GATHER")}
    env.update(TRANSGO_BENCH_STANDIN="1", TRANSGO_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + "--gpus 8 --games 64 --steps 3 --warmup 1 --no-cpu-baseline".split(),
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert "NOT a measurement" in line["data"] and "REHEARSAL" in line["metric"] and line["n_gpus"] == 8 and line["scaling"] == "weak"
    rk = line["ranks"]
    assert rk["world"] == 8 and rk["backend"] == "gloo" and rk["launcher"] == "bench.py" and rk["seeds_disjoint"] is True
    assert len({d["pid"] for d in rk["devices"]}) == 8 and sorted(d["rank"] for d in rk["devices"]) == list(range(8))
    assert rk["transport"].startswith("all_gather")
    pr = rk["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8)) and all(p["games_finished"] == 3 * (1 + p["rank"] % 3) for p in pr)
    sg = line["selfplay_games"]
    assert sg["finished_and_stored"] == sg["finished_all_ranks"] == sum(p["games_finished"] for p in pr)     # every rank's games reached rank 0
    assert sg["positions_stored"] == sum(p["games_finished"] * (2 + p["rank"] % 4) for p in pr)
    assert abs(sum(p["sims"] for p in pr) / (line["ms_per_step"] * line["steps"] * 1e-3) - line["value"]) <= 1e-3 * line["value"]
    # the same job over the opt-in exact-length send/recv stores the same games
    env["TRANSGO_GATHER"] = "p2p"
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + "--gpus 8 --games 64 --steps 3 --warmup 1 --no-cpu-baseline".split(),
                        cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    l2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert "isend/irecv" in l2["ranks"]["transport"] and l2["selfplay_games"]["positions_stored"] == sg["positions_stored"]
