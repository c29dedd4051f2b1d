"""Whole hot path on the GPU (search + move selection + re-rooting + game end + targets + augmentation) against the tuples
the reference's continuous_self_play appended for the same seed (tests/golden/targets_game.npz)."""
import os

import numpy as np
import pytest

from oracle import evaluators

pytestmark = pytest.mark.gpu


def test_selfplay_game_reproduces_reference_appends(golden_dir):
    from transgo_amd.configure import Config
    from transgo_amd.self_play import BatchedSelfPlay
    with np.load(os.path.join(golden_dir, "targets_game.npz")) as z:
        b = {k: z[k] for k in z.files}
    cfg = Config(num_simulation=int(b["sims"]), max_step=int(b["max_step"]))
    sp = BatchedSelfPlay(cfg, 3, evaluator=evaluators.sharp, seed_fn=lambda g, k: int(b["seed"]) + 1000 * g + 7 * k)
    finished = []
    for _ in range(int(b["max_step"])):
        finished += sp.step()
    assert len(finished) == 3 and sp.games_finished == 3
    rec = [r for r in finished if r.seed == int(b["seed"])][0]
    tup = sp.targets(rec)
    assert len(tup) == len(b["z"])
    for i, (o, p, zz, w) in enumerate(tup):
        assert (o == b["obs"][i]).all() and (p == b["pi"][i]).all() and zz == b["z"][i] and (w == b["own"][i]).all(), i
    # the slots restarted with fresh seeds and keep playing
    sp.step()
    assert sp.engine.stats()["errors"] == 0


def test_continuous_self_play_feeds_storage():
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import ReplayMemory_Random
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    cfg = Config(num_simulation=8, max_step=6, num_features=32, num_blocks=2, buffer_size=4096)
    st = SharedStorage({"weights": model.random_weights(9, 10, 32, 2), "now_play_steps": 0, "now_play_games": 0,
                        "now_train_steps": 10 ** 9, "train_play_ratio": 0.075, "adjust_train_play_ratio": True,
                        "game_total_num": 1e8, "adjust_lr": False, "learn_rate": 1e-4}, cfg)
    mem = ReplayMemory_Random(cfg)
    SelfPlay(cfg, n_games=5).continuous_self_play(st, mem, max_moves=6)
    assert st.get_info("now_play_games") == 5 and st.get_info("now_play_steps") == 30
    assert mem.info()["index"] == 5 * 6 * 8
    s, p, z, o = map(np.stack, zip(*mem.sample(32)))     # trainer.py:49
    assert s.shape == (32, 10, 9, 9) and p.shape == (32, 82) and set(np.unique(z)) <= {-1.0, 1.0} and o.shape == (32, 81)


def test_end_to_end_visit_counts_with_the_real_network():
    """Whole engine (tree kernels + MFMA network) against the CPU oracle driving the fp32 torch tower with the same weights
    and seeds.  The two networks differ by ~1e-8, so visit counts can only diverge at near-exact PUCT ties: require the great
    majority of (game, move) visit-count vectors -- and every game's first move -- to be identical."""
    import torch
    from oracle.go_oracle import OracleGoEnv
    from oracle.net import TowerNetwork
    from oracle.wp_mcts import OracleSearch
    from transgo_amd import model
    from transgo_amd.engine import SelfPlayEngine
    torch.set_num_threads(4)
    G, sims, moves, F, NB = 12, 48, 4, 32, 2
    sd = model.random_weights(9, 10, F, NB, seed=77)
    net = TowerNetwork(9, 10, F, NB).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})

    def ev(obs):
        with torch.no_grad():
            p, v, _ = net.main_prediction(torch.from_numpy(obs))
        return p.numpy(), v.numpy()
    eng = SelfPlayEngine(G, num_simulation=sims, net_blocks=NB, net_filters=F)
    model.load_into(eng.ctx, sd, 9, 10, F, NB)
    seeds = np.arange(500, 500 + G)
    eng.reset(seeds)
    orcs = [OracleSearch(OracleGoEnv(), ev, np.random.RandomState(int(s)), num_simulation=sims) for s in seeds]
    same, total, alive = 0, 0, np.ones(G, bool)
    for m in range(moves):
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        acts, pis = eng.choose_moves(vis, st)
        for g, o in enumerate(orcs):
            if not alive[g]:
                continue
            a, pi, obs, info = o.search_move()
            raw = np.array([o.root.kids[i].n if i in o.root.kids else 0 for i in range(82)])
            ok = (raw == vis[g]).all() and a == acts[g]
            total += 1; same += int(ok)
            if m == 0:
                assert ok, (g, "first move must match exactly")
            if not ok:
                alive[g] = False                 # trajectories have split; stop comparing this game
            else:
                o.advance(a)
        eng.play(acts)
    print(f"identical visit-count vectors: {same}/{total}")
    assert same >= 0.85 * total


def test_device_replay_sampler_equals_reference_buffer_plus_trainer_stacking():
    """tg_replay_sample against the reference data path: 8-fold augmented tuples appended in order into the object ring
    (replay_buffer.py:30-34), sampled by index, np.stack'ed and converted to float32 as trainer.py:46-54 does."""
    import torch
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import DeviceReplayMemory, ReplayMemory_Random
    from transgo_amd.self_play import GameRecord, game_targets
    cfg = Config(buffer_size=8 * 64)
    rng = np.random.RandomState(5)
    host = ReplayMemory_Random(cfg)
    dev = DeviceReplayMemory(cfg, capacity_positions=64)
    for g in range(5):                                    # 5 games x 17 positions = 85 > 64: exercises the ring wrap
        r = GameRecord(g)
        for m in range(17):
            r.observations.append((rng.rand(10, 9, 9) < 0.25).astype(np.float32))
            v = rng.randint(0, 30, 82).astype(np.int32); v[rng.randint(82)] += 5; v[rng.randint(82)] = 1
            r.visits.append(v)
            c = np.where(v == 1, 0, v); r.pis.append(c / np.sum(c)); r.players.append(1 + m % 2)
        r.winner = 1 + g % 2; r.territory = rng.randint(-1, 2, 81).astype(np.float32)
        for t in game_targets(r.observations, r.pis, r.players, r.winner, r.territory, 9):
            host.append(*t)
        dev.append_game(r)
    assert dev.info()["entries"] == 64 * 8 and dev.info()["full"] and host.info()["full"]
    assert dev.info()["index"] == host.info()["index"]
    idx = rng.choice(64 * 8, 200, replace=False)
    s, p, z, o = map(np.stack, zip(*host.data[idx]))     # trainer.py:49
    s, p, z, o = (torch.FloatTensor(a).numpy() for a in (s, p, z, o))
    ds, dp, dz, do = dev.sample_entries(idx)
    assert np.array_equal(ds, s) and np.array_equal(dp, p) and np.array_equal(dz, z) and np.array_equal(do, o)
    ds2, dp2, dz2, do2 = dev.sample(32)
    assert ds2.shape == (32, 10, 9, 9) and np.allclose(dp2.sum(1), 1, atol=1e-6)
    # the device-resident form of the same batch (what a trainer on this GPU consumes): identical values, no host copy
    ts, tp, tz, to = dev.sample_entries_device(idx)
    assert ts.is_cuda and ts.dtype == torch.float32 and tuple(ts.shape) == (200, 10, 9, 9)
    assert np.array_equal(ts.cpu().numpy(), s) and np.array_equal(tp.cpu().numpy(), p)
    assert np.array_equal(tz.cpu().numpy(), z) and np.array_equal(to.cpu().numpy(), o)
    assert tuple(dev.sample_device(16)[1].shape) == (16, 82)
    dev.close()


def test_fp16_self_play_through_the_host_mirror():
    """BASELINE config 5's arithmetic end to end: Config.inference_dtype = "f16" reaches tg_config.net_precision, the batched
    self-play loop runs on the fp16 tower, games finish and produce the reference's target tuples (counts consistent, pi a
    distribution, z = +-1)."""
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.self_play import BatchedSelfPlay
    cfg = Config(num_simulation=24, num_features=128, num_blocks=2, max_step=12, inference_dtype="f16")
    sp = BatchedSelfPlay(cfg, 16)
    sp.set_weights(model.random_weights(9, 10, 128, 2, seed=3))
    sp.start()
    finished = []
    for _ in range(14):
        finished += sp.step()
        if len(finished) >= 16:
            break
    assert len(finished) >= 16 and sp.engine.stats()["errors"] == 0
    obs, pi, z, own = sp.targets(finished[0])[0]
    assert obs.shape == (10, 9, 9) and abs(pi.sum() - 1.0) < 1e-9 and z in (-1.0, 1.0) and own.shape == (81,)


def test_plain_c_host_drives_the_engine_through_the_c_abi(tmp_path):
    """examples/c_host_min.c: no Python and no torch between the caller and libtransgo_hip.so."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_host_min")
    libdir = os.path.join(root, "transgo_amd")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_host_min.c"),
                           "-L" + libdir, "-ltransgo_hip", "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "c_host_min ok" in out.stdout, out.stdout + out.stderr
    sims = int(out.stdout.split("c_host_min ok: ")[1].split()[0])
    assert sims >= 3 * 8 * 32


def test_example_selfplay_to_trainer_runs():
    """examples/selfplay_to_trainer.py: complete games -> device replay store -> mini-batches through the reference trainer's own
    batch code and loss (trainer.py:46-72) -> the updated weights back into the engine."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ex_sp2tr", os.path.join(root, "examples", "selfplay_to_trainer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    losses = mod.main(games=16, sims=16, max_step=8, batch=64, steps=3)
    assert len(losses) == 3 and all(np.isfinite(losses))


def test_staggered_start_spreads_game_ends_without_changing_games():
    """BatchedSelfPlay.start(stagger=T): slot g sits parked until step g mod T, so a generation's games end spread over T steps; a
    delayed game is the SAME game (same seed -> same record as in an unstaggered run)."""
    from transgo_amd.configure import Config
    from transgo_amd.self_play import BatchedSelfPlay
    G, T = 12, 6
    cfg = Config(num_simulation=16, max_step=T)
    seed_fn = lambda g, k: 5000 + 100 * g + k
    ref = BatchedSelfPlay(cfg, G, evaluator=evaluators.sharp, seed_fn=seed_fn)
    first = {}
    for _ in range(T):
        for r in ref.step():
            first[r.seed] = r
    assert len(first) == G                                            # unstaggered: all twelve end on step T
    sp = BatchedSelfPlay(cfg, G, evaluator=evaluators.sharp, seed_fn=seed_fn)
    sp.start(stagger=T)
    per_step, got = [], {}
    for _ in range(3 * T):
        recs = sp.step()
        per_step.append(len(recs))
        for r in recs:
            got.setdefault(r.seed, r)
    assert per_step[:T - 1] == [0] * (T - 1) and per_step[T - 1:] == [G // T] * (2 * T + 1)      # two games end on every step
    assert sp.engine.stats()["errors"] == 0 and sp.games_dropped == 0
    for seed, r in first.items():                                     # first-generation games: identical records
        q = got[seed]
        assert q.winner == r.winner and q.players == r.players
        assert all(np.array_equal(a, b) for a, b in zip(q.visits, r.visits))
        assert all(np.array_equal(a, b) for a, b in zip(q.observations, r.observations))


@pytest.mark.parametrize("dtype", ["f32", "f32x3"])
def test_default_actor_takes_a_mainnetwork_state_dict(dtype):
    """(f32x3: the same through the opt-in split-precision mode -- residual blocks on the split convs, every attention block one
    fused kernel, k_attention_x3 -- whose outputs sit as close to the torch network as the exact-f32 path's.)
    What the reference trainer publishes is a MainNetwork state_dict (model.py:49-76: 9 residual + 3 attention blocks, attention
    policy head).  A DEFAULT-configured actor (tower layout) handed that dict through the storage must switch layouts by the key
    names and search with it: one searched move of every game, visit counts equal to the oracle search driving the torch
    restatement of MainNetwork with the same weights and seeds (networks agree to ~1e-7, so allow one near-tie to split)."""
    import torch
    from oracle.go_oracle import OracleGoEnv
    from oracle.net import TransGoMain
    from oracle.wp_mcts import OracleSearch
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import ReplayMemory_Random
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    torch.manual_seed(5); torch.set_num_threads(4)
    net = TransGoMain(9, 10, 128).eval()
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            if hasattr(m, "gamma"):
                m.gamma.copy_(0.5 + torch.rand(1, generator=gen))
    sd = {k: t.numpy() for k, t in net.state_dict().items()}
    cfg = Config(num_simulation=24, max_step=30, buffer_size=1024, inference_dtype=dtype)   # everything else: the reference defaults (tower, 128)
    assert getattr(cfg, "network", "tower") == "tower"
    G = 4
    actor = SelfPlay(cfg, n_games=G)
    assert not actor.worker.arch.policy_attention
    st = SharedStorage({"weights": sd, "now_play_steps": 0, "now_play_games": 0, "now_train_steps": 1,
                        "train_play_ratio": 0.075, "adjust_train_play_ratio": False, "game_total_num": 1e8,
                        "adjust_lr": False, "learn_rate": 1e-4}, cfg)
    actor._refresh_weights(st)                                                 # self_play.py:913
    assert actor.worker.arch.code == model.transgo_arch().code
    eng = actor.worker.engine
    actor.worker.start()
    eng.search()
    vis, rn, pl, stp, ob = eng.root_info()

    def ev(obs):
        with torch.no_grad():
            p, v, _ = net.main_prediction(torch.from_numpy(obs))
        return p.numpy(), v.numpy()
    same = 0
    for g in range(G):
        o = OracleSearch(OracleGoEnv(max_step=30), ev, np.random.RandomState(int(actor.worker.seeds[g])), num_simulation=24)
        a, pi, obs, _ = o.search_move()
        raw = np.array([o.root.kids[i].n if i in o.root.kids else 0 for i in range(82)])
        assert raw.sum() == vis[g].sum() and np.array_equal(obs, ob[g])
        same += int(np.array_equal(raw, vis[g]))
    print(f"MainNetwork through a default actor ({dtype}): {same}/{G} first-move visit vectors identical to the oracle's")
    assert same >= G - 1
    # and the actor loop runs on with it
    actor.continuous_self_play(st, ReplayMemory_Random(cfg), max_moves=2)
    assert st.get_info("now_play_steps") == 2 * G and eng.stats()["errors"] == 0


def test_grouped_selfplay_plays_the_same_games():
    """GroupedSelfPlay: the boards of a GPU as K independent groups on their own contexts / HIP streams, advanced by K host threads
    (the groups' kernels overlap on the GPU).  Nothing about a game may change: slot g of group k is slot k*G/K + g of the job (same
    seeds), and a network row does not depend on which rows share its batch -- after every move the root visit counts of all 24
    games equal the ungrouped engine's bit for bit, and so do the finished games."""
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.self_play import BatchedSelfPlay, GroupedSelfPlay
    cfg = Config(num_simulation=40, max_step=6, num_features=32, num_blocks=2)
    sd = model.random_weights(9, 10, 32, 2, seed=9)
    G, K = 24, 4
    one = BatchedSelfPlay(cfg, G)
    one.set_weights(sd)
    grp = GroupedSelfPlay(cfg, G, groups=K)
    grp.set_weights(sd)
    one.start(); grp.start()
    assert [int(x) for x in one.seeds] == [int(x) for p in grp.parts for x in p.seeds]
    fin_one, fin_grp = [], []
    for move in range(7):                                              # one full generation + the first move of the next
        # search without playing: compare the visit counts the move will be chosen from
        h1 = one.advance()
        hs = grp.advance()
        if h1 is not None:
            fin_one += [(r.seed, r.winner, r.score, [v.tolist() for v in r.visits]) for r in h1.records()]
        for h in hs:
            if h is not None:
                fin_grp += [(r.seed, r.winner, r.score, [v.tolist() for v in r.visits]) for r in h.records()]
        v1 = one.engine.root_visits()[0]
        vg = np.concatenate([p.engine.root_visits()[0] for p in grp.parts])
        assert np.array_equal(v1, vg), move
    assert len(fin_one) == G and sorted(fin_one) == sorted(fin_grp)
    assert grp.stats()["errors"] == 0 and grp.games_finished == one.games_finished == G


def test_actor_loop_with_game_groups():
    """Config.game_groups = K: the actor's boards run as K groups on their own streams; counters, stored tuples and the weight
    refresh behave as with one group."""
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import ReplayMemory_Random
    from transgo_amd.self_play import GroupedSelfPlay, SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    cfg = Config(num_simulation=8, max_step=6, num_features=32, num_blocks=2, buffer_size=4096, game_groups=2)
    st = SharedStorage({"weights": model.random_weights(9, 10, 32, 2), "now_play_steps": 0, "now_play_games": 0,
                        "now_train_steps": 10 ** 9, "train_play_ratio": 0.075, "adjust_train_play_ratio": True,
                        "game_total_num": 1e8, "adjust_lr": False, "learn_rate": 1e-4}, cfg)
    mem = ReplayMemory_Random(cfg)
    actor = SelfPlay(cfg, n_games=6)
    assert isinstance(actor.worker, GroupedSelfPlay) and actor.worker.K == 2
    actor.continuous_self_play(st, mem, max_moves=6)
    assert st.get_info("now_play_games") == 6 and st.get_info("now_play_steps") == 36
    assert mem.info()["index"] == 6 * 6 * 8
    st.set_info({"weights": model.random_weights(9, 10, 32, 2, seed=3), "now_train_steps": 10 ** 9 + 1})
    actor.continuous_self_play(st, mem, max_moves=1)                   # the refresh reaches every group
    assert actor.worker.stats()["errors"] == 0
