"""The finished-game path that lives on the device: the per-move record written by tg_sp_play, tg_sp_harvest's target
generation (self_play.py:929-967), the device -> device append into the replay store, the multi-rank actor loop -- against the
same material assembled on the host the way the reference's self-play loop does (self_play.py:917-926: append root observation,
pi and player per move; :932-940 score, winner, territory at the end)."""
import socket

import numpy as np
import pytest

from oracle import evaluators

pytestmark = pytest.mark.gpu


def _play_and_assemble(eng, seeds, moves):
    """Drive the engine move by move, keeping the reference's three per-game lists on the host from root_info; returns
    ({slot: (obs list, visits list, players list, winner, territory, score)} for finished games, [harvest batches])."""
    G = eng.G
    lists = [([], [], []) for _ in range(G)]
    finished, batches = {}, []
    eng.reset(seeds)
    for _ in range(moves):
        live = ~eng.finished
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        acts, pis = eng.choose_moves(vis, st)
        for g in np.flatnonzero(live):
            lists[g][0].append(ob[g].copy()); lists[g][1].append(vis[g].copy()); lists[g][2].append(int(pl[g]))
        done = eng.play(acts)
        if done.any():
            score, terr, win = eng.final()
            for g in np.flatnonzero(done & live):
                finished[int(g)] = lists[g] + (int(win[g]), terr[g].copy(), float(score[g]))
            batches.append((eng.harvest(device=False, seeds=seeds), eng.harvest(device=True, seeds=seeds)))
        if eng.finished.all():
            break
    return finished, batches


@pytest.mark.parametrize("S,G,sims,max_step", [(9, 7, 24, 14), (19, 3, 64, 9)])
def test_device_records_equal_host_assembled_records(S, G, sims, max_step):
    from transgo_amd.engine import SelfPlayEngine
    from transgo_amd.self_play import game_targets
    eng = SelfPlayEngine(G, board_size=S, num_simulation=sims, max_step=max_step, evaluator=evaluators.sharp)
    seeds = np.arange(40, 40 + G).astype(np.uint32)
    finished, batches = _play_and_assemble(eng, seeds, max_step + 1)
    assert len(finished) == G and batches
    seen = set()
    for host, dev in batches:
        assert not host.on_device and dev.on_device and host.nbytes == dev.nbytes
        assert np.array_equal(host.buf, dev.to_host().buf)           # the device batch is byte-identical to the host one
        slots = host.view("slot")
        assert list(slots) == sorted(slots)
        recs = host.records()
        tup = host.targets()
        o = 0
        for i, g in enumerate(slots):
            obs, vis, pls, win, terr, score = finished[int(g)]
            r = recs[i]
            assert r.seed == int(seeds[g]) and r.winner == win and r.score == score and np.array_equal(r.territory, terr)
            assert r.players == pls and len(r.observations) == len(obs) == host.view("n_moves")[i]
            assert all(np.array_equal(a, b) for a, b in zip(r.observations, obs))
            assert all(np.array_equal(a, b) for a, b in zip(r.visits, vis))
            want = game_targets(obs, r.pis, pls, win, terr, S)       # the reference's appends for this game
            got = tup[o:o + len(want)]
            assert all(all(np.array_equal(p, q) for p, q in zip(x, y)) for x, y in zip(got, want))
            o += len(want)
            seen.add(int(g))
        assert o == len(tup)
    assert seen == set(range(G))
    eng.close()


def test_harvest_feeds_the_device_replay_store_without_the_host():
    """BatchedSelfPlay.advance(device=True) -> DeviceReplayMemory.append_harvest (tg_replay_append_dev): sampled entries equal
    the reference data path (8 augmented tuples per position appended in order, np.stack, float32; trainer.py:46-54)."""
    import torch
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import DeviceReplayMemory, ReplayMemory_Random
    from transgo_amd.self_play import BatchedSelfPlay
    cfg = Config(num_simulation=16, max_step=10, buffer_size=8 * 4096)
    sp = BatchedSelfPlay(cfg, 12, evaluator=evaluators.flat)
    dev = DeviceReplayMemory(cfg, capacity_positions=4096)
    host = ReplayMemory_Random(cfg)
    games = 0
    for _ in range(23):                                              # two generations of games and a bit
        h = sp.advance(device=True)
        if h is None:
            continue
        assert h.on_device
        dev.append_harvest(h)
        for t in h.targets():
            host.append(*t)
        games += h.n_games
    assert games == 24 and sp.games_finished == 24 and dev.info()["entries"] == host.info()["index"] == 24 * 10 * 8
    idx = np.random.RandomState(0).choice(host.info()["index"], 300, replace=False)
    s, p, z, o = map(np.stack, zip(*host.data[idx]))                 # trainer.py:49
    s, p, z, o = (torch.FloatTensor(a).numpy() for a in (s, p, z, o))
    ds, dp, dz, do = dev.sample_entries(idx)
    assert np.array_equal(ds, s) and np.array_equal(dp, p) and np.array_equal(dz, z) and np.array_equal(do, o)
    dev.close()


def test_allgather_transport_on_device_buffers_with_a_simulated_peer(monkeypatch):
    """The default finished-game transport over RCCL has never had a second rank on this project's one-GPU boxes.  Everything of it
    that is not the collective itself -- padding the device harvest buffer, cutting the gathered buffers back to exact lengths,
    handing a PEER's games to the device replay store as a view of the gathered buffer (no copy queued on torch's stream behind
    which tg_replay_append_dev, on the library's own stream, would have to wait) -- runs here on the GPU with `all_gather` replaced
    by a local stand-in that delivers a second rank's (different, longer) buffer."""
    import torch
    import torch.distributed as dist
    from transgo_amd import distributed, records
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import DeviceReplayMemory, ReplayMemory_Random
    from transgo_amd.self_play import BatchedSelfPlay
    cfg = Config(num_simulation=16, max_step=6, buffer_size=8 * 4096)
    dev = torch.device("cuda", 0)

    def finished(n_games, seed0):
        sp = BatchedSelfPlay(cfg, n_games, evaluator=evaluators.flat, seed_fn=lambda g, k: seed0 + g + 100 * k)
        for _ in range(6):
            h = sp.advance(device=True)
        assert h is not None and h.on_device and h.n_games == n_games
        return h
    mine, peer = finished(5, 10), finished(3, 500)                    # rank 0 finished 5 games, "rank 1" 3: its buffer arrives padded
    sizes = [(mine.n_games, mine.n_positions), (peer.n_games, peer.n_positions)]

    def fake_all_gather(bucket, pad):
        assert pad.is_cuda and len(bucket) == 2 and all(b.is_cuda and b.numel() == pad.numel() for b in bucket)
        bucket[0].copy_(pad)
        bucket[1].zero_(); bucket[1][:peer.nbytes] = peer.buf
    monkeypatch.setattr(dist, "all_gather", fake_all_gather)
    monkeypatch.delenv("TRANSGO_GATHER", raising=False)
    out = distributed._gather_payloads(mine, 9, 10, 0, sizes, 2, 0, True, dev)
    assert len(out) == 2 and out[0] is mine and out[1].on_device and out[1].nbytes == peer.nbytes < out[1].buf.untyped_storage().nbytes()
    assert torch.equal(out[1].buf, peer.buf)
    store, host = DeviceReplayMemory(cfg, capacity_positions=1024), ReplayMemory_Random(cfg)
    for hb in out:
        store.append_harvest(hb)
        for t in hb.targets():
            host.append(*t)
    n = host.info()["index"]
    assert store.info()["entries"] == n == (mine.n_positions + peer.n_positions) * 8
    idx = np.arange(n)
    s, p, z, o = map(np.stack, zip(*host.data[idx]))
    s, p, z, o = (torch.FloatTensor(a).numpy() for a in (s, p, z, o))
    ds, dp, dz, do = store.sample_entries(idx)
    assert np.array_equal(ds, s) and np.array_equal(dp, p) and np.array_equal(dz, z) and np.array_equal(do, o)
    # a rank that is not the owner contributes and receives nothing
    assert distributed._gather_payloads(peer, 9, 10, 0, sizes, 2, 1, True, dev) == []
    store.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _actor_worker(rank, world, port, q, backend):
    """One rank of the actor loop: its own shard of games on the GPU, finished games gathered to rank 0, only rank 0 appends
    and counts.  backend gloo = both ranks on GPU 0 (payloads staged through the host); nccl = one GPU per rank."""
    import torch
    import torch.distributed as dist
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import DeviceReplayMemory, ReplayMemory_Random
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    from transgo_amd import distributed
    gpu = rank if backend == "nccl" else 0
    distributed.init_process_group(backend, rank, world, device_index=gpu, timeout_s=300.0, init_method=f"tcp://127.0.0.1:{port}")
    cfg = Config(num_simulation=24, max_step=5, buffer_size=8 * 1024)
    G = 3 + rank                                                     # ragged shards
    actor = SelfPlay(cfg, n_games=G, device=gpu, rank=rank, world=world, evaluator=evaluators.sharp)
    st = mem = None
    if rank == 0:
        # the trainer is BEHIND when the first games finish (now_train_steps 0) and catches up 3 s into the run: every rank must
        # sit out the same 0.5-s rounds on the host -- nobody inside a pending collective -- and then resume
        st = SharedStorage({"weights": None, "now_play_steps": 0, "now_play_games": 0, "now_train_steps": 0,
                            "train_play_ratio": 0.075, "adjust_train_play_ratio": True, "game_total_num": 1e8,
                            "adjust_lr": False, "learn_rate": 1e-4}, cfg)
        mem = DeviceReplayMemory(cfg, capacity_positions=1024, device=gpu) if backend == "nccl" else ReplayMemory_Random(cfg)
        import threading
        import time as _t

        def trainer():
            _t.sleep(3.0)
            st.set_info("now_train_steps", 10 ** 9)
        threading.Thread(target=trainer, daemon=True).start()
    actor.continuous_self_play(st, mem, max_moves=11)                # 2 full generations (5 moves each) + 1 move
    out = {"rank": rank, "finished_local": actor.worker.games_finished, "G": G, "dropped": actor.worker.games_dropped,
           "rounds": actor.throttle_rounds}
    if rank == 0:
        info = mem.info()
        out.update(steps=st.get_info("now_play_steps"), games=st.get_info("now_play_games"),
                   entries=info.get("entries", info["index"]))
    # second scenario: the real network, weights published on rank 0 only -> packed there, broadcast, loaded in the background on
    # every rank (self_play.py:913 for all actors at once); two weight versions, then both ranks must hold the second one
    from transgo_amd import model
    cfg2 = Config(num_simulation=8, max_step=30, num_features=32, num_blocks=2)
    actor2 = SelfPlay(cfg2, n_games=2, device=gpu, rank=rank, world=world)
    st2 = None
    if rank == 0:
        st2 = SharedStorage({"weights": model.random_weights(9, 10, 32, 2, seed=1), "now_play_steps": 0, "now_play_games": 0,
                             "now_train_steps": 1, "train_play_ratio": 0.075, "adjust_train_play_ratio": False,
                             "game_total_num": 1e8, "adjust_lr": False, "learn_rate": 1e-4}, cfg2)
    actor2.continuous_self_play(st2, ReplayMemory_Random(cfg2) if rank == 0 else None, max_moves=2)
    if rank == 0:
        st2.set_info({"weights": model.random_weights(9, 10, 32, 2, seed=2), "now_train_steps": 2})
    actor2.continuous_self_play(st2, ReplayMemory_Random(cfg2) if rank == 0 else None, max_moves=2)
    import ctypes
    pend = ctypes.c_int(-1)
    actor2.worker.engine.ctx.call("tg_net_load_poll", 1, ctypes.byref(pend))
    h = model.HipNetwork(9, 10, 32, 2, rows_cap=4, device=gpu)
    h.set_weights(model.random_weights(9, 10, 32, 2, seed=2))
    probe = (np.random.RandomState(5).rand(3, 10, 9, 9) < 0.2).astype(np.float32)
    want = h.main_prediction(probe)
    pol = np.empty((3, 82), np.float32); val = np.empty(3, np.float32)
    actor2.worker.engine.ctx.call("tg_net_predict", probe.ctypes.data_as(ctypes.c_void_p), 3, pol.ctypes.data_as(ctypes.c_void_p),
                                  val.ctypes.data_as(ctypes.c_void_p), None)
    out["weights_ok"] = bool(np.array_equal(pol, want[0]) and np.array_equal(val, want[1].reshape(-1)))
    q.put(out)
    dist.destroy_process_group()


def _run_actor_ranks(backend):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_actor_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda d: d["rank"])
    [p.join(120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    r0, r1 = res
    assert r0["dropped"] == r1["dropped"] == 0
    assert r0["rounds"] == r1["rounds"] >= 1                          # the stalled trainer was waited for by both ranks together
    assert r0["weights_ok"] and r1["weights_ok"]                      # the second weight version reached both ranks' GPUs
    assert r0["finished_local"] == 2 * 3 and r1["finished_local"] == 2 * 4
    assert r0["games"] == 14                                          # every rank's finished games reached the owner
    assert r0["entries"] == 14 * 5 * 8                                # 5 positions per game, 8 reference entries per position
    # now_play_steps: one per move actually played, summed over the ranks inside the gather's size exchange (3 + 4 live slots)
    assert r0["steps"] == 11 * 7
    return res


def test_actor_loop_two_ranks_gloo_on_one_gpu():
    """SelfPlay.continuous_self_play with torch.distributed initialised: the ACTOR path, not just the
    gather helper -- two processes share GPU 0, gloo carries the payloads."""
    _run_actor_ranks("gloo")


def test_actor_loop_two_ranks_rccl():
    """The same over RCCL with device-resident payloads end to end (harvest in HBM -> send/recv -> tg_replay_append_dev).
    Needs two GPUs; skipped on the one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_actor_ranks("nccl")


@pytest.mark.parametrize("C", [9, 13])
def test_other_plane_counts_through_the_engine(C):
    """encode9 / encode13 (board_feature.cc:225-253) through the ENGINE's bit-packed paths -- root observation, evaluation batch,
    per-move record, harvest -- against the oracle environment replaying the same moves (the 10-plane default is covered by every
    other test; the rules fixtures cover the plane variants only through GoEnv.encode)."""
    from oracle.go_oracle import OracleGoEnv
    from transgo_amd.engine import SelfPlayEngine
    G, moves = 5, 10
    seen = []

    def ev(obs):
        seen.append(np.asarray(obs).copy())
        return evaluators.sharp(obs)
    eng = SelfPlayEngine(G, num_simulation=24, max_step=moves, encode_dim=C, evaluator=ev)
    env = OracleGoEnv(max_step=moves, encode_dim=C)
    seeds = np.arange(70, 70 + G).astype(np.uint32)
    eng.reset(seeds)
    states = [env.reset()[0] for _ in range(G)]
    roots = [[] for _ in range(G)]
    for m in range(moves):
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        assert ob.shape == (G, C, 9, 9)
        for g in range(G):
            want = env.encode(states[g])
            assert np.array_equal(ob[g], want), (C, m, g)
            roots[g].append(want)
        acts, _ = eng.choose_moves(vis, st)
        done = eng.play(acts)
        states = [env.step(states[g], int(acts[g]))[0] for g in range(G)]
    assert done.all() and not eng.game_errors().any()
    h = eng.harvest(device=False, seeds=seeds)
    assert h.n_games == G and h.n_positions == G * moves
    obs = h.observations().reshape(G, moves, C, 9, 9)
    for i, g in enumerate(h.view("slot")):
        assert np.array_equal(obs[i], np.stack(roots[int(g)]))
    # every batch the evaluator saw is a stack of legal C-plane encodings: planes are 0/1 and stones never overlap
    for b in seen:
        assert b.shape[1] == C and set(np.unique(b)) <= {0.0, 1.0}
        assert (b[:, 0:3].sum(1) + b[:, 3:6].sum(1)).max() <= 1.0
    eng.close()
