"""Edge cases of the hot path on the GPU: empty batches, a position where pass is the only move and the game ends by double pass
(self_play.py:857-872 + go_env.cc:60-66), finished positions handed to the engine, nothing-to-harvest, out-of-range replay
indices -- the cases the reference's own code guards (environment.py:121-129 pass filter, go_env.cc:52-55 finished states)."""
import ctypes

import numpy as np
import pytest

from oracle import evaluators

pytestmark = pytest.mark.gpu


def test_empty_batches_are_noops():
    from transgo_amd.environment import GoEnv
    from transgo_amd.model import HipNetwork, random_weights
    env = GoEnv()
    st = env.reset_batch(0)
    assert st.shape == (0, env.ssz)
    out, done, ok = env.step_batch(st, np.zeros(0, np.int32))
    assert out.shape == (0, env.ssz) and len(done) == 0 and len(ok) == 0
    q = env.query_batch(st, legal=True, obs=True, score=True, terr=True, meta=True)
    assert q["legal"].shape == (0, 82) and q["obs"].shape == (0, 10, 9, 9)
    h = HipNetwork(9, 10, 32, 2, rows_cap=4)
    h.set_weights(random_weights(9, 10, 32, 2))
    p, v, o = h.main_prediction(np.zeros((0, 10, 9, 9), np.float32))
    assert p.shape == (0, 82) and v.shape == (0, 1) and o.shape == (0, 81)


def _pass_only_state(seed):
    """Random legal play (never filling an own true eye) on the CPU oracle until the side to move has no board move left."""
    from oracle.go_oracle import OracleGoEnv
    env = OracleGoEnv(max_step=400)
    rng = np.random.RandomState(seed)
    s, done = env.reset()
    trail = []
    while True:
        la = list(env.getLegalAction(s))
        if la == [81]:
            return env, s, trail
        noeye = [a for a in env.getLegalNoEye(s) if a != 81]
        a = int(noeye[rng.randint(len(noeye))]) if noeye else int(la[rng.randint(len(la))])
        s, done = env.step(s, a)
        trail.append(a)
        assert not done


def test_pass_only_position_ends_by_double_pass():
    """The side to move has no board move left: the root has the single child `pass` (environment.py:126-127), the search still
    runs its simulations through it, and -- the opponent having passed just before (last_move1 = PASS in the handed-over state) --
    the move ends the game by double pass (go_env.cc:60-63) long before the ply limit; the record holds exactly that one move."""
    from transgo_amd import _lib
    from transgo_amd.engine import SelfPlayEngine
    from transgo_amd.environment import GoEnv
    trails = [_pass_only_state(seed)[2] for seed in range(3)]
    G = len(trails)
    cfg = _lib.default_config(); cfg.max_step = 400; cfg.n_games = 0
    renv = GoEnv()
    renv.ctx.close(); renv.ctx = _lib.Context(cfg); renv.max_step = 400      # rules context with the oracle's ply limit
    states = renv.reset_batch(G)
    for t in range(max(len(c) for c in trails)):
        acts = np.array([c[t] if t < len(c) else -9 for c in trails], np.int32)
        live = acts != -9
        nxt, d, ok = renv.step_batch(states, np.where(live, acts, 0))
        assert ok[live].all() and not d[live].any()
        states[live] = nxt[live]
    legal = renv.query_batch(states, legal=True)["legal"]
    assert (legal[:, :81].sum(1) == 0).all()                    # pass is the only move
    # "the opponent has just passed": last_move1 (int16 right behind the two 128-bit bitboards of the 48-byte state) = PASS (-1)
    states[:, 32:34] = 0xFF
    eng = SelfPlayEngine(G, num_simulation=16, max_step=400, evaluator=evaluators.flat)
    eng.reset(np.arange(G))
    eng.reset_from(states)
    eng.search()
    vis, rn, pl, st, _ = eng.root_info(obs=False)
    assert (vis[:, :81] == 0).all() and (vis[:, 81] >= 15).all()   # every simulation goes through the pass child (terminal: +-1 backups)
    acts, pis = eng.choose_moves(vis, st)
    assert (acts == 81).all() and (pis[:, 81] == 1.0).all()
    done = eng.play(acts)
    assert done.all()
    h = eng.harvest()
    assert h.n_games == G and list(h.view("n_moves")) == [1] * G
    score, terr, win = eng.final()
    want = renv.query_batch(eng.root_states(), score=True)["score"]
    assert np.array_equal(score, want) and list(h.view("winner")) == [1 if s > 0 else 2 for s in want] == list(win)
    assert eng.stats()["errors"] == 0
    eng.close()


def test_finished_positions_and_empty_harvest():
    """Handing the engine an already finished position parks the slot (go_env.cc:52-55: stepping it changes nothing); a harvest
    with nothing finished is None; a replay entry index out of range is an error, not a read."""
    from transgo_amd._lib import TransgoError
    from transgo_amd.configure import Config
    from transgo_amd.engine import SelfPlayEngine
    from transgo_amd.environment import GoEnv
    from transgo_amd.replay_buffer import DeviceReplayMemory
    env = GoEnv()
    states = env.reset_batch(2)
    nxt, d, _ = env.step_batch(states, [81, 81]); nxt, d, _ = env.step_batch(nxt, [81, 40])      # game 0: pass, pass -> over
    assert list(d) == [True, False]
    eng = SelfPlayEngine(2, num_simulation=8, evaluator=evaluators.sharp)
    eng.reset([1, 2])
    assert eng.harvest() is None
    eng.reset_from(nxt)
    eng.search()
    vis, st = eng.root_visits()
    done = eng.play(eng.choose_moves(vis, st)[0])
    assert not done[1] and eng.finished[0]
    assert eng.harvest() is None                               # slot 0 was over before the engine ever moved in it
    mem = DeviceReplayMemory(Config(), capacity_positions=8)
    with pytest.raises(TransgoError):
        mem.sample_entries([0])                                # empty store
    mem.close(); eng.close()


def test_c_abi_refuses_bad_calls_with_a_message():
    """Error behaviour of the boundary (include/transgo_hip.h): bad configurations and out-of-order calls come back as a negative
    status with tg_last_error set -- never a crash, never a silent fallback."""
    from transgo_amd import _lib
    from transgo_amd._lib import TransgoError
    from transgo_amd.engine import SelfPlayEngine
    cfg = _lib.default_config(); cfg.board_size = 13
    with pytest.raises(TransgoError, match="board_size"):
        _lib.Context(cfg)
    cfg = _lib.default_config(); cfg.n_games = 2; cfg.max_step = 2000
    with pytest.raises(TransgoError, match="max_step"):
        _lib.Context(cfg)
    eng = SelfPlayEngine(2, num_simulation=8, evaluator=evaluators.flat, record_games=False)
    with pytest.raises(TransgoError, match="first reset"):
        eng.ctx.call("tg_sp_reset", np.zeros(2, np.uint32).ctypes.data_as(ctypes.c_void_p), np.ones(2, np.uint8).ctypes.data_as(ctypes.c_void_p))
    eng.reset([1, 2])
    with pytest.raises(TransgoError, match="no leaf batch"):
        eng.ctx.call("tg_sp_absorb")
    eng.search()
    vis, st = eng.root_visits()
    bad = np.full(2, 81, np.int32)                             # pass is not among the root's children while board moves exist
    done = np.zeros(2, np.uint8)
    eng.ctx.call("tg_sp_play", bad.ctypes.data_as(ctypes.c_void_p), done.ctypes.data_as(ctypes.c_void_p))
    assert list(done) == [2, 2] and (eng.game_errors() & 4).all()          # refused per game, reported, the engine stays usable
    eng._evaluate(); eng.ctx.call("tg_sp_expand_roots")                    # close the (empty) root batch tg_sp_play opened
    eng.reset([3, 4], np.ones(2, np.uint8))
    eng.search(); vis, st = eng.root_visits()
    assert eng.play(eng.choose_moves(vis, st)[0]).sum() == 0 and eng.stats()["errors"] == 0
    with pytest.raises(TransgoError, match="record_games"):
        _force_harvest()
    blob = np.zeros(10, np.float32)
    with pytest.raises(TransgoError, match="blob size"):
        eng.ctx.call("tg_net_load", blob.ctypes.data_as(ctypes.c_void_p), blob.size, 0)
    eng.close()


def _force_harvest():
    """tg_sp_harvest on a context created without game records, with a finished game pending."""
    from transgo_amd.engine import SelfPlayEngine
    e2 = SelfPlayEngine(1, num_simulation=8, max_step=1, evaluator=evaluators.flat, record_games=False)
    e2.reset([5]); e2.search()
    vis, st = e2.root_visits()
    assert e2.play(e2.choose_moves(vis, st)[0]).all()
    try:
        e2.harvest()
    finally:
        e2.close()
