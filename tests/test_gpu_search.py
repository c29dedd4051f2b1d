"""HIP tree engine against (a) golden vectors recorded from the imported reference WP_MCTS and (b) the CPU oracle on
many concurrent games: inherited root visits, raw visit counts, chosen action, pi and the MT19937 stream position
must be bit-identical move by move.  The evaluator is a stand-in computed on the host from the observations the GPU
produced (oracle/evaluators.py) or the recorded network outputs, so the network's own rounding is out of the picture."""
import os

import numpy as np
import pytest

from oracle import evaluators

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name)) as z:
        return {k: z[k] for k in z.files}


def _engine(fn, G, sims, **kw):
    from transgo_amd.engine import SelfPlayEngine
    return SelfPlayEngine(G, num_simulation=sims, evaluator=fn, **kw)


def _replay_fixture(blob, tag, fn, max_moves=None, **kw):
    seed, sims = int(tag.split("_")[1][1:]), int(tag.split("_")[2][1:])
    eng = _engine(fn, 1, sims, **kw)
    eng.reset([seed])
    n_moves = len(blob[f"{tag}/action"])
    if max_moves:
        n_moves = min(n_moves, max_moves)
    for m in range(n_moves):
        _, rn0, _, _, _ = eng.root_info(obs=False)
        assert rn0[0] == blob[f"{tag}/n0"][m], (tag, m, "n0")
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        assert (vis[0] == blob[f"{tag}/counts"][m]).all(), (tag, m, "counts", vis[0], blob[f"{tag}/counts"][m])
        assert rn[0] == blob[f"{tag}/root_n"][m] and pl[0] == blob[f"{tag}/player"][m] and st[0] == blob[f"{tag}/step"][m]
        acts, pis = eng.choose_moves(vis, st)
        assert acts[0] == blob[f"{tag}/action"][m], (tag, m, "action")
        assert (pis[0] == blob[f"{tag}/pi"][m]).all(), (tag, m, "pi")
        done = eng.play(acts)
        assert eng.rng_state(0)[1] == blob[f"{tag}/pos"][m], (tag, m, "rng pos")
        assert int(done[0]) == blob[f"{tag}/done"][m]
    return eng


@pytest.mark.parametrize("tag", ["flat_s0_n64", "sharp_s1_n64", "flat_s5_n16", "sharp_s6_n8"])
def test_reference_full_games(golden_dir, tag):
    blob = _load(golden_dir, "search_analytic.npz")
    eng = _replay_fixture(blob, tag, evaluators.BY_NAME[tag.split("_")[0]])
    key, _ = eng.rng_state(0)
    assert (key == blob[f"{tag}/final_key"]).all()
    score, terr, win = eng.final()
    assert score[0] == blob[f"{tag}/final_score"] and (terr[0].astype(np.int8) == blob[f"{tag}/final_terr"]).all()
    assert win[0] == blob[f"{tag}/winner"]
    assert eng.stats()["errors"] == 0


@pytest.mark.parametrize("tag,full", [("sharp_s21_n48", True), ("flat_s22_n24", True), ("sharp_s23_n160", False),
                                      ("sharp_s24_n800", False),      # BASELINE configs[3]'s 800 simulations/move
                                      ("sharp_s25_n1600", False)])    # BASELINE configs[4]'s 1600 simulations/move
def test_reference_19x19_search(golden_dir, tag, full):
    """Board size 19 against vectors recorded from the reference WP_MCTS on a 19x19 build of its engine
    (tests/golden/gen_search19.py): visit counts, moves, pi, RNG position per move; final score / territory of full games."""
    blob = _load(golden_dir, "search_s19.npz")
    eng = _replay_fixture(blob, tag, evaluators.BY_NAME[tag.split("_")[0]], board_size=19, max_step=int(blob["max_step"]))
    key, _ = eng.rng_state(0)
    assert (key == blob[f"{tag}/final_key"]).all()
    if full:
        score, terr, win = eng.final()
        assert score[0] == blob[f"{tag}/final_score"] and (terr[0].astype(np.int8) == blob[f"{tag}/final_terr"]).all()
        assert win[0] == blob[f"{tag}/winner"]
    assert eng.stats()["errors"] == 0


@pytest.mark.parametrize("tag", ["sharp_s2_n210", "flat_s3_n400", "sharp_s4_n400"])
def test_reference_deep_search(golden_dir, tag):
    blob = _load(golden_dir, "search_analytic.npz")
    eng = _replay_fixture(blob, tag, evaluators.BY_NAME[tag.split("_")[0]])
    assert eng.stats()["errors"] == 0


def test_reference_replayed_network(golden_dir):
    blob = _load(golden_dir, "search_replay.npz")
    table = {o.tobytes(): (p, v) for o, p, v in zip(blob["log_obs"], blob["log_policy"], blob["log_value"])}

    def lookup(obs):
        ps, vs = zip(*[table[np.packbits(o.astype(np.uint8).reshape(-1)).tobytes()] for o in obs])
        return np.stack(ps), np.stack(vs)
    _replay_fixture(blob, "real_s11_n64", lookup)


@pytest.mark.parametrize("S,G,sims,moves,max_step", [(9, 32, 48, 24, 120), (19, 6, 40, 10, 450)])
def test_many_games_vs_oracle(S, G, sims, moves, max_step):
    """Concurrent games with different seeds against sequential oracle runs, at both board sizes."""
    from oracle.go_oracle import OracleGoEnv
    from oracle.wp_mcts import OracleSearch
    A = S * S + 1
    fn = evaluators.sharp
    seeds = np.arange(100, 100 + G)
    eng = _engine(fn, G, sims, board_size=S, max_step=max_step)
    eng.reset(seeds)
    orcs = []
    for s in seeds:
        rng = np.random.RandomState(int(s))
        orcs.append((OracleSearch(OracleGoEnv(board_size=S, max_step=max_step), fn, rng, num_simulation=sims, board_size=S), rng))
    for m in range(moves):
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        acts, pis = eng.choose_moves(vis, st)
        for g, (o, rng) in enumerate(orcs):
            a, pi, obs, info = o.search_move()
            raw = np.array([o.root.kids[i].n if i in o.root.kids else 0 for i in range(A)])
            assert (vis[g] == raw).all(), (g, m)
            assert a == acts[g] and (pi == pis[g]).all() and (obs == ob[g]).all(), (g, m)
            o.advance(a)
        eng.play(acts)
        for g, (o, rng) in enumerate(orcs):
            assert eng.rng_state(g)[1] == rng.get_state()[2], (g, m)
    st = eng.stats()
    assert st["errors"] == 0 and st["sims"] > 0


def test_select_action_matches_reference(golden_dir):
    """Evaluation-mode search on the GPU (fresh root from a given position, no noise, temperature 0.12) against actions,
    visit counts and RNG positions recorded from the reference's WP_MCTS.select_action, two agents sharing one stream."""
    from transgo_amd.engine import SelfPlayEngine
    from transgo_amd.environment import GoEnv
    b = _load(golden_dir, "select_action.npz")
    which = {"fn": evaluators.sharp}
    eng = SelfPlayEngine(1, num_simulation=int(b["sims"]), evaluator=lambda obs: which["fn"](obs))
    eng.reset([int(b["seed"])])               # seeds the stream; the throw-away root evaluation draws nothing
    env = GoEnv()
    states = env.reset_batch(1)
    for ply, want in enumerate(b["actions"]):
        which["fn"] = evaluators.sharp if ply % 2 == 0 else evaluators.flat
        eng.reset_from(states)
        eng.search(selfplay=False)
        vis, _, _, steps, _ = eng.root_info(obs=False)
        acts, _ = eng.choose_moves(vis, steps, selfplay=False)
        assert (vis[0] == b["counts"][ply]).all() and acts[0] == want, ply
        assert eng.rng_state(0)[1] == b["pos"][ply], ply
        states, done, _ = env.step_batch(states, acts)


def test_policy_evaluate_runs_matches_and_reports():
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    cfg = Config(num_simulation=8, max_step=14, num_features=32, num_blocks=2)
    w_new, w_old = model.random_weights(9, 10, 32, 2, seed=1), model.random_weights(9, 10, 32, 2, seed=2)
    st = SharedStorage({"weights": w_new, "evaluate_weights": w_old, "evaluate_score": 100}, cfg)
    sp = SelfPlay(cfg, n_games=4)
    ratio, info2, info3 = sp.policy_evaluate(n_games=6, shared_storage_worker=st)
    assert 0.0 <= ratio <= 1.0 and "model player is" in info2 and info3.startswith("evaluate_score:100")
    if ratio == 1:
        assert st.get_info("evaluate_score") == 200 and st.get_info("evaluate_weights") is w_new
    else:
        assert st.get_info("evaluate_score") == 100


def test_policy_evaluate_equals_the_oracle_twin():
    """SelfPlay.policy_evaluate (all games concurrently, two engines, per-game streams handed between them ply by ply) against
    the oracle's sequential restatement of self_play.py:986-1040 with the same per-game seeds: identical winners game by game,
    identical win ratio, identical promotion.  Odd n_games on purpose (the last game is an even one: info2's line)."""
    from oracle.go_oracle import OracleGoEnv
    from oracle.wp_mcts import policy_evaluate as oracle_policy_evaluate
    from transgo_amd.configure import Config
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    cfg = Config(num_simulation=24, max_step=16)
    for fns, n_games, seed in (((evaluators.sharp, evaluators.flat), 5, 300), ((evaluators.flat, evaluators.sharp), 4, 77)):
        st = SharedStorage({"weights": "new", "evaluate_weights": "old", "evaluate_score": 100}, cfg)
        sp = SelfPlay(cfg, n_games=2, evaluator=evaluators.flat)
        ratio, info2, info3 = sp.policy_evaluate(n_games, st, seed=seed, evaluators={"train": fns[0], "eval": fns[1]})
        w, c, want = oracle_policy_evaluate(OracleGoEnv(max_step=16), fns[0], fns[1], n_games, seed, num_simulation=24)
        assert list(sp.last_evaluation["winners"]) == list(w) and list(sp.last_evaluation["colours"]) == list(c)
        assert ratio == want
        assert info2 == "simulate round: {},  winer is : {},  model player is : {}\n".format(n_games, int(w[-1]), int(c[-1]))
        assert info3 == "evaluate_score:100, win: {}, lose: {}\n".format(int((w == c).sum()), int((w != c).sum()))
        if want == 1:
            assert st.get_info("evaluate_score") == 200 and st.get_info("evaluate_weights") == "new"
        else:
            assert st.get_info("evaluate_score") == 100 and st.get_info("evaluate_weights") == "old"
        sp.worker.engine.close()


def test_arena_overflow_parks_the_game_and_the_slot_restarts():
    """A deliberately tiny tree arena: the overflowing games are parked (no out-of-bounds write, no endless search, no
    exception), tg_sp_play reports them, the other entry points stay usable, and resetting the slots starts new games."""
    from transgo_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(4, num_simulation=64, evaluator=evaluators.flat, arena_slots=6 * 84 + 16)
    eng.reset(np.arange(4))
    eng.search()                                               # the search ends: parked games are not active
    err = eng.game_errors()
    assert (err & 1).all() and eng.stats()["errors"] == 4
    vis, st = eng.root_visits()
    done = eng.play(eng.choose_moves(vis, st)[0])
    assert not done.any() and eng.errored.all() and eng.finished.all()
    assert eng.harvest() is None                               # parked games are not finished games
    eng.reset(np.arange(10, 14), eng.errored)                  # new games in the same slots
    assert eng.stats()["errors"] == 0 and not eng.errored.any()
    eng.search(num_simulation=2)                               # one wave (R = 4 read-outs = 4 blocks) fits even this arena
    assert eng.stats()["errors"] == 0
    eng.close()


def test_peaked_policy_full_games_stay_inside_the_default_arena():
    """a confident policy keeps most of the tree on every re-rooting (kept tree ~ sims / (1 - r) blocks).  With the
    default arena the engine must get through whole games without parking any: the re-rooting copy is bounded (deepest blocks
    dropped first, counted), everything else as usual.  Also checks the bound is actually exercised by this evaluator."""
    from transgo_amd.configure import Config
    from transgo_amd.self_play import BatchedSelfPlay
    cfg = Config(num_simulation=96, max_step=60)
    sp = BatchedSelfPlay(cfg, 8, evaluator=evaluators.spike)
    fin, marks = [], []
    for _ in range(64):
        fin += sp.step()
        marks.append(sp.engine.stats()["max_slots"])
    st = sp.engine.stats()
    # max_slots is a high-water mark (per-slot maximum over the run, surviving the games' restarts), not the current fill
    assert all(b >= a for a, b in zip(marks, marks[1:])) and marks[-1] > 0
    print("peaked policy: arena high-water", st["max_slots"], "truncated blocks", st["truncated_blocks"], "finished", len(fin))
    assert st["errors"] == 0 and sp.games_dropped == 0 and len(fin) >= 8
    assert all(len(r.players) == len(r.pis) >= 1 for r in fin)
    # same evaluator, an arena four times smaller than the default: the bound is hit and handled, still no parked game
    sp2 = BatchedSelfPlay(cfg, 8, evaluator=evaluators.spike, arena_slots=((3 * 96 + 256) * 84) // 4)
    for _ in range(40):
        sp2.step()
    st2 = sp2.engine.stats()
    print("quarter arena: high-water", st2["max_slots"], "truncated blocks", st2["truncated_blocks"])
    assert st2["errors"] == 0 and sp2.games_dropped == 0


def test_wp_mcts_mirror_follows_the_oracle_call_by_call():
    """transgo_amd.self_play.WP_MCTS -- the reference's per-game object (self_play.py:575-881) over one engine slot: the same
    calls in the same order (get_action_probs with and without self-play noise, update_with_action, select_action on a given
    position, then searching on) return the oracle's actions, pis and root observations, with one MT19937 stream per object."""
    from oracle.go_oracle import OracleGoEnv
    from oracle.wp_mcts import OracleSearch
    from transgo_amd.configure import Config
    from transgo_amd.self_play import WP_MCTS
    cfg = Config(num_simulation=48, max_step=40)
    seed = 321
    m = WP_MCTS(cfg, evaluator=evaluators.sharp, seed=seed)
    o = OracleSearch(OracleGoEnv(max_step=40), evaluators.sharp, np.random.RandomState(seed), num_simulation=48)
    assert str(m) == "WP_MCTS"
    for ply in range(7):
        selfplay = ply < 4
        a, pi, obs = m.get_action_probs(is_selfplay=selfplay)
        oa, opi, oobs, info = o.search_move(selfplay=selfplay)
        assert a == oa and np.array_equal(pi, opi) and np.array_equal(obs, oobs), ply
        assert m.root.visit_count(a) == (o.root.kids[a].n if a in o.root.kids else 0)
        assert m.update_with_action(a) == o.advance(oa)
    # select_action: fresh tree at the current position, no root noise, temperature 0.12 (self_play.py:689-703) ...
    assert m.select_action(m.root.state) == o.select_action(o.root.state)
    # ... and the object keeps searching from that tree with the same stream
    a, pi, obs = m.get_action_probs(is_selfplay=False)
    oa, opi, oobs, _ = o.search_move(selfplay=False)
    assert a == oa and np.array_equal(pi, opi) and np.array_equal(obs, oobs)
    m.close()


def test_wp_mcts_mirror_reset_root_keeps_the_random_stream():
    """the reference's reset_root (self_play.py:595-605) only rebuilds the root; NumPy's global stream keeps advancing
    across games (continuous_self_play calls it at the top of every game, self_play.py:915).  Two games through reset_root must
    follow the oracle's ONE RandomState -- and therefore differ from each other."""
    from oracle.go_oracle import OracleGoEnv
    from oracle.wp_mcts import OracleSearch
    from transgo_amd.configure import Config
    from transgo_amd.self_play import WP_MCTS
    cfg = Config(num_simulation=32, max_step=6)
    seed = 77
    m = WP_MCTS(cfg, evaluator=evaluators.sharp, seed=seed)
    o = OracleSearch(OracleGoEnv(max_step=6), evaluators.sharp, np.random.RandomState(seed), num_simulation=32)
    games = []
    for game in range(3):
        if game:
            m.reset_root(); o.reset_root()
        moves = []
        for ply in range(6):
            a, pi, obs = m.get_action_probs(is_selfplay=True)
            oa, opi, oobs, _ = o.search_move(selfplay=True)
            assert a == oa and np.array_equal(pi, opi) and np.array_equal(obs, oobs), (game, ply)
            moves.append((a, pi.tobytes()))
            done = m.update_with_action(a)
            assert done == o.advance(oa)
            if done:
                break
        games.append(moves)
    key, pos = m.engine.rng_state(0)
    ok, opos = o.rng.get_state()[1], o.rng.get_state()[2]
    assert np.array_equal(key, ok) and pos == opos
    assert games[0] != games[1] and games[1] != games[2]          # re-seeding every game would make them identical
    m.close()


def test_tree_pool_is_shared_accounted_and_survives_exhaustion():
    """Round 4: all games of a context take their tree chunks from ONE pool (provisioned for the population, not for the worst game
    times the number of games).  (1) accounting: after a full reset every game owns exactly one chunk; searching grows the fill,
    re-rooting returns the old trees, a full reset returns everything.  (2) a pool far too small for its games: games that find it
    empty are parked and reported exactly like games that hit their own cap -- nothing faults, the others finish their searches,
    restarting the parked slots works -- and the games that did fit searched exactly what the same seeds search in a roomy pool."""
    from transgo_amd.engine import SelfPlayEngine
    G, sims, CH = 48, 64, 1024
    roomy = SelfPlayEngine(G, num_simulation=sims, evaluator=evaluators.sharp)
    roomy.reset(np.arange(G))
    p0 = roomy.pool_stats()
    assert p0["pool_in_use"] == G * CH and p0["pool_exhausted"] == 0 and p0["pool_slots"] >= G * (2 * sims + 128) * 84
    roomy.search()
    vis_r, st_r = roomy.root_visits()
    p1 = roomy.pool_stats()
    assert p1["pool_in_use"] > p0["pool_in_use"] and p1["pool_high_water"] >= p1["pool_in_use"] and p1["pool_exhausted"] == 0
    acts = roomy.choose_moves(vis_r, st_r)[0]
    roomy.play(acts)
    p2 = roomy.pool_stats()
    assert p2["pool_in_use"] < p1["pool_in_use"]                  # the old trees went back, the kept sub-trees are smaller
    assert p2["pool_high_water"] >= p1["pool_in_use"]
    roomy.reset(np.arange(G))
    assert roomy.pool_stats()["pool_in_use"] == G * CH
    s = roomy.stats()
    assert s["errors"] == 0 and s["pool_exhausted"] == 0 and s["max_slots"] % CH == 0 and s["max_slots"] >= 2 * CH
    # the smallest pool the library accepts: 2 chunks per game + one largest tree
    tiny = SelfPlayEngine(G, num_simulation=sims, evaluator=evaluators.sharp, pool_slots=1)
    tiny.reset(np.arange(G))
    t0 = tiny.pool_stats()
    assert t0["pool_slots"] < p0["pool_slots"] / 4 and t0["pool_in_use"] == G * CH
    tiny.search()
    err = tiny.game_errors()
    t1 = tiny.pool_stats()
    # (games of one context advance in lock step: when the pool runs dry they all ask for their next chunk within a wave or two)
    assert t1["pool_exhausted"] > 0 and (err != 0).sum() > 0 and ((err == 0) | (err == 1)).all()
    assert t1["pool_in_use"] <= t1["pool_slots"]
    vis_t, st_t = tiny.root_visits()
    ok = err == 0
    assert (vis_t[ok] == vis_r[ok]).all()                         # where the pool sufficed, the same search as in the roomy pool
    done = tiny.play(tiny.choose_moves(vis_t, st_t)[0])
    assert (tiny.errored == (err != 0)).all() and not done[ok].any()
    tiny.reset(np.arange(100, 100 + G), tiny.errored)              # the parked slots hand their chunks back and start new games,
                                                                   # even when the pool had run completely dry
    assert tiny.stats()["errors"] == 0
    tiny.search(num_simulation=2)                                  # the engine goes on (which games fit next is up to the pool)
    t2 = tiny.pool_stats()
    assert t2["pool_in_use"] <= t2["pool_slots"] and t2["pool_high_water"] <= t2["pool_slots"]
    roomy.close(); tiny.close()


def test_policy_evaluate_against_the_reference_recording(golden_dir):
    """Row f1 against the reference itself: for single evaluation games the reference's one NumPy stream and this engine's per-game
    streams (seed + i) are the same stream, so SelfPlay.policy_evaluate must return what the reference's policy_evaluate returned in
    tests/golden/policy_evaluate.json -- win ratio, both info strings, and the promotion it wrote (or did not write) to the storage."""
    import json
    import os
    from transgo_amd.configure import Config
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    fns = {"sharp": evaluators.sharp, "flat": evaluators.flat}
    with open(os.path.join(golden_dir, "policy_evaluate.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["n_games"] == 1]
    assert len(cases) >= 2 and {bool(c["promoted"]) for c in cases} == {True, False}
    for c in cases:
        cfg = Config(num_simulation=c["sims"], max_step=c["max_step"], komi=c["komi"])
        st = SharedStorage({"weights": c["train"], "evaluate_weights": c["evalu"], "evaluate_score": 100}, cfg)
        sp = SelfPlay(cfg, n_games=2, evaluator=evaluators.flat)
        ratio, info2, info3 = sp.policy_evaluate(1, st, seed=c["seed"], evaluators={"train": fns[c["train"]], "eval": fns[c["evalu"]]})
        assert ratio == c["ratio"] and info2 == c["info2"] and info3 == c["info3"]
        assert list(sp.last_evaluation["winners"]) == c["winners"] and list(sp.last_evaluation["colours"]) == c["colours"]
        if c["promoted"]:
            assert st.get_info("evaluate_score") == c["score_written"][0] and st.get_info("evaluate_weights") == c["train"]
        else:
            assert st.get_info("evaluate_score") == 100 and st.get_info("evaluate_weights") == c["evalu"]
        sp.worker.engine.close()
