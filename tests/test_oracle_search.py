"""The search oracle (oracle/wp_mcts.py) against golden vectors recorded from the imported reference WP_MCTS
(tests/golden/gen_search.py): inherited visits, raw child visit counts, chosen action, pi and MT19937 stream position
must all be bit-identical, move by move."""
import os

import numpy as np
import pytest

from oracle import evaluators
from oracle.go_oracle import OracleGoEnv
from oracle.wp_mcts import OracleSearch


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name)) as z:
        return {k: z[k] for k in z.files}


def replay_case(blob, tag, fn, max_moves=None, size=9, max_step=120):
    name, seed, sims = tag.split("_")
    seed, sims = int(seed[1:]), int(sims[1:])
    env = OracleGoEnv(board_size=size, max_step=max_step)
    rng = np.random.RandomState(seed)
    s = OracleSearch(env, fn, rng, num_simulation=sims, board_size=size)
    n_moves = len(blob[f"{tag}/action"])
    if max_moves:
        n_moves = min(n_moves, max_moves)
    for m in range(n_moves):
        n0 = s.root.n
        a, pi, obs, info = s.search_move()
        raw = np.array([s.root.kids[i].n if i in s.root.kids else 0 for i in range(size * size + 1)])
        assert n0 == blob[f"{tag}/n0"][m], (tag, m)
        assert (raw == blob[f"{tag}/counts"][m]).all(), (tag, m)
        assert a == blob[f"{tag}/action"][m], (tag, m)
        assert (pi == blob[f"{tag}/pi"][m]).all(), (tag, m)
        assert s.root.n == blob[f"{tag}/root_n"][m]
        done = s.advance(a)
        assert rng.get_state()[2] == blob[f"{tag}/pos"][m], (tag, m)
        assert int(done) == blob[f"{tag}/done"][m]
    return s, rng


@pytest.mark.parametrize("tag", ["flat_s0_n64", "sharp_s1_n64", "flat_s5_n16", "sharp_s6_n8"])
def test_full_games(golden_dir, tag):
    blob = _load(golden_dir, "search_analytic.npz")
    s, rng = replay_case(blob, tag, evaluators.BY_NAME[tag.split("_")[0]])
    assert (rng.get_state()[1] == blob[f"{tag}/final_key"]).all()
    env = s.env
    score, terr = env.getScoreAndTerritory(s.root.state)
    assert score == blob[f"{tag}/final_score"] and (terr.astype(np.int8) == blob[f"{tag}/final_terr"]).all()
    assert env.getWinner(s.root.state) == blob[f"{tag}/winner"]


@pytest.mark.parametrize("tag,moves", [("sharp_s2_n210", 12), ("flat_s3_n400", 4), ("sharp_s4_n400", 4)])
def test_deep_search(golden_dir, tag, moves):
    blob = _load(golden_dir, "search_analytic.npz")
    replay_case(blob, tag, evaluators.BY_NAME[tag.split("_")[0]], max_moves=moves)


@pytest.mark.parametrize("tag,full", [("sharp_s21_n48", True), ("flat_s22_n24", True), ("sharp_s23_n160", False),
                                      ("sharp_s24_n800", False),      # BASELINE configs[3]: 800 simulations/move
                                      ("sharp_s25_n1600", False)])    # BASELINE configs[4]: 1600 simulations/move
def test_19x19_search_matches_reference(golden_dir, tag, full):
    """The same observables at board size 19, recorded from the reference WP_MCTS running on a 19x19 build of its engine
    (tests/golden/gen_search19.py): two games to the ply limit and one deeper search."""
    blob = _load(golden_dir, "search_s19.npz")
    s, rng = replay_case(blob, tag, evaluators.BY_NAME[tag.split("_")[0]], size=19, max_step=int(blob["max_step"]))
    assert (rng.get_state()[1] == blob[f"{tag}/final_key"]).all()
    if full:
        score, terr = s.env.getScoreAndTerritory(s.root.state)
        assert score == blob[f"{tag}/final_score"] and (terr.astype(np.int8) == blob[f"{tag}/final_terr"]).all()
        assert s.env.getWinner(s.root.state) == blob[f"{tag}/winner"]


def test_replayed_network(golden_dir):
    """Real TransGoNetwork outputs, recorded per evaluated leaf; replayed through a lookup keyed by the bit-packed
    observation, so the oracle must also reproduce every leaf observation exactly."""
    blob = _load(golden_dir, "search_replay.npz")
    table = {o.tobytes(): (p, v) for o, p, v in zip(blob["log_obs"], blob["log_policy"], blob["log_value"])}

    def lookup(obs):
        ps, vs = zip(*[table[np.packbits(o.astype(np.uint8).reshape(-1)).tobytes()] for o in obs])
        return np.stack(ps), np.stack(vs)
    replay_case(blob, "real_s11_n64", lookup)


def test_select_action_matches_reference(golden_dir):
    """Evaluation-mode search (self_play.py:689-703, temperature 0.12, no root noise, fresh tree every move), two agents
    sharing one RNG stream as in policy_evaluate (self_play.py:1013-1016)."""
    b = _load(golden_dir, "select_action.npz")
    env = OracleGoEnv()
    rng = np.random.RandomState(int(b["seed"]))
    bots = {1: OracleSearch(env, evaluators.sharp, rng, num_simulation=int(b["sims"])),
            2: OracleSearch(env, evaluators.flat, rng, num_simulation=int(b["sims"]))}
    state, _ = env.reset()
    for ply, want in enumerate(b["actions"]):
        bot = bots[env.getPlayer(state)]
        a = bot.select_action(state)
        raw = np.array([bot.root.kids[i].n if i in bot.root.kids else 0 for i in range(82)])
        assert (raw == b["counts"][ply]).all() and a == want and rng.get_state()[2] == b["pos"][ply], ply
        state, _ = env.step(state, a)


def test_policy_evaluate_loop_matches_the_reference(golden_dir):
    """Row f1: the evaluation-match loop of self_play.py:986-1040 -- colour alternation from BLACK, every move by select_action, both
    agents on ONE NumPy stream carried from game to game, winner by getWinner, win ratio -- against what the reference's own
    policy_evaluate did (tests/golden/gen_policy_evaluate.py: fake storage actors, exact stand-in evaluators): winner and train colour
    of every game, the ratio, the reference's own info strings and the position of the stream afterwards."""
    import json
    from oracle.wp_mcts import policy_evaluate
    fns = {"sharp": evaluators.sharp, "flat": evaluators.flat}
    with open(os.path.join(golden_dir, "policy_evaluate.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 5 and any(c["promoted"] for c in cases) and any(1 in c["winners"] for c in cases)
    for c in cases:
        env = OracleGoEnv(max_step=c["max_step"], komi=c["komi"])
        w, col, ratio, rng = policy_evaluate(env, fns[c["train"]], fns[c["evalu"]], c["n_games"], c["seed"], num_simulation=c["sims"],
                                             shared_stream=True, return_rng=True)
        assert list(w) == c["winners"] and list(col) == c["colours"] and ratio == c["ratio"], (c["train"], c["evalu"], c["seed"])
        assert rng.get_state()[2] == c["rng_pos"]
        assert c["info2"] == "simulate round: {},  winer is : {},  model player is : {}\n".format(c["n_games"], c["winners"][-1], c["colours"][-1])
        win = sum(int(a == b) for a, b in zip(c["winners"], c["colours"]))
        assert c["info3"] == "evaluate_score:100, win: {}, lose: {}\n".format(win, c["n_games"] - win)
        assert c["promoted"] == (["evaluate_score", "evaluate_weights"] if ratio == 1 else []) and c["score_written"] == ([200] if ratio == 1 else [])
