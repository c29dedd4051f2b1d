"""Golden-vector generator: runs the REFERENCE (compiled go_env.so + imported self_play.py/model.py) in the build
container and writes small data fixtures next to this file.  Usage:  python tests/golden/gen_fixtures.py [rules|rng|search|net|targets|all]

Fixtures are data only (inputs + expected outputs).  No reference source text is stored.  The GPU box never runs
this script (it has no /root/reference); it only reads the committed .npz files.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

P9 = 81


def pack_bits(a):
    return np.packbits(np.asarray(a, dtype=np.uint8).reshape(-1))


# ---------------------------------------------------------------------------------------------------------------------
# F1 rules: seeded random games through the reference env, every observable per ply.
# ---------------------------------------------------------------------------------------------------------------------
def crafted_sequences():
    """Hand-built move lists (action indices, 81 = pass) that force the quirks listed in SURVEY.md §8a."""
    def xy(x, y):
        return y * 9 + x
    seqs = {}
    # textbook ko in the centre, then immediate retake attempt (illegal), pass, retake
    seqs["ko_basic"] = [xy(3, 4), xy(4, 4), xy(4, 3), xy(5, 3), xy(4, 5), xy(5, 5), xy(0, 0), xy(6, 4),
                        xy(5, 4), xy(4, 4), 81, xy(8, 8), xy(4, 4), xy(5, 4), 81, xy(0, 8), xy(5, 4)]
    # corner suicide attempt, then filling one's own eye
    seqs["corner"] = [xy(1, 0), xy(8, 8), xy(0, 1), xy(0, 0), xy(7, 7), xy(0, 0), 81, xy(0, 0)]
    # double pass ends the game; the generator then records one more step on the finished game
    seqs["double_pass"] = [xy(2, 2), 81, 81]
    # pass / pass at the very start (step_count > 1 guard of board.cc:658) and an occupied-point move
    seqs["early_pass"] = [81, xy(1, 1), xy(1, 1), 81, 81]
    # snapback-ish multi stone capture where the capturing stone has no liberties before capture
    seqs["multi_capture"] = [xy(0, 1), xy(0, 0), xy(1, 1), xy(1, 0), xy(2, 0), xy(8, 8), xy(7, 7), xy(8, 7),
                             xy(7, 8), xy(0, 0), xy(8, 8)]
    # merge of four groups with one stone
    seqs["merge4"] = [xy(4, 3), xy(0, 0), xy(4, 5), xy(8, 0), xy(3, 4), xy(0, 8), xy(5, 4), xy(8, 8), xy(4, 4),
                      xy(1, 0), xy(0, 1)]
    return seqs


def gen_rules(R, n_games=160, seed=20240917):
    env = R.environment.GoEnv(R.cfg)
    rng = np.random.RandomState(seed)
    rec = {k: [] for k in ("game", "action", "done", "player", "step", "legal", "obs", "score", "terr", "noeye",
                           "check")}
    games = []
    crafted = crafted_sequences()
    names = list(crafted.keys())
    for g in range(n_games + len(names)):
        state, _ = env.reset()
        pass_p = [0.0, 0.02, 0.10][g % 3]
        scripted = crafted[names[g - n_games]] if g >= n_games else None
        ply = 0
        done = False
        while True:
            legal = np.asarray(env.getLegalAction(state), dtype=np.int64)
            mask = np.zeros(P9 + 1, np.uint8); mask[legal] = 1
            noeye = np.zeros(P9 + 1, np.uint8); noeye[np.asarray(env.getLegalNoEye(state))] = 1
            score, terr = env.getScoreAndTerritory(state)
            chk = np.array([env.checkAction(state, a) for a in range(P9)], np.uint8)
            rec["game"].append(g)
            rec["player"].append(env.getPlayer(state)); rec["step"].append(env.getStep(state))
            rec["legal"].append(pack_bits(mask)); rec["noeye"].append(pack_bits(noeye))
            rec["obs"].append(pack_bits(env.encode(state)))
            rec["score"].append(score); rec["terr"].append(terr.astype(np.int8))
            rec["check"].append(pack_bits(chk))
            if done:
                # one extra step on a finished game: must return done and leave the state unchanged
                rec["action"].append(0); rec["done"].append(1)
                break
            if scripted is not None:
                if ply >= len(scripted):
                    rec["action"].append(-9); rec["done"].append(0)
                    break
                act = scripted[ply]
            else:
                r = rng.rand()
                if r < pass_p:
                    act = P9
                elif r < pass_p + 0.02:
                    act = int(rng.randint(P9))          # unfiltered point: often illegal (occupied / suicide / ko)
                else:
                    act = int(legal[rng.randint(len(legal))])
            state, done = env.step(state, act)
            rec["action"].append(act); rec["done"].append(int(done))
            ply += 1
        games.append(ply)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["score"] = out["score"].astype(np.float32)
    out["crafted_names"] = np.array(names)
    out["n_random"] = np.int64(n_games)
    np.savez_compressed(os.path.join(HERE, "rules_s9.npz"), **out)
    print("rules_s9: games", len(games), "records", len(out["action"]),
          "bytes", os.path.getsize(os.path.join(HERE, "rules_s9.npz")))


# ---------------------------------------------------------------------------------------------------------------------
# F5 RNG: NumPy legacy RandomState draw sequences the search and the move selection depend on.
# ---------------------------------------------------------------------------------------------------------------------
def gen_rng():
    out = {}
    for seed in (0, 1, 12345, 2**31 - 1):
        rs = np.random.RandomState(seed)
        seq = []
        # interleave the three draw kinds exactly as self_play.py does: dirichlet (90-95), choice(list) (709-713),
        # choice(A, p) (683)
        for n in (81, 82, 40, 3, 1, 79):
            d = rs.dirichlet([0.03] * n)
            seq.append(("dir", n, d))
            for k in (1, 2, 3, 5, 7, 64, 65, 81, 82):
                c = rs.choice(list(range(100, 100 + k)))
                seq.append(("tie", k, np.array([c - 100], np.float64)))
            p = rs.dirichlet([0.5] * 82)
            u = rs.choice(np.arange(82), p=p)
            seq.append(("pick", 82, np.concatenate([p, [u]])))
        out[f"s{seed}_kinds"] = np.array([{"dir": 0, "tie": 1, "pick": 2}[s[0]] for s in seq], np.int32)
        out[f"s{seed}_n"] = np.array([s[1] for s in seq], np.int32)
        out[f"s{seed}_vals"] = np.concatenate([s[2] for s in seq])
        out[f"s{seed}_final_pos"] = np.int64(rs.get_state()[2])
        out[f"s{seed}_final_key"] = rs.get_state()[1].astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "rng_mt19937.npz"), **out)
    print("rng_mt19937 written")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    R = ref_harness.load_reference()
    if what in ("rules", "all"):
        gen_rules(R)
    if what in ("rng", "all"):
        gen_rng()
    if what in ("search", "all"):
        import gen_search
        gen_search.run(R, HERE)
    if what in ("net", "all"):
        import gen_net
        gen_net.run(R, HERE)
    if what in ("targets", "all"):
        import gen_targets
        gen_targets.run(R, HERE)


if __name__ == "__main__":
    main()
