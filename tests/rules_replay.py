"""Shared driver: replay tests/golden/rules_s9.npz through any env with the reference GoEnv surface and compare every
recorded observable.  Used for the CPU oracle (test_oracle_rules.py) and for the HIP engine (test_gpu_rules.py)."""
import os

import numpy as np

P9 = 81


def unpack(bits, n):
    return np.unpackbits(bits)[:n]


def load(golden_dir, name="rules_s9.npz"):
    with np.load(os.path.join(golden_dir, name)) as z:      # NpzFile decompresses on every [] access
        return {k: z[k] for k in z.files}


def replay(env, blob, games=None):
    S = int(blob["size"]) if "size" in blob else 9
    P9 = S * S
    game = blob["game"]
    n_rec = len(game)
    checked = 0
    i = 0
    while i < n_rec:
        g = game[i]
        j = i
        while j < n_rec and game[j] == g:
            j += 1
        if games is not None and g not in games:
            i = j
            continue
        state, _ = env.reset()
        for r in range(i, j):
            legal = np.zeros(P9 + 1, np.uint8); legal[np.asarray(env.getLegalAction(state), dtype=np.int64)] = 1
            assert (legal == unpack(blob["legal"][r], P9 + 1)).all(), ("legal", g, r - i)
            noeye = np.zeros(P9 + 1, np.uint8); noeye[np.asarray(env.getLegalNoEye(state), dtype=np.int64)] = 1
            assert (noeye == unpack(blob["noeye"][r], P9 + 1)).all(), ("noeye", g, r - i)
            obs = env.encode(state)
            assert obs.dtype == np.float32 and obs.shape == (10, S, S)
            assert (obs.reshape(-1).astype(np.uint8) == unpack(blob["obs"][r], 10 * P9)).all(), ("obs", g, r - i)
            assert env.getPlayer(state) == blob["player"][r] and env.getStep(state) == blob["step"][r]
            score, terr = env.getScoreAndTerritory(state)
            assert np.float32(score) == blob["score"][r], ("score", g, r - i)
            assert (terr.astype(np.int8) == blob["terr"][r]).all(), ("terr", g, r - i)
            assert np.float32(env.getScore(state)) == blob["score"][r]
            if hasattr(env, "checkActionAll"):
                chk = np.asarray(env.checkActionAll(state), np.uint8)
                a0 = (r * 7) % P9
                assert bool(env.checkAction(state, a0)) == bool(chk[a0])
            else:
                chk = np.array([env.checkAction(state, a) for a in range(P9)], np.uint8)
            assert (chk == unpack(blob["check"][r], P9)).all(), ("check", g, r - i)
            act = int(blob["action"][r])
            if act == -9:
                break
            state, done = env.step(state, act)
            assert int(done) == blob["done"][r], ("done", g, r - i)
            checked += 1
        i = j
    return checked


def replay_encode_variants(make_env, blob):
    """tests/golden/rules_enc_variants_s9.npz: the 9- and 13-plane encodings (go_env.cc:96-115) recorded from the compiled
    reference engine at every 4th position of seeded random games.  `make_env(encode_dim)` builds an env with that encoding."""
    envs = {9: make_env(9), 13: make_env(13)}
    checked = 0
    for g in range(len(blob["action"])):
        rows = np.flatnonzero(blob["game"] == g)
        plies = {int(blob["ply"][r]): r for r in rows}
        states = {d: e.reset()[0] for d, e in envs.items()}
        for ply, act in enumerate(blob["action"][g]):
            if act < 0:
                break
            if ply in plies:
                r = plies[ply]
                for d, key in ((9, "obs9"), (13, "obs13")):
                    obs = envs[d].encode(states[d])
                    assert obs.shape == (d, 9, 9) and obs.dtype == np.float32
                    assert (obs.reshape(-1).astype(np.uint8) == unpack(blob[key][r], d * 81)).all(), (d, g, ply)
                    checked += 1
            for d, e in envs.items():
                states[d], _ = e.step(states[d], int(act))
    return checked
