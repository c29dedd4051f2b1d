"""19x19 search fixture: the imported reference WP_MCTS (self_play.py:575-875) driven at board size 19.

The reference fixes the board size in three places: the C++ engine (go_comm.h:20, compile time), the ctypes mirror and
buffer sizes in GoEnv/environment.py (module constants BOARD_SIZE / MAX_COORD / MAX_BLOCK) and Config.board_size.  This
script (build container only) builds the 19x19 engine in a scratch directory outside the repository exactly as
gen_rules19.py does, imports the reference modules unmodified, and adjusts the LOADED module objects: BOARD_SIZE = 19, an
opaque 32-KB c_GoState (the Python side never looks inside the state), env.board_size / cfg.board_size = 19 and a second
Init() call with the ply limit.  Nothing of the reference is copied; only the recorded vectors are committed
(tests/golden/search_s19.npz)."""
import ctypes
import os
import shutil
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_rules19  # noqa: E402
from gen_search import FakeModel  # noqa: E402
from oracle import evaluators  # noqa: E402

S, P, A = 19, 361, 362
REF = "/root/reference"


def load_reference_19(max_step):
    tmp, so = gen_rules19.build()
    os.makedirs(os.path.join(tmp, "GoEnv"))
    shutil.move(so, os.path.join(tmp, "GoEnv", "go_env.so"))          # environment.py:42 loads ./GoEnv/go_env.so
    sys.dont_write_bytecode = True
    ray = types.ModuleType("ray"); ray.remote = lambda c: c; ray.get = lambda x: x
    sys.modules.setdefault("ray", ray)
    sys.path.insert(0, REF)
    os.chdir(tmp)
    import configure
    import self_play
    from GoEnv import environment

    class OpaqueState(ctypes.Structure):
        _fields_ = [("raw", ctypes.c_char * 32768)]
    environment.BOARD_SIZE = S
    environment.c_GoState = OpaqueState
    cfg = configure.Config(); cfg.device = torch.device("cpu"); cfg.board_size = S
    env = environment.GoEnv(cfg)
    env.board_size = S
    env.c_init(1, 10, max_step, 7.5)
    return tmp, cfg, env, self_play


def play(cfg, env, self_play, fn, seed, sims, max_moves):
    cfg.num_simulation = sims
    np.random.seed(seed)
    agent = self_play.WP_MCTS(cfg, env, FakeModel(fn))
    rec = dict(n0=[], counts=[], action=[], pi=[], pos=[], done=[], player=[], step=[], root_n=[])
    for _ in range(max_moves):
        n0 = agent.root.total_visit_count
        a, pi, obs = agent.get_action_probs()
        rec["n0"].append(n0); rec["counts"].append(np.array([agent.root.visit_count(i) for i in range(A)], np.int32))
        rec["action"].append(int(a)); rec["pi"].append(pi); rec["root_n"].append(agent.root.total_visit_count)
        rec["player"].append(env.getPlayer(agent.root.state)); rec["step"].append(env.getStep(agent.root.state))
        done = agent.update_with_action(a)
        rec["pos"].append(int(np.random.get_state()[2])); rec["done"].append(int(done))
        if done:
            break
    score, terr = env.getScoreAndTerritory(agent.root.state)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["final_score"] = np.float32(score); out["final_terr"] = terr.astype(np.int8)
    out["winner"] = np.int32(env.getWinner(agent.root.state))
    out["final_key"] = np.random.get_state()[1].astype(np.uint32)
    return out


def main():
    torch.set_num_threads(1)
    max_step = 40                                              # short games: the ply limit ends them (go_env.cc:67)
    tmp, cfg, env, self_play = load_reference_19(max_step)
    try:
        cases = [("sharp", 21, 48, 60), ("flat", 22, 24, 60), ("sharp", 23, 160, 6), ("sharp", 24, 800, 2),   # BASELINE configs[3] search depth
                 ("sharp", 25, 1600, 2)]  # BASELINE configs[4] search depth (1600 simulations/move)
        blob = {}
        for name, seed, sims, mm in cases:
            r = play(cfg, env, self_play, evaluators.BY_NAME[name], seed, sims, mm)
            tag = f"{name}_s{seed}_n{sims}"
            for k, v in r.items():
                blob[f"{tag}/{k}"] = v
            print(tag, "moves", len(r["action"]), "done", r["done"][-1], "sum counts", r["counts"].sum())
        blob["cases"] = np.array([f"{n}_s{s}_n{k}" for n, s, k, _ in cases])
        blob["size"] = np.int64(S); blob["max_step"] = np.int64(max_step)
        np.savez_compressed(os.path.join(HERE, "search_s19.npz"), **blob)
        print("bytes", os.path.getsize(os.path.join(HERE, "search_s19.npz")))
    finally:
        os.chdir(HERE)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
