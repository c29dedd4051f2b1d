"""F2 search fixtures: drive the imported reference WP_MCTS (self_play.py:575-875) move by move, exactly as
SelfPlay.continuous_self_play does (self_play.py:910-929), with a stand-in model, and record per move the inherited
root visits, raw child visit counts, chosen action, pi and the MT19937 stream position."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import evaluators  # noqa: E402


class FakeModel:
    """Quacks like TransGoNetwork for WP_MCTS.policyValueFn (self_play.py:777-786)."""

    def __init__(self, fn, log=None):
        self.fn, self.log = fn, log
        self._p = torch.nn.Parameter(torch.zeros(1))

    def parameters(self):
        return iter([self._p])

    def main_prediction(self, x):
        obs = x.numpy()
        p, v = self.fn(obs)
        if self.log is not None:
            for o, pp, vv in zip(obs, p, v):
                self.log.append((np.packbits(o.astype(np.uint8).reshape(-1)), pp.copy(), vv.copy()))
        return torch.from_numpy(p), torch.from_numpy(v), torch.zeros(len(p), 81)


def play(R, fn, seed, sims, max_moves, log=None):
    cfg = R.Config(); cfg.device = torch.device("cpu"); cfg.num_simulation = sims
    env = R.environment.GoEnv(cfg)
    np.random.seed(seed)
    agent = R.self_play.WP_MCTS(cfg, env, FakeModel(fn, log))
    rec = dict(n0=[], counts=[], action=[], pi=[], pos=[], done=[], player=[], step=[], root_n=[])
    for _ in range(max_moves):
        n0 = agent.root.total_visit_count
        a, pi, obs = agent.get_action_probs()
        counts = np.array([agent.root.visit_count(i) for i in range(82)], np.int32)
        rec["n0"].append(n0); rec["counts"].append(counts); rec["action"].append(int(a)); rec["pi"].append(pi)
        rec["root_n"].append(agent.root.total_visit_count)
        rec["player"].append(env.getPlayer(agent.root.state)); rec["step"].append(env.getStep(agent.root.state))
        done = agent.update_with_action(a)
        rec["pos"].append(int(np.random.get_state()[2])); rec["done"].append(int(done))
        if done:
            break
    final = agent.root.state
    score, terr = env.getScoreAndTerritory(final)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["final_score"] = np.float32(score); out["final_terr"] = terr.astype(np.int8)
    out["winner"] = np.int32(env.getWinner(final))
    out["final_key"] = np.random.get_state()[1].astype(np.uint32)
    return out


def run(R, outdir):
    torch.set_num_threads(1)
    cases = [("flat", 0, 64, 200), ("sharp", 1, 64, 200), ("sharp", 2, 210, 40), ("flat", 3, 400, 10),
             ("sharp", 4, 400, 12), ("flat", 5, 16, 200), ("sharp", 6, 8, 200)]
    blob = {}
    for name, seed, sims, mm in cases:
        r = play(R, evaluators.BY_NAME[name], seed, sims, mm)
        tag = f"{name}_s{seed}_n{sims}"
        for k, v in r.items():
            blob[f"{tag}/{k}"] = v
        print(tag, "moves", len(r["action"]), "done", r["done"][-1], "sum counts", r["counts"].sum())
    blob["cases"] = np.array([f"{n}_s{s}_n{k}" for n, s, k, _ in cases])
    np.savez_compressed(os.path.join(outdir, "search_analytic.npz"), **blob)

    # real network (reference TransGoNetwork, seeded), evaluator outputs recorded for replay
    torch.manual_seed(7)
    cfg = R.Config(); cfg.device = torch.device("cpu")
    net = R.model.TransGoNetwork(cfg).eval()
    with torch.no_grad():
        for m in net.modules():                      # non-trivial BN statistics and attention gain
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            if hasattr(m, "gamma"):
                m.gamma.fill_(0.3)
    log = []

    def real(obs):
        with torch.no_grad():
            p, v, _ = net.main_prediction(torch.from_numpy(obs))
        return p.numpy(), v.numpy()
    r = play(R, real, 11, 64, 4, log)
    blob = {f"real_s11_n64/{k}": v for k, v in r.items()}
    blob["log_obs"] = np.stack([l[0] for l in log]); blob["log_policy"] = np.stack([l[1] for l in log])
    blob["log_value"] = np.stack([l[2] for l in log])
    np.savez_compressed(os.path.join(outdir, "search_replay.npz"), **blob)
    print("replay leaves", len(log))
