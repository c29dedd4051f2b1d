"""f1 fixture: the reference's OWN SelfPlay.policy_evaluate (self_play.py:986-1040) run with fake storage actors and stand-in
models (exact evaluators instead of networks), recording what it returns and does: winner and train colour of every game, the win
ratio, the two info strings, the promotion writes.  The reference draws every game of a call from ONE global NumPy stream, so
the fixture seeds that stream once per case.  Usage: python tests/golden/gen_policy_evaluate.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402
from oracle import evaluators  # noqa: E402
from gen_search import FakeModel  # noqa: E402

EVALS = {"sharp": evaluators.sharp, "flat": evaluators.flat}


class _Remote:
    def __init__(self, fn):
        self.remote = fn


class _Model(FakeModel):
    """TransGoNetwork's surface as policy_evaluate uses it: .to(device), .eval(), .set_weights(name of the evaluator)."""

    def to(self, dev):
        return self

    def eval(self):
        return self

    def set_weights(self, w):
        self.fn = EVALS[w]


def run_case(R, train, evalu, n_games, seed, sims, max_step, komi):
    import contextlib
    import io
    import re
    cfg = R.Config(); cfg.device = torch.device("cpu"); cfg.num_simulation = sims
    sp = R.self_play.SelfPlay(cfg)
    sp.env.c_init(1, 10, max_step, komi)
    sp.model = _Model(EVALS[train])
    R.self_play.TransGoNetwork = lambda c: _Model(EVALS[evalu])      # `model_evaluate = TransGoNetwork(self.config)...`
    info = {"weights": train, "evaluate_weights": evalu, "evaluate_score": 100}
    writes = []

    class Store:
        pass
    st = Store()
    st.get_info = _Remote(lambda k: info[k])
    st.set_info = _Remote(lambda k, v=None: writes.append((k, v)))
    np.random.seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ratio, info2, info3 = sp.policy_evaluate(n_games=n_games, shared_storage_worker=st)
    pos = int(np.random.get_state()[2])
    rounds = re.findall(r"simulate round: (\d+) ,  winer is : (\d) ,  model player is : (\d)", buf.getvalue())   # the reference's own per-game line
    assert [int(r[0]) for r in rounds] == list(range(1, n_games + 1))
    winners, colours = [int(r[1]) for r in rounds], [int(r[2]) for r in rounds]
    promoted = [k for k, _ in writes]
    return dict(train=train, evalu=evalu, n_games=n_games, seed=seed, sims=sims, max_step=max_step, komi=komi, winners=winners, colours=colours,
                ratio=float(ratio), info2=info2, info3=info3, promoted=promoted, score_written=[v for k, v in writes if k == "evaluate_score"],
                rng_pos=pos)


def main():
    R = ref_harness.load_reference()
    cases = [run_case(R, "sharp", "flat", 3, 300, 24, 24, 0.5), run_case(R, "flat", "sharp", 4, 77, 24, 20, 0.5),
             run_case(R, "sharp", "sharp", 2, 5, 16, 12, 7.5), run_case(R, "sharp", "flat", 2, 11, 16, 30, 0.5),
             # single games: the reference's one stream and this repo's per-game streams (seed + i) coincide; the first one is a clean
             # sweep (promotion: evaluate_score + 100, evaluate_weights <- weights, self_play.py:1035-1038)
             run_case(R, "flat", "sharp", 1, 77, 24, 20, 0.5), run_case(R, "sharp", "flat", 1, 300, 24, 24, 0.5)]
    import json
    out = os.path.join(HERE, "policy_evaluate.json")
    with open(out, "w") as f:
        json.dump({"source": "reference SelfPlay.policy_evaluate (self_play.py:986-1040) with fake storage actors and exact stand-in "
                             "evaluators in place of the two networks; one np.random.seed(seed) per case", "cases": cases}, f, indent=1)
    for c in cases:
        print(c["train"], "vs", c["evalu"], c["winners"], c["colours"], c["ratio"], repr(c["info3"]), c["promoted"], c["rng_pos"])


if __name__ == "__main__":
    main()
