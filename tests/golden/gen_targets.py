"""F4 targets fixture: run the reference's SelfPlay.continuous_self_play (self_play.py:902-983) for ONE short game with
fake storage actors and a stand-in model, recording every tuple it appends (8 symmetries x positions, in order)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import evaluators  # noqa: E402
from gen_search import FakeModel  # noqa: E402


class _Stop(Exception):
    pass


class _Remote:
    def __init__(self, fn):
        self.remote = fn


def run(R, outdir):
    cfg = R.Config(); cfg.device = torch.device("cpu"); cfg.num_simulation = 12
    sp = R.self_play.SelfPlay(cfg)
    sp.env.c_init(1, 10, 24, 7.5)                      # stop the game after 24 plies (Init, go_env.cc:21-32)
    fm = FakeModel(evaluators.sharp)
    fm.set_weights = lambda w: None
    sp.model = fm
    appended = []
    info = {"weights": None}

    class Store:
        pass
    st = Store()
    st.get_info = _Remote(lambda k: info.get(k))

    def set_info(k, v=None):
        if k == "now_play_games":
            raise _Stop()
    st.set_info = _Remote(set_info)
    mem = Store()
    mem.append = _Remote(lambda o, p, z, w: appended.append((np.array(o), np.array(p), float(z), np.array(w))))
    np.random.seed(21)
    try:
        sp.continuous_self_play(st, mem)
    except _Stop:
        pass
    blob = dict(obs=np.stack([a[0] for a in appended]).astype(np.float32), pi=np.stack([a[1] for a in appended]),
                z=np.array([a[2] for a in appended]), own=np.stack([a[3] for a in appended]),
                seed=np.int64(21), sims=np.int64(12), max_step=np.int64(24))
    np.savez_compressed(os.path.join(outdir, "targets_game.npz"), **blob)
    print("targets tuples", len(appended), "positions", len(appended) // 8)
