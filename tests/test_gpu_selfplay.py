"""Whole hot path on the GPU (search + move selection + re-rooting + game end + targets + augmentation) against the tuples
the reference's continuous_self_play appended for the same seed (tests/golden/targets_game.npz)."""
import os

import numpy as np
import pytest

from oracle import evaluators

pytestmark = pytest.mark.gpu


def test_selfplay_game_reproduces_reference_appends(golden_dir):
    from transgo_amd.configure import Config
    from transgo_amd.self_play import BatchedSelfPlay
    with np.load(os.path.join(golden_dir, "targets_game.npz")) as z:
        b = {k: z[k] for k in z.files}
    cfg = Config(num_simulation=int(b["sims"]), max_step=int(b["max_step"]))
    sp = BatchedSelfPlay(cfg, 3, evaluator=evaluators.sharp, seed_fn=lambda g, k: int(b["seed"]) + 1000 * g + 7 * k)
    finished = []
    for _ in range(int(b["max_step"])):
        finished += sp.step()
    assert len(finished) == 3 and sp.games_finished == 3
    rec = [r for r in finished if r.seed == int(b["seed"])][0]
    tup = sp.targets(rec)
    assert len(tup) == len(b["z"])
    for i, (o, p, zz, w) in enumerate(tup):
        assert (o == b["obs"][i]).all() and (p == b["pi"][i]).all() and zz == b["z"][i] and (w == b["own"][i]).all(), i
    # the slots restarted with fresh seeds and keep playing
    sp.step()
    assert sp.engine.stats()["errors"] == 0


def test_continuous_self_play_feeds_storage():
    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import ReplayMemory_Random
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    cfg = Config(num_simulation=8, max_step=6, num_features=32, num_blocks=2, buffer_size=4096)
    st = SharedStorage({"weights": model.random_weights(9, 10, 32, 2), "now_play_steps": 0, "now_play_games": 0,
                        "now_train_steps": 10 ** 9, "train_play_ratio": 0.075, "adjust_train_play_ratio": True,
                        "game_total_num": 1e8, "adjust_lr": False, "learn_rate": 1e-4}, cfg)
    mem = ReplayMemory_Random(cfg)
    SelfPlay(cfg, n_games=5).continuous_self_play(st, mem, max_moves=6)
    assert st.get_info("now_play_games") == 5 and st.get_info("now_play_steps") == 30
    assert mem.info()["index"] == 5 * 6 * 8
    s, p, z, o = map(np.stack, zip(*mem.sample(32)))     # trainer.py:49
    assert s.shape == (32, 10, 9, 9) and p.shape == (32, 82) and set(np.unique(z)) <= {-1.0, 1.0} and o.shape == (32, 81)
