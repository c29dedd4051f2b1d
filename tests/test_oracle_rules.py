"""oracle/go_oracle.c against golden vectors recorded from the compiled reference rules engine
(tests/golden/gen_fixtures.py rules): legal sets, no-eye sets, checkAction, all 10 feature planes, player, step,
score, territory and done flags for every ply of 160 random + 6 crafted games (ko, suicide, double pass, illegal
moves left unchanged, stepping a finished game)."""
import os

import numpy as np
import pytest

from oracle.go_oracle import OracleGoEnv
from tests import rules_replay

HAVE_REF = os.path.exists("/root/reference/GoEnv/cpp_src/board.cc")


def test_rules_fixture(golden_dir):
    blob = rules_replay.load(golden_dir)
    n = rules_replay.replay(OracleGoEnv(), blob)
    assert n > 15000


def test_reset_and_empty_board():
    env = OracleGoEnv()
    s, done = env.reset()
    assert not done and env.getPlayer(s) == 1 and env.getStep(s) == 1
    assert env.getScore(s) == -7.5                      # board.cc:932-935 + go_env.cc:129
    assert list(env.getLegalAction(s)) == list(range(81))
    assert env.encode(s).sum() == 0


def test_pass_only_when_no_point_is_legal():
    """environment.py:121-129: pass is reported only when it is the only move."""
    env = OracleGoEnv()
    s, _ = env.reset()
    assert 81 not in list(env.getLegalAction(s))


def test_max_step_cutoff():
    env = OracleGoEnv(max_step=5)
    s, _ = env.reset()
    for i, a in enumerate([0, 1, 2, 3]):
        s, done = env.step(s, a)
        assert not done
    s, done = env.step(s, 4)                            # step_count becomes 6 > 5 (go_env.cc:67)
    assert done and env.isTerminated(s)


@pytest.mark.skipif(not HAVE_REF, reason="reference tree only exists in the build container")
def test_against_compiled_reference_19x19_logic_is_size_generic():
    """Size-generic code paths: the same oracle binary at S=9 equals the compiled reference on fresh random games
    (not the fixture seeds)."""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "GoEnv", "go_env.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built")
    lib = ctypes.CDLL(so)
    lib.Init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    lib.Init(1, 10, 120, 7.5)
    lib.Step.restype = ctypes.c_bool
    lib.getScore.restype = ctypes.c_float
    lib.checkAction.restype = ctypes.c_bool; lib.isTerminated.restype = ctypes.c_bool
    env = OracleGoEnv()
    rng = np.random.RandomState(99)
    for g in range(40):
        a = ctypes.create_string_buffer(1200); lib.Reset(a)
        b, _ = env.reset()
        done = False
        while not done:
            buf = (ctypes.c_int * 82)()
            n = lib.getLegalAction(a, buf)
            la = list(buf[:n]); la = la if n == 1 else la[:-1]
            assert la == [int(x) for x in env.getLegalAction(b)]
            e = np.zeros(810, np.float32); lib.Encode(a, e.ctypes.data_as(ctypes.c_void_p))
            assert (e == env.encode(b).reshape(-1)).all()
            assert lib.getScore(a) == env.getScore(b)
            if env.getStep(b) % 16 == 1:                 # the reference prints a line for every refusal: sample, do not flood
                for c in (81, -1, -2):                   # S*S is NOT translated to PASS here (only Step does that); -1 / -2 are
                    assert bool(lib.checkAction(a, ctypes.c_int(c))) == bool(env.checkAction(b, c)), c
            assert bool(lib.isTerminated(a)) == bool(env.isTerminated(b))
            assert (lib.getPlayer(a) & 0xFF) == env.getPlayer(b) and lib.getStep(a) == env.getStep(b)
            act = la[rng.randint(len(la))] if rng.rand() > 0.04 else 81
            a2 = ctypes.create_string_buffer(1200)
            d1 = lib.Step(a, a2, ctypes.c_int(act)); a = a2
            b, d2 = env.step(b, act)
            assert bool(d1) == d2
            done = d2


def test_rules_fixture_19x19(golden_dir):
    """19x19 games recorded from a 19x19 build of the reference engine (tests/golden/gen_rules19.py)."""
    blob = rules_replay.load(golden_dir, "rules_s19.npz")
    n = rules_replay.replay(OracleGoEnv(board_size=19, max_step=int(blob["max_step"])), blob)
    assert n > 1500


def test_encode_9_and_13_planes_match_the_reference_engine(golden_dir):
    from tests import rules_replay
    blob = rules_replay.load(golden_dir, "rules_enc_variants_s9.npz")
    n = rules_replay.replay_encode_variants(lambda d: OracleGoEnv(encode_dim=d), blob)
    assert n > 2000
