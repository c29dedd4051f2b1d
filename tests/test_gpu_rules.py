"""HIP rules engine (transgo_amd.environment.GoEnv -> libtransgo_hip.so) against the golden vectors recorded from the
compiled reference, and against the CPU oracle on fresh batched random games at 9x9 and 19x19.  Bit-exact."""
import numpy as np
import pytest

from tests import rules_replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def henv():
    from transgo_amd.environment import GoEnv
    return GoEnv()


def test_rules_fixture_gpu(golden_dir, henv):
    blob = rules_replay.load(golden_dir)
    n = rules_replay.replay(henv, blob)
    assert n > 15000


def _lockstep(S, n_games, max_step, seed, pass_p):
    from oracle.go_oracle import OracleGoEnv
    from transgo_amd.environment import GoEnv
    henv = GoEnv(board_size=S) if max_step is None else None
    class C: pass
    c = C(); c.board_size = S; c.max_step = max_step; c.komi = 7.5; c.encode_state_channels = 10
    henv = GoEnv(c)
    oenv = OracleGoEnv(c)
    P = S * S
    rng = np.random.RandomState(seed)
    hs = henv.reset_batch(n_games)
    os_ = [oenv.reset()[0] for _ in range(n_games)]
    alive = np.ones(n_games, bool)
    plies = 0
    while alive.any():
        q = henv.query_batch(hs, legal=True, noeye=True, obs=True, score=True, terr=True, meta=True)
        acts = np.zeros(n_games, np.int32)
        for g in range(n_games):
            o = os_[g]
            assert q["terminated"][g] == (not alive[g])
            legal = np.zeros(P + 1, np.uint8)
            la = np.asarray(oenv.getLegalAction(o), dtype=np.int64)
            legal[la] = 1; legal[P] = 1
            assert (q["legal"][g] == legal).all(), ("legal", S, g, plies)
            ne = np.zeros(P + 1, np.uint8); ne[np.asarray(oenv.getLegalNoEye(o), dtype=np.int64)] = 1
            assert (q["noeye"][g] == ne).all(), ("noeye", S, g, plies)
            assert (q["obs"][g] == oenv.encode(o)).all(), ("obs", S, g, plies)
            sc, te = oenv.getScoreAndTerritory(o)
            assert q["score"][g] == np.float32(sc) and (q["terr"][g] == te).all(), ("score", S, g, plies)
            assert q["player"][g] == oenv.getPlayer(o) and q["step"][g] == oenv.getStep(o)
            if g == 0 and plies % 8 == 0:                # single-state wrapper incl. the pass index (off the board for checkAction)
                st0 = hs[0].tobytes()
                for c in (P, -1, -2, int(la[0]), P + 3):
                    assert henv.checkAction(st0, c) == oenv.checkAction(o, c), ("checkAction", c)
            r = rng.rand()
            if r < pass_p:
                acts[g] = P
            elif r < pass_p + 0.03:
                acts[g] = rng.randint(P)                 # possibly illegal
            else:
                acts[g] = la[rng.randint(len(la))]
        hs, done, ok = henv.step_batch(hs, acts)
        for g in range(n_games):
            os_[g], d = oenv.step(os_[g], int(acts[g]))
            assert d == done[g], ("done", S, g, plies)
            if alive[g]:
                alive[g] = not d
        plies += 1
        assert plies < 2000
    return plies


def test_batch_vs_oracle_9x9():
    assert _lockstep(9, 96, 120, 5, 0.02) >= 100


def test_batch_vs_oracle_19x19():
    assert _lockstep(19, 24, 220, 6, 0.01) >= 200


def test_no_gpu_fallback_symbols():
    """The library is the only implementation: GoEnv has no Python/CPU rules code to fall back to."""
    import inspect
    from transgo_amd import environment
    src = inspect.getsource(environment)
    assert "oracle" not in src.replace("no CPU", "")


def test_reference_ctypes_binding_works_unchanged(golden_dir):
    """Bind the library exactly the way GoEnv/environment.py:42-90 does (same restype/argtypes, a 1196-byte c_GoState-sized
    buffer per state) and replay crafted + random fixture games through Init/Reset/Step/getLegalAction/Encode/getTerritory."""
    import ctypes
    from numpy.ctypeslib import ndpointer
    from transgo_amd import _lib
    lib = ctypes.cdll.LoadLibrary(_lib.LIB_PATH)
    State = ctypes.c_char * 1196
    lib.Init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    lib.Init(1, 10, 120, 7.5)
    lib.Step.argtypes = [ctypes.POINTER(State), ctypes.POINTER(State), ctypes.c_int]; lib.Step.restype = ctypes.c_bool
    lib.Reset.argtypes = [ctypes.POINTER(State)]
    lib.Encode.argtypes = [ctypes.POINTER(State), ndpointer(ctypes.c_float)]
    lib.getTerritory.argtypes = [ctypes.POINTER(State), ndpointer(ctypes.c_float)]; lib.getTerritory.restype = ctypes.c_float
    lib.getLegalAction.argtypes = [ctypes.POINTER(State), ndpointer(ctypes.c_int)]; lib.getLegalAction.restype = ctypes.c_int
    lib.getPlayer.argtypes = [ctypes.POINTER(State)]; lib.getPlayer.restype = ctypes.c_int
    lib.getStep.argtypes = [ctypes.POINTER(State)]; lib.getStep.restype = ctypes.c_int

    class Env:                                            # the method bodies of environment.py:92-173, verbatim in spirit
        def reset(self):
            s = State(); lib.Reset(s); return s, False
        def step(self, state, action):
            n = State(); done = lib.Step(state, n, action); return n, done
        def encode(self, state):
            e = np.zeros([10, 9, 9], dtype="float32"); lib.Encode(state, e); return e
        def getLegalAction(self, state):
            buf = np.zeros([82], dtype="int32"); n = lib.getLegalAction(state, buf); acts = buf[:n]
            return acts if n == 1 else [a for a in acts if a != 81]
        def getLegalNoEye(self, state):
            return henv_noeye(state)
        def getPlayer(self, state): return lib.getPlayer(state)
        def getStep(self, state): return lib.getStep(state)
        def getScoreAndTerritory(self, state):
            t = np.zeros([81], dtype="float32"); s = lib.getTerritory(state, t); return s, t
        def getScore(self, state): return self.getScoreAndTerritory(state)[0]
        def checkAction(self, state, a): return bool(lib.checkAction(state, ctypes.c_int(a)))

    lib.getLegalNoEye.argtypes = [ctypes.POINTER(State), ndpointer(ctypes.c_int)]; lib.getLegalNoEye.restype = ctypes.c_int
    lib.checkAction.argtypes = [ctypes.POINTER(State), ctypes.c_int]; lib.checkAction.restype = ctypes.c_bool

    def henv_noeye(state):
        buf = np.zeros([82], dtype="int32"); n = lib.getLegalNoEye(state, buf); return buf[:n]
    blob = rules_replay.load(golden_dir)
    crafted = set(range(int(blob["n_random"]), int(blob["n_random"]) + len(blob["crafted_names"])))
    n = rules_replay.replay(Env(), blob, games=crafted | {0, 1, 2})
    assert n > 300


def test_rules_fixture_19x19_gpu(golden_dir):
    from transgo_amd.environment import GoEnv
    blob = rules_replay.load(golden_dir, "rules_s19.npz")

    class C: pass
    c = C(); c.board_size = 19; c.max_step = int(blob["max_step"]); c.komi = 7.5; c.encode_state_channels = 10
    n = rules_replay.replay(GoEnv(c), blob)
    assert n > 1500


def test_encode_9_and_13_planes_match_the_reference_engine_gpu(golden_dir):
    """The two feature encodings the reference's Python never selects (encode9 / encode13, board_feature.cc:213-253)."""
    from transgo_amd.environment import GoEnv

    def make(d):
        class C: pass
        c = C(); c.board_size = 9; c.max_step = 120; c.komi = 7.5; c.encode_state_channels = d
        return GoEnv(c)
    blob = rules_replay.load(golden_dir, "rules_enc_variants_s9.npz")
    assert rules_replay.replay_encode_variants(make, blob) > 2000


def test_compat_abi_in_lockstep_with_the_compiled_reference():
    """Every one of the 15 go_env.h symbols, called with identical arguments on this library and on oracle/_ref (the reference
    engine compiled in place; a built artefact that travels with the snapshot): return values and output buffers must agree call
    by call over seeded random games, including refused moves, the in-place Step_, steps on finished games, checkAction on every
    coordinate from RESIGN to S*S, and getSubEncode."""
    import ctypes, os
    from transgo_amd import _lib
    ref_so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "GoEnv", "go_env.so")
    if not os.path.exists(ref_so):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    libs = [ctypes.CDLL(ref_so), ctypes.CDLL(_lib.LIB_PATH)]
    from transgo_amd.environment import GoEnv
    mirror = GoEnv()                                         # the Python mirror's subEncode (environment.py:110-113) rides along
    for L in libs:
        L.Init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        for f in ("Step", "Step_", "checkAction", "isTerminated", "Reset", "Encode"):
            getattr(L, f).restype = ctypes.c_bool
        L.getScore.restype = ctypes.c_float; L.getTerritory.restype = ctypes.c_float
        L.Init(1, 10, 60, 7.5)
    P = 81
    rng = np.random.RandomState(515)
    new = lambda: ctypes.create_string_buffer(2048)
    calls = 0
    for g in range(12):
        st = [new(), new()]
        assert libs[0].Reset(st[0]) == libs[1].Reset(st[1])
        done = False; extra = 2
        while extra:
            outs = []
            for L, s in zip(libs, st):
                la = (ctypes.c_int * (P + 1))(); n1 = L.getLegalAction(s, la)
                ne = (ctypes.c_int * (P + 1))(); n2 = L.getLegalNoEye(s, ne)
                enc = np.zeros(10 * P, np.float32); L.Encode(s, enc.ctypes.data_as(ctypes.c_void_p))
                ter = np.zeros(P, np.float32); sc2 = L.getTerritory(s, ter.ctypes.data_as(ctypes.c_void_p))
                sub = np.zeros(5 * 10 * 49, np.float32)
                L.getSubEncode(enc.ctypes.data_as(ctypes.c_void_p), sub.ctypes.data_as(ctypes.c_void_p), 7, 10, 5)
                chk = [bool(L.checkAction(s, ctypes.c_int(c))) for c in (-2, -1, P, 0, 13, 40, 67, 80)]
                outs.append((list(la[:n1]), list(ne[:n2]), enc, float(L.getScore(s)), float(sc2), ter, sub, chk,
                             L.getPlayer(s) & 0xFF, L.getStep(s), bool(L.isTerminated(s))))
            a, b = outs
            assert np.array_equal(mirror.subEncode(a[2].reshape(10, 9, 9)).reshape(-1), a[6][:4 * 10 * 49]), (g, "GoEnv.subEncode")
            for i, (x, y) in enumerate(zip(a, b)):
                assert np.array_equal(np.asarray(x), np.asarray(y)), (g, "field", i)
            calls += 1
            legal = [c for c in a[0] if c != P] or [P]
            r = rng.rand()
            act = P if r < 0.05 else int(rng.randint(P)) if r < 0.10 else int(legal[rng.randint(len(legal))])
            if rng.rand() < 0.5:
                nx = [new(), new()]
                d = [bool(L.Step(s, n, ctypes.c_int(act))) for L, s, n in zip(libs, st, nx)]
                st = nx
            else:
                d = [bool(L.Step_(s, ctypes.c_int(act))) for L, s in zip(libs, st)]
            assert d[0] == d[1], (g, "done")
            if done:
                extra -= 1                                   # two more calls on a finished game
            done = done or d[0]
    assert calls > 300
