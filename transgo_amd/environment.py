"""GoEnv -- the reference's environment interface (GoEnv/environment.py:32-198) backed by the HIP rules engine.

Same method names, argument meaning and return types as the reference class, so self-play / evaluation code written
against it runs unchanged.  A state is an opaque bytes-like blob (48 B at 9x9) owned by the caller, exactly as the
reference's `c_GoState` is (environment.py:93-103).  Every call runs the gfx950 kernels of transgo_amd/csrc (batch
of one); the `*_batch` methods are the same kernels over n states and are what throughput-sensitive callers use.
There is no CPU implementation behind this class.
"""
import ctypes

import numpy as np

from . import _lib

BLACK, WHITE = 1, 2


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class GoEnv:
    def __init__(self, config=None, board_size=None, device=0):
        self.config = config
        self.history_dim = 1                                  # environment.py:35
        self.encoded_dim = getattr(config, "encode_state_channels", 10)   # environment.py:36
        self.max_step = getattr(config, "max_step", 120)      # environment.py:37
        self.komi = getattr(config, "komi", 7.5)              # environment.py:38
        self.board_size = board_size or getattr(config, "board_size", 9)   # environment.py:39
        self.sub_board_size = 7
        cfg = _lib.default_config()
        cfg.board_size, cfg.encode_dim, cfg.max_step, cfg.komi = self.board_size, self.encoded_dim, self.max_step, self.komi
        cfg.n_games, cfg.device = 0, device
        self.ctx = _lib.Context(cfg)
        self.P = self.board_size ** 2
        self.A = self.P + 1
        self.ssz = self.ctx.state_size

    # ---- batched forms ------------------------------------------------------------------------------------------------
    def reset_batch(self, n):
        st = np.zeros((n, self.ssz), np.uint8)
        self.ctx.call("tg_env_reset", _ptr(st), n)
        return st

    def step_batch(self, states, actions):
        states = np.ascontiguousarray(states, np.uint8)
        n = states.shape[0]
        actions = np.ascontiguousarray(actions, np.int32)
        out = np.empty_like(states)
        done = np.zeros(n, np.uint8); ok = np.zeros(n, np.uint8)
        self.ctx.call("tg_env_step", _ptr(states), _ptr(out), _ptr(actions), n, _ptr(done), _ptr(ok))
        return out, done.astype(bool), ok.astype(bool)

    def query_batch(self, states, legal=False, noeye=False, obs=False, score=False, terr=False, meta=False):
        states = np.ascontiguousarray(states, np.uint8)
        n = states.shape[0]
        r = {}
        if legal: r["legal"] = np.zeros((n, self.A), np.uint8)
        if noeye: r["noeye"] = np.zeros((n, self.A), np.uint8)
        if obs: r["obs"] = np.zeros((n, self.encoded_dim, self.board_size, self.board_size), np.float32)
        if score: r["score"] = np.zeros(n, np.float32)
        if terr: r["terr"] = np.zeros((n, self.P), np.float32)
        if meta:
            r["player"] = np.zeros(n, np.int32); r["step"] = np.zeros(n, np.int32); r["terminated"] = np.zeros(n, np.uint8)
        g = lambda k: _ptr(r[k]) if k in r else None
        self.ctx.call("tg_env_query", _ptr(states), n, g("legal"), g("noeye"), g("obs"), g("score"), g("terr"),
                      g("player"), g("step"), g("terminated"))
        return r

    # ---- reference surface (one state per call) -------------------------------------------------------------------------
    @staticmethod
    def _one(state):
        return np.frombuffer(state, np.uint8).reshape(1, -1)

    def reset(self):                                          # environment.py:92-96
        return self.reset_batch(1)[0].tobytes(), False

    def step(self, state, action):                            # environment.py:98-103
        out, done, _ = self.step_batch(self._one(state), [int(action)])
        return out[0].tobytes(), bool(done[0])

    def encode(self, state):                                  # environment.py:105-108
        return self.query_batch(self._one(state), obs=True)["obs"][0]

    def subEncode(self, encode):                              # environment.py:110-113 (only the dead sub_model branch calls it)
        encode = np.ascontiguousarray(encode, np.float32)
        sub_encode = np.zeros([4, self.encoded_dim, self.sub_board_size, self.sub_board_size], dtype="float32")
        self.ctx.lib.tg_host_sub_encode(self.board_size, _ptr(encode), _ptr(sub_encode), self.sub_board_size, self.encoded_dim, 4)
        return sub_encode

    def getScore(self, state):                                # environment.py:115-116
        return float(self.query_batch(self._one(state), score=True)["score"][0])

    def getWinner(self, state):                               # environment.py:118-119
        return BLACK if self.getScore(state) > 0 else WHITE

    def getLegalAction(self, state):                          # environment.py:121-129
        mask = self.query_batch(self._one(state), legal=True)["legal"][0]
        acts = np.flatnonzero(mask).astype(np.int32)
        if len(acts) != 1:
            acts = [a for a in acts if a != self.P]
        return acts

    def getLegalNoEye(self, state):                           # environment.py:163-166
        return np.flatnonzero(self.query_batch(self._one(state), noeye=True)["noeye"][0]).astype(np.int32)

    def getPlayer(self, state):                               # environment.py:132-133
        return int(self.query_batch(self._one(state), meta=True)["player"][0])

    def getStep(self, state):                                 # environment.py:171-172
        return int(self.query_batch(self._one(state), meta=True)["step"][0])

    def justStarted(self, state):                             # environment.py:174-175
        return self.getStep(state) == 1

    def isTerminated(self, state):
        return bool(self.query_batch(self._one(state), meta=True)["terminated"][0])

    def checkAction(self, state, action):                     # environment.py:155-156 -> go_env.cc:84-88 -> TryPlay2 (board.cc:437-464)
        action = int(action)
        if action in (-1, -2):                                # the engine's internal PASS / RESIGN codes are "playable"
            return True
        if action < 0 or action >= self.P:                    # S*S is only translated to PASS by step(); here it is off the board
            return False
        return bool(self.query_batch(self._one(state), legal=True)["legal"][0][action])

    def checkActionAll(self, state):
        return self.query_batch(self._one(state), legal=True)["legal"][0][: self.P].copy()

    def getScoreAndTerritory(self, state):                    # environment.py:158-161
        r = self.query_batch(self._one(state), score=True, terr=True)
        return float(r["score"][0]), r["terr"][0]

    def show(self, state):                                    # environment.py:168-169
        st = np.frombuffer(state, np.uint8)
        self.ctx.call("tg_env_show", _ptr(st))

    def action_to_location(self, action):                     # environment.py:135-138
        return [action // self.board_size, action % self.board_size]

    def location_to_action(self, location):                   # environment.py:140-143
        return self.board_size * location[0] + location[1]
