"""ctypes binding of oracle/libgo_oracle.so with the reference GoEnv method surface.  TEST INFRASTRUCTURE ONLY.

Mirrors /root/reference/GoEnv/environment.py:32-198 (same method names, argument meaning and return types) on top of
this repo's own CPU restatement (oracle/go_oracle.c).  Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this module; the product (transgo_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgo_oracle.so")
BLACK, WHITE = 1, 2


class OgCfg(ctypes.Structure):
    _fields_ = [("size", ctypes.c_int32), ("max_step", ctypes.c_int32), ("komi", ctypes.c_float),
                ("encode_dim", ctypes.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "go_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _LIB


def _load():
    lib = ctypes.CDLL(build())
    vp, ip, fp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)
    cp = ctypes.POINTER(OgCfg)
    lib.og_reset.argtypes = [cp, vp]
    lib.og_step.argtypes = [cp, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.og_step_inplace.argtypes = [cp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.og_check_action.argtypes = [cp, vp, ctypes.c_int]
    lib.og_legal_actions.argtypes = [cp, vp, ip]
    lib.og_legal_no_eye.argtypes = [cp, vp, ip]
    lib.og_encode.argtypes = [cp, vp, fp]
    lib.og_score.argtypes = [cp, vp]
    lib.og_score.restype = ctypes.c_float
    lib.og_territory.argtypes = [cp, vp, fp]
    lib.og_territory.restype = ctypes.c_float
    lib.og_check_all.argtypes = [cp, vp, ctypes.c_void_p]
    for n in ("og_player", "og_step_count", "og_terminated"):
        getattr(lib, n).argtypes = [vp]
    return lib


class OracleGoEnv:
    """Same surface as reference GoEnv (environment.py:32-198); states are opaque ctypes buffers."""

    def __init__(self, config=None, board_size=9, max_step=120, komi=7.5, encode_dim=10):
        self.config = config
        if config is not None:
            board_size = getattr(config, "board_size", board_size)
            max_step = getattr(config, "max_step", max_step)
            komi = getattr(config, "komi", komi)
            encode_dim = getattr(config, "encode_state_channels", encode_dim)
        self.board_size, self.max_step, self.komi, self.encoded_dim = board_size, max_step, komi, encode_dim
        self.lib = _load()
        self.cfg = OgCfg(board_size, max_step, komi, encode_dim)
        self._ssize = self.lib.og_state_size()
        self.P = board_size * board_size

    def _new(self):
        return ctypes.create_string_buffer(self._ssize)

    def reset(self):                                   # environment.py:92-96
        s = self._new()
        self.lib.og_reset(ctypes.byref(self.cfg), s)
        return s, False

    def step(self, state, action):                     # environment.py:98-103
        s = self._new()
        ok = ctypes.c_int(1)
        done = self.lib.og_step(ctypes.byref(self.cfg), state, s, int(action), ctypes.byref(ok))
        return s, bool(done)

    def encode(self, state):                           # environment.py:105-108
        out = np.zeros([self.encoded_dim, self.board_size, self.board_size], dtype=np.float32)
        self.lib.og_encode(ctypes.byref(self.cfg), state, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return out

    def getScore(self, state):                         # environment.py:115-116
        return self.lib.og_score(ctypes.byref(self.cfg), state)

    def getWinner(self, state):                        # environment.py:118-119
        return BLACK if self.getScore(state) > 0 else WHITE

    def getLegalAction(self, state):                   # environment.py:121-129 (pass filtered unless only move)
        buf = np.zeros([self.P + 1], dtype=np.int32)
        n = self.lib.og_legal_actions(ctypes.byref(self.cfg), state, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        acts = buf[:n]
        if n != 1:
            acts = [a for a in acts if a != self.P]
        return acts

    def getLegalNoEye(self, state):                    # environment.py:163-166
        buf = np.zeros([self.P + 1], dtype=np.int32)
        n = self.lib.og_legal_no_eye(ctypes.byref(self.cfg), state, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return buf[:n]

    def getPlayer(self, state):                        # environment.py:132-133
        return self.lib.og_player(state)

    def getStep(self, state):                          # environment.py:171-172
        return self.lib.og_step_count(state)

    def checkAction(self, state, action):              # environment.py:155-156
        return bool(self.lib.og_check_action(ctypes.byref(self.cfg), state, int(action)))

    def checkActionAll(self, state):                   # bulk checkAction over the board points
        out = np.zeros([self.P], dtype=np.uint8)
        self.lib.og_check_all(ctypes.byref(self.cfg), state, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def getScoreAndTerritory(self, state):             # environment.py:158-161
        terr = np.zeros([self.P], dtype=np.float32)
        score = self.lib.og_territory(ctypes.byref(self.cfg), state, terr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return score, terr

    def isTerminated(self, state):
        return bool(self.lib.og_terminated(state))

    def action_to_location(self, action):              # environment.py:135-138
        return [action // self.board_size, action % self.board_size]

    def location_to_action(self, location):            # environment.py:140-143
        return self.board_size * location[0] + location[1]
