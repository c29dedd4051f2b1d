"""19x19 rules fixture.  The reference fixes the board size at compile time (go_comm.h:20), so this script builds a 19x19
variant of the reference engine in a temporary directory OUTSIDE the repository (a scratch copy of GoEnv/cpp_src with the two
constants BOARD_SIZE / MAX_BLOCK edited, as SURVEY.md 8c prescribes; nothing of it is kept), plays seeded random games through
its C ABI and records every observable per ply.  Only the recorded data (tests/golden/rules_s19.npz) is committed."""
import ctypes
import glob
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/GoEnv/cpp_src"
S, P = 19, 361


def build():
    tmp = tempfile.mkdtemp(prefix="transgo_ref19_")
    for f in glob.glob(os.path.join(REF, "*")):
        shutil.copy(f, tmp)
    p = os.path.join(tmp, "go_comm.h")
    src = open(p).read()
    src, n1 = re.subn(r"BOARD_SIZE = 9;", "BOARD_SIZE = 19;", src)
    src, n2 = re.subn(r"MAX_BLOCK = 64;", "MAX_BLOCK = 512;", src)
    assert n1 == 1 and n2 == 1
    open(p, "w").write(src)
    so = os.path.join(tmp, "go_env19.so")
    subprocess.check_call(["g++"] + glob.glob(os.path.join(tmp, "*.cc")) + ["-std=gnu++11", "-O2", "-shared", "-fPIC", "-o", so])
    return tmp, so


def main():
    tmp, so = build()
    try:
        lib = ctypes.CDLL(so)
        St = ctypes.c_char * 32768
        lib.Init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        lib.Step.restype = ctypes.c_bool; lib.checkAction.restype = ctypes.c_bool
        lib.getTerritory.restype = ctypes.c_float
        max_step = 260
        lib.Init(1, 10, max_step, 7.5)
        rng = np.random.RandomState(19)
        rec = {k: [] for k in ("game", "action", "done", "player", "step", "legal", "obs", "score", "terr", "noeye", "check")}
        for g in range(7):
            st = St(); lib.Reset(st)
            pass_p = [0.0, 0.01, 0.05][g % 3]
            done = False
            while True:
                buf = (ctypes.c_int * (P + 1))()
                n = lib.getLegalAction(st, buf)
                la = np.array(buf[:n]); la = la if n == 1 else la[:-1]
                mask = np.zeros(P + 1, np.uint8); mask[la] = 1
                n2 = lib.getLegalNoEye(st, buf); ne = np.zeros(P + 1, np.uint8); ne[np.array(buf[:n2])] = 1
                enc = np.zeros(10 * P, np.float32); lib.Encode(st, enc.ctypes.data_as(ctypes.c_void_p))
                terr = np.zeros(P, np.float32); score = lib.getTerritory(st, terr.ctypes.data_as(ctypes.c_void_p))
                chk = np.array([lib.checkAction(st, ctypes.c_int(a)) for a in range(P)], np.uint8)
                rec["game"].append(g); rec["player"].append(lib.getPlayer(st) & 0xFF); rec["step"].append(lib.getStep(st))
                rec["legal"].append(np.packbits(mask)); rec["noeye"].append(np.packbits(ne))
                rec["obs"].append(np.packbits(enc.astype(np.uint8))); rec["score"].append(score)
                rec["terr"].append(terr.astype(np.int8)); rec["check"].append(np.packbits(chk))
                if done:
                    rec["action"].append(0); rec["done"].append(1)
                    break
                r = rng.rand()
                act = P if r < pass_p else int(rng.randint(P)) if r < pass_p + 0.02 else int(la[rng.randint(len(la))])
                nxt = St(); done = bool(lib.Step(st, nxt, ctypes.c_int(act))); st = nxt
                rec["action"].append(act); rec["done"].append(int(done))
        out = {k: np.asarray(v) for k, v in rec.items()}
        out["score"] = out["score"].astype(np.float32)
        out["size"] = np.int64(S); out["max_step"] = np.int64(max_step)
        np.savez_compressed(os.path.join(HERE, "rules_s19.npz"), **out)
        print("rules_s19 records", len(out["action"]), "bytes", os.path.getsize(os.path.join(HERE, "rules_s19.npz")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
