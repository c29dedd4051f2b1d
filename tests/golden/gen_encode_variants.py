"""Fixture for the other two feature encodings the reference engine offers (Init(..., encode_dim = 9 | 13 ...), go_env.cc:96-115,
board_feature.cc:213-253); the reference's Python only ever uses 10, so these are recorded straight from the compiled
reference engine (oracle/_ref/GoEnv/go_env.so, `make -C oracle ref`) through its own C ABI: seeded random games, and at every
4th position the 9- and 13-plane encodings bit-packed.  Output: tests/golden/rules_enc_variants_s9.npz."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "GoEnv", "go_env.so")
S, P = 9, 81


def main():
    lib = ctypes.CDLL(SO)
    St = ctypes.c_char * 4096
    lib.Init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    lib.Step.restype = ctypes.c_bool
    rng = np.random.RandomState(913)
    rec = {k: [] for k in ("game", "ply", "action", "obs9", "obs13")}
    for g in range(40):
        lib.Init(1, 10, 120, 7.5)
        st = St(); lib.Reset(st)
        done, ply, actions = False, 0, []
        while not done:
            buf = (ctypes.c_int * (P + 1))()
            n = lib.getLegalAction(st, buf)
            la = np.array(buf[:n]); la = la if n == 1 else la[:-1]
            act = P if rng.rand() < 0.03 else int(la[rng.randint(len(la))])
            if ply % 4 == 0:
                for dim, key in ((9, "obs9"), (13, "obs13")):
                    lib.Init(1, dim, 120, 7.5)                      # the encoding is a file-static of the engine (go_env.cc:9-12)
                    enc = np.zeros(dim * P, np.float32); lib.Encode(st, enc.ctypes.data_as(ctypes.c_void_p))
                    rec[key].append(np.packbits(enc.astype(np.uint8)))
                lib.Init(1, 10, 120, 7.5)
                rec["game"].append(g); rec["ply"].append(ply)
            nxt = St(); done = bool(lib.Step(st, nxt, ctypes.c_int(act))); st = nxt
            actions.append(act); ply += 1
        rec["action"].append(np.array(actions + [-1] * (130 - len(actions)), np.int16))
    out = {k: np.asarray(v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, "rules_enc_variants_s9.npz"), **out)
    print("positions", len(out["game"]), "bytes", os.path.getsize(os.path.join(HERE, "rules_enc_variants_s9.npz")))


if __name__ == "__main__":
    main()
