"""Diagnostic: phase shares of k_conv3x3_h (fp16 conv) from a -DTG_SD_STAMP build of the library (development aid)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd.model import HipNetwork, random_weights
from transgo_amd import _lib
B, F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 256
h = HipNetwork(9, 10, F, 6, rows_cap=B, precision="f16")
h.set_weights(random_weights(9, 10, F, 6))
x = (np.random.RandomState(0).rand(B, 10, 9, 9) < 0.2).astype(np.float32)
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
h.main_prediction(x); lib.tg_dbg_read(out)
h.main_prediction(x); lib.tg_dbg_read(out)
n = out[5]; nst = 9 * F // 64          # barrier intervals: 64 channels x 1 tap (shape 1) = 2 x (32 channels x 1 tap) (shape 2)
print("waves", n, "per-wave: prologue %.0f  loop %.0f  epilogue %.0f" % (out[0] / n, out[3] / n, out[4] / n))
print("per stage (%d): dma-issue %.0f  compute %.0f  dma-wait %.0f  barrier %.0f ; MFMA per stage per wave = 64 x 16 = 1024 cycles" %
      (nst, out[6] / n / nst, out[7] / n / nst, out[1] / n / nst, out[2] / n / nst))
