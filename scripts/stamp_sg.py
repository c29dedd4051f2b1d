"""Diagnostic: phase shares of k_conv3x3_sg (f32 conv) from a -DTG_SD_STAMP build of the library (development aid).
s_memtime ticks are a constant-rate counter (not shader cycles) and every stamp drains the LDS queue, so read the numbers as shares."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd.model import HipNetwork, random_weights
from transgo_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
h = HipNetwork(9, 10, 128, 6, rows_cap=B)
h.set_weights(random_weights(9, 10, 128, 6))
x = (np.random.RandomState(0).rand(B, 10, 9, 9) < 0.2).astype(np.float32)
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
h.main_prediction(x); lib.tg_dbg_read(out)
h.main_prediction(x); lib.tg_dbg_read(out)
n = out[5]; tot = (out[0] + out[3] + out[4]) / n
print("waves", n, "ticks per wave %.0f: prologue %.1f%%  loop %.1f%%  epilogue %.1f%%" % (tot, 100 * out[0] / n / tot, 100 * out[3] / n / tot, 100 * out[4] / n / tot))
print("inside the loop: compute %.1f%%  vmcnt waits %.1f%%  barrier waits %.1f%%  (of wave lifetime)" %
      (100 * out[7] / n / tot, 100 * out[1] / n / tot, 100 * out[2] / n / tot))
