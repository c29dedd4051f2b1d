"""Diagnostic: phase shares of k_conv3x3_sd from the -DTG_SD_STAMP build (development aid)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd.model import HipNetwork, random_weights
from transgo_amd import _lib
B = 16384
h = HipNetwork(9, 10, 128, 6, rows_cap=B)
h.set_weights(random_weights(9, 10, 128, 6))
x = (np.random.RandomState(0).rand(B, 10, 9, 9) < 0.2).astype(np.float32)
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
h.main_prediction(x); lib.tg_dbg_read(out)
h.main_prediction(x); lib.tg_dbg_read(out)
n = out[5]
print("waves", n, "per-wave cycles: prologue %.0f  dma-wait %.0f  barrier-wait %.0f  loop %.0f  epilogue %.0f" %
      tuple(out[i] / n for i in range(5)))
print("per stage (72): dma-wait %.0f  barrier %.0f  stage %.0f cycles; MFMA per stage per wave = 96 x 32 = 3072 cycles (x waves per SIMD)" %
      (out[1] / n / 72, out[2] / n / 72, out[3] / n / 72))
