"""Quick device timing of the network forward at a given batch (development aid)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd.model import HipNetwork, random_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 6
S = int(sys.argv[4]) if len(sys.argv) > 4 else 9
PREC = sys.argv[5] if len(sys.argv) > 5 else "f32"
h = HipNetwork(S, 10, F, NB, rows_cap=B, precision=PREC)
h.set_weights(random_weights(S, 10, F, NB))
x = (np.random.RandomState(0).rand(B, 10, S, S) < 0.2).astype(np.float32)
h.main_prediction(x[:256])
h.ctx.call("tg_prof_enable", 1, 4096)
t = time.time()
for _ in range(3):
    h.main_prediction(x)
dt = (time.time() - t) / 3
ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
h.ctx.call("tg_prof_read", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
print(f"{PREC} S={S} B={B} F={F} N={NB}: wall {dt*1e3:.1f} ms/forward (incl. PCIe); conv3x3 FxF: {n.value} launches, "
      f"{ms.value/n.value:.3f} ms avg, {fl.value/ms.value/1e9:.1f} TFLOP/s")
