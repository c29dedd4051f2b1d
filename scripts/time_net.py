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
DATA = sys.argv[6] if len(sys.argv) > 6 else "random"       # "zeros": all-zero weights and planes -- same instruction stream, the
h = HipNetwork(S, 10, F, NB, rows_cap=B, precision=PREC)    # MFMAs toggle no bits: shows what the chip's clock management costs
sd = random_weights(S, 10, F, NB)
if DATA == "zeros":
    sd = {k: (np.zeros_like(v) if "running_var" not in k and np.ndim(v) else v) for k, v in sd.items()}
h.set_weights(sd)
x = (np.random.RandomState(0).rand(B, 10, S, S) < (0.0 if DATA == "zeros" else 0.2)).astype(np.float32)
h.main_prediction(x[:256])
h.ctx.call("tg_prof_enable", 1, 4096)
t = time.time()
for _ in range(3):
    h.main_prediction(x)
dt = (time.time() - t) / 3
ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
h.ctx.call("tg_prof_read", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
print(f"{DATA} data, {PREC} S={S} B={B} F={F} N={NB}: wall {dt*1e3:.1f} ms/forward (incl. PCIe); conv3x3 FxF: {n.value} launches, "
      f"{ms.value/n.value:.3f} ms avg, {fl.value/ms.value/1e9:.1f} TFLOP/s")
