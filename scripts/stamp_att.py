"""Phase breakdown of k_attention_x3 from in-kernel s_memtime stamps (diagnostic build: net.hip with -DTG_ATT_STAMP linked into
build/libtransgo_hip_astamp.so and copied over transgo_amd/libtransgo_hip.so on the GPU box).  One MainNetwork forward of `rows`
boards under net_precision 3; the stamps are those of the third board of wave 0 of workgroup 0 in the last trunk attention launch."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd import _lib, model
from transgo_amd.model import HipNetwork, transgo_arch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
h = HipNetwork(9, 10, 128, rows_cap=rows, arch=transgo_arch(), precision="f32x3")
h.set_weights(model.random_transgo_weights(9, 10, 128, seed=1234))
x = (np.random.RandomState(0).rand(rows, 10, 9, 9) < 0.2).astype(np.float32)
for _ in range(3):
    t0 = time.time(); h.main_prediction(x); dt = time.time() - t0
print(f"forward of {rows} boards incl. H2D/D2H: {dt * 1e3:.1f} ms")
lib = _lib.load()
fn = lib.tg_debug_att_stamps
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 32)()
assert fn(out, 32) == 0
s = [int(v) for v in out]
tick = 10.0                                                   # ns per s_memtime tick (100 MHz)
print(f"board total {(s[26] - s[0]) * tick / 1e3:.2f} us; phase A (q, k projection) {(s[1] - s[0]) * tick / 1e3:.2f} us (its first two groups {(s[29] - s[0]) * tick / 1e3:.2f})")
print(f"  row loads + accumulator init + touches issued: {(s[27] - s[1]) * tick / 1e3:.2f} us; epilogue first half {(s[28] - s[25]) * tick / 1e3:.2f}, second half {(s[26] - s[28]) * tick / 1e3:.2f} us")
prev = s[1]
tot = dict(v=0, g1=0, sm=0, g2=0)
for tm in range(6):
    a, b, c, d = s[2 + 4 * tm], s[3 + 4 * tm], s[4 + 4 * tm], s[5 + 4 * tm]
    print(f"  block {tm}: v projection {(a - prev) * tick / 1e3:.2f}  energy GEMM {(b - a) * tick / 1e3:.2f}  softmax {(c - b) * tick / 1e3:.2f}  "
          f"output GEMM {(d - c) * tick / 1e3:.2f} us")
    tot["v"] += a - prev; tot["g1"] += b - a; tot["sm"] += c - b; tot["g2"] += d - c
    prev = d
print(f"sums: v projection {tot['v'] * tick / 1e3:.2f}  energy {tot['g1'] * tick / 1e3:.2f}  softmax {tot['sm'] * tick / 1e3:.2f}  "
      f"output {tot['g2'] * tick / 1e3:.2f}  epilogue {(s[26] - prev) * tick / 1e3:.2f} us")
