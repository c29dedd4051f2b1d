"""Phase breakdown of k_collect from in-kernel s_memtime stamps (diagnostic build: engine.hip with -DTG_TREE_STAMP, linked into
build/libtransgo_hip_stamp.so; run as `python scripts/stamp_tree.py [games sims filters blocks moves]` after copying that library
over transgo_amd/libtransgo_hip.so).  Prints the share of a game-wave's cycles per phase."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd import _lib, model
from transgo_amd.configure import Config
from transgo_amd.self_play import BatchedSelfPlay

G, sims, F, NB, moves = [int(x) for x in (sys.argv[1:6] + [4096, 400, 32, 2, 12][len(sys.argv) - 1:])][:5]
cfg = Config(num_simulation=sims, num_features=F, num_blocks=NB)
sp = BatchedSelfPlay(cfg, G)
sp.set_weights(model.random_weights(9, 10, F, NB, seed=1234))
sp.start()
lib = _lib.load()
fn = lib.tg_debug_tree_stamps
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(moves):                     # into the middle game first
    sp.advance(num_simulation=16)
fn(None, 1)
sp.advance()
out = (ctypes.c_ulonglong * 24)()
fn(out, 0)
names = ["selection", "parent state + step", "terminal scoring + backup", "child load_colors + analyze", "make_block (legality + child records)",
         "encode_bits (features)", "pending + bookkeeping", "-"]
tot, nw = out[8], out[9]
print(f"game-waves {nw}, mean cycles per game-wave {tot / max(1, nw):.0f}")
for i, n in enumerate(names[:7]):
    print(f"  {n:45s} {100.0 * out[i] / tot:5.1f} %   {out[i] / max(1, nw):8.0f} cycles")
print(f"  {'unaccounted (setup, exit)':45s} {100.0 * (tot - sum(out[:7])) / tot:5.1f} %")
sub = ["encode: suicide mask (shared with legality when make_block ran first)", "encode: liberty classes + eyes", "encode: mark_alive + alive_at",
       "encode: plane masks", "encode: ballot packing + store", "make_block: legal_words", "make_block: header + child records"]
print("sub-phases inside the board code:")
for i, n in enumerate(sub):
    print(f"  {n:75s} {100.0 * out[16 + i] / tot:5.1f} %   {out[16 + i] / max(1, nw):8.0f} cycles")
