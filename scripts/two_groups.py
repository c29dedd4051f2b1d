"""Experiment: G boards as K independent groups, each with its own context / HIP stream, driven by K host threads -- the GPU then
runs the groups' kernels concurrently and one group's partial last conv round, tree stage, stem and heads fill with the other
group's work.  usage: two_groups.py [boards] [groups] [moves] [filters] [blocks] [dtype]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd import model
from transgo_amd.configure import Config
from transgo_amd.self_play import BatchedSelfPlay

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 4
F = int(sys.argv[4]) if len(sys.argv) > 4 else 128
NB = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dtype = sys.argv[6] if len(sys.argv) > 6 else "f32"
cfg = Config(num_simulation=400, num_features=F, num_blocks=NB, inference_dtype=dtype)
w = model.random_weights(9, 10, F, NB, seed=1234)
sps = []
for k in range(K):
    sp = BatchedSelfPlay(cfg, G // K, rank=k, world=K)
    sp.set_weights(w)
    sp.start()
    sps.append(sp)
# mid-game positions: a few cheap moves first
for sp in sps:
    for _ in range(12):
        sp.advance(num_simulation=16)

def run(sp, n):
    for _ in range(n):
        sp.advance()

def timed(n):
    s0 = [sp.engine.stats()["sims"] for sp in sps]
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(sp, n)) for sp in sps]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    s1 = [sp.engine.stats()["sims"] for sp in sps]
    return sum(b - a for a, b in zip(s0, s1)) / dt, dt / n

timed(1)
v, step = timed(moves)
print(f"{G} boards as {K} group(s) of {G // K}, {dtype} {NB}x{F}: {v:,.0f} sims/s, {step * 1e3:.1f} ms per move of all groups")
