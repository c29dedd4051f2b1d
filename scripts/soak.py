"""Soak run of the batched self-play loop (development aid): several game generations per slot, checks that no tree error is
raised, that finished games carry consistent records, and reports throughput and the arena high-water mark."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd import model
from transgo_amd.configure import Config
from transgo_amd.self_play import BatchedSelfPlay

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
SIMS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
DTYPE = sys.argv[4] if len(sys.argv) > 4 else "f32"
S = int(sys.argv[5]) if len(sys.argv) > 5 else 9
MAXSTEP = int(sys.argv[6]) if len(sys.argv) > 6 else 120
F, NB = (128, 2) if (DTYPE in ("f16", "f16r", "f32x3") or S == 19) else (32, 2)      # 19x19 is built for 128 and 256 filters
cfg = Config(num_simulation=SIMS, num_features=F, num_blocks=NB, inference_dtype=DTYPE, board_size=S, max_step=MAXSTEP)
sp = BatchedSelfPlay(cfg, G)
sp.set_weights(model.random_weights(S, 10, F, NB, seed=1))
sp.start()
t0 = time.time(); fin = 0; lens = []; winners = np.zeros(3, np.int64)
for i in range(STEPS):
    done = sp.step()
    for r in done:
        assert len(r.pis) == len(r.players) == len(r.visits) and r.winner in (1, 2)
        assert all(abs(p.sum() - 1.0) < 1e-9 for p in r.pis[:3])
        lens.append(len(r.pis)); winners[r.winner] += 1
    fin += len(done)
    if (i + 1) % 50 == 0:
        st = sp.engine.stats()
        print(f"step {i+1}: finished {fin} games, mean length {np.mean(lens) if lens else 0:.1f}, errors {st['errors']}, "
              f"arena high-water {st['max_slots']}, {st['sims'] / (time.time() - t0):.0f} sims/s", flush=True)
st = sp.engine.stats()
assert st["errors"] == 0 and sp.games_dropped == 0, (st, sp.games_dropped)
print("largest tree", st["max_slots"], "slots; pool", st["pool_slots"], "slots, high-water", st["pool_high_water"], f"({st['pool_high_water'] / max(1, st['pool_slots']):.1%}), ran empty",
      st["pool_exhausted"], "; truncated tree blocks", st["truncated_blocks"], "; games dropped", sp.games_dropped, "; fp16 overflows", st["fp16_overflows"])
assert fin >= G * (STEPS // (MAXSTEP + 5)), (fin, "fewer finished games than the ply limit guarantees")
print("soak ok:", fin, "games; black/white wins", winners[1], winners[2], "; mean length", round(float(np.mean(lens)), 1) if lens else 0)
