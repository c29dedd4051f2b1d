"""Soak of the finished-game path (development aid): several game generations of 4096 staggered boards, every finished game
harvested in device memory, appended device -> device to the replay store AND checked on a host copy (lengths, winners, z = +-1
consistent with the winner, territory sign, visit totals, pi a distribution), batches sampled from the store now and then."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd import model
from transgo_amd.configure import Config
from transgo_amd.replay_buffer import DeviceReplayMemory
from transgo_amd.self_play import BatchedSelfPlay

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
SIMS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
MAXSTEP = int(sys.argv[4]) if len(sys.argv) > 4 else 120
cfg = Config(num_simulation=SIMS, num_features=32, num_blocks=2, max_step=MAXSTEP)
sp = BatchedSelfPlay(cfg, G)
sp.set_weights(model.random_weights(9, 10, 32, 2, seed=1))
mem = DeviceReplayMemory(cfg, capacity_positions=1 << 20)
sp.start()
offs = np.arange(G) % MAXSTEP
for s in range(MAXSTEP - 1):                                   # stagger as bench.py does
    m = offs == (MAXSTEP - 1 - s)
    if s > 0 and m.any():
        sp._reset(m)
    sp.advance(num_simulation=16)
t0 = time.time(); games = pos = 0; lens = []
for i in range(STEPS):
    h = sp.advance(device=True)
    if h is not None:
        mem.append_harvest(h)
        hh = h.to_host()
        nm, win, terr = hh.view("n_moves"), hh.view("winner"), hh.view("terr")
        z, own, pl, cnt = hh.view("z"), hh.view("own"), hh.view("player"), hh.view("counts")
        assert (nm >= 1).all() and (nm <= MAXSTEP).all() and np.isin(win, (1, 2)).all() and int(nm.sum()) == hh.n_positions
        wpos, tpos = np.repeat(win, nm), np.repeat(terr, nm, axis=0)
        assert np.array_equal(z, np.where(pl == wpos, 1.0, -1.0).astype(np.float32))
        assert np.array_equal(own, np.where((pl == 1)[:, None], tpos, -tpos))
        assert (cnt.sum(1) >= 15).all() and (cnt >= 0).all()        # the staggering moves searched 16 simulations
        assert np.allclose(hh.pis().sum(1), 1.0)
        games += hh.n_games; pos += hh.n_positions; lens += list(nm)
    if (i + 1) % 100 == 0:
        st = sp.engine.stats()
        s_, p_, z_, o_ = mem.sample(256)
        assert s_.shape == (256, 10, 9, 9) and np.allclose(p_.sum(1), 1.0, atol=1e-5) and set(np.unique(z_)) <= {-1.0, 1.0}
        print(f"step {i+1}: {games} games / {pos} positions harvested, mean length {np.mean(lens):.1f}, replay entries {mem.info()['entries']}, "
              f"errors {st['errors']}, dropped {sp.games_dropped}, truncated blocks {st['truncated_blocks']}, {st['sims'] / (time.time() - t0):.0f} sims/s", flush=True)
st = sp.engine.stats()
assert st["errors"] == 0 and sp.games_dropped == 0 and games == sp.games_finished
assert mem.info()["entries"] == min(8 * pos, 8 * (1 << 20))
assert games >= G * (STEPS // MAXSTEP) * 0.95
print("soak_records ok:", games, "games,", pos, "positions; mean length", round(float(np.mean(lens)), 2))
