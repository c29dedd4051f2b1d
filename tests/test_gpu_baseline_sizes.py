"""BASELINE.json's configurations at their own sizes (VERDICT r1 "configs not exercised at their size"):
C2 = 4096 concurrent boards x 400 simulations (tree kernels, atomic row allocator, 16 384-row batches) against the oracle;
C4 = 20-block x 256-filter tower at 19x19 against the fp32 torch tower (the 800-simulation 19x19 search itself is pinned by
tests/test_gpu_search.py::test_reference_19x19_search[sharp_s24_n800]); C5 = 40-block x 256-filter fp16 chain against the
half-storage oracle and the f32 tower, with the error growth over depth printed."""
import numpy as np
import pytest

from oracle import evaluators

pytestmark = pytest.mark.gpu


def test_c2_4096_boards_400_sims_vs_oracle():
    from oracle.go_oracle import OracleGoEnv
    from oracle.wp_mcts import OracleSearch
    from transgo_amd.engine import SelfPlayEngine
    G, sims, moves = 4096, 400, 2
    fn = evaluators.sharp
    eng = SelfPlayEngine(G, num_simulation=sims, evaluator=fn)
    seeds = np.arange(G).astype(np.uint32)
    eng.reset(seeds)
    sample = np.random.RandomState(0).choice(G, 64, replace=False)
    sample[:4] = [0, 1, G - 2, G - 1]
    orcs = {int(g): OracleSearch(OracleGoEnv(), fn, np.random.RandomState(int(seeds[g])), num_simulation=sims) for g in sample}
    n0 = np.zeros(G, np.int64)
    for m in range(moves):
        waves = eng.search()
        vis, rn, pl, st, _ = eng.root_info(obs=False)
        assert (rn >= n0 + sims).all() and (rn < n0 + sims + 4).all(), m          # self_play.py:662-664, R = 4 read-outs per wave
        assert ((vis.sum(1) == rn) | (vis.sum(1) == rn - 1)).all()
        acts, pis = eng.choose_moves(vis, st)
        for g, o in orcs.items():
            a, pi, _, _ = o.search_move()
            raw = np.array([o.root.kids[i].n if i in o.root.kids else 0 for i in range(82)])
            assert (raw == vis[g]).all() and a == acts[g] and (pi == pis[g]).all(), (g, m)
            o.advance(a)
        eng.play(acts)
        n0 = np.array([vis[g, acts[g]] for g in range(G)], np.int64)
        print(f"move {m}: {waves} waves, mean inherited visits {n0.mean():.1f}")
    s = eng.stats()
    assert s["errors"] == 0 and s["truncated_blocks"] == 0 and s["sims"] >= G * sims * moves
    eng.close()


def _positions(S, n, seed):
    from oracle.go_oracle import OracleGoEnv
    env = OracleGoEnv(board_size=S, max_step=S * S)
    rng = np.random.RandomState(seed)
    obs = []
    while len(obs) < n:
        s, done = env.reset()
        while not done and len(obs) < n:
            la = env.getLegalAction(s)
            s, done = env.step(s, int(la[rng.randint(len(la))]))
            if rng.rand() < 0.2:
                obs.append(env.encode(s))
    return np.stack(obs)


def test_c4_tower_20x256_at_19x19_vs_torch():
    import torch
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(8)
    net = seeded_tower(19, 10, 256, 20, seed=2024)
    x = _positions(19, 4, 21)
    with torch.no_grad():
        p, v, o = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
    h = HipNetwork(19, 10, 256, 20, rows_cap=8)
    h.set_weights(net.get_weights())
    hp, hv, ho = h.main_prediction(x)
    e = [np.abs(a - b).max() for a, b in ((hp, p), (hv, v), (ho, o))]
    print(f"C4 20x256 @19x19 f32: max abs err policy {e[0]:.2e} value {e[1]:.2e} own {e[2]:.2e}")
    assert max(e) < 1e-3                                               # north_star: policy/value within 1e-3 fp32


def test_c5_fp16_40x256_at_19x19_error_growth():
    import torch
    from oracle.net import half_storage_forward, seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(8)
    x = _positions(19, 2, 22)
    rows = []
    for NB in (4, 10, 20, 40):
        net = seeded_tower(19, 10, 256, NB, seed=500 + NB)
        p16, v16, o16 = [t.numpy() for t in half_storage_forward(net, torch.from_numpy(x))]
        with torch.no_grad():
            p32, v32, o32 = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
        h = HipNetwork(19, 10, 256, NB, rows_cap=4, precision="f16")
        h.set_weights(net.get_weights())
        hp, hv, ho = h.main_prediction(x)
        e16 = [float(np.abs(a - b).max()) for a, b in ((hp, p16), (hv, v16), (ho, o16))]
        e32 = [float(np.abs(a - b).max()) for a, b in ((hp, p32), (hv, v32), (ho, o32))]
        rows.append((NB, e16, e32))
        print(f"fp16 {NB:2d}x256 @19x19: vs half-storage oracle policy {e16[0]:.2e} value {e16[1]:.2e} own {e16[2]:.2e} | "
              f"vs f32 tower policy {e32[0]:.2e} value {e32[1]:.2e} own {e32[2]:.2e}")
        h.ctx.close()
    for NB, e16, e32 in rows:
        assert max(e16) < 1e-3, NB                                     # same rounding points, only the summation order differs
        assert e32[0] < 5e-3 and e32[1] < 2e-2 and e32[2] < 2e-2, NB   # fp16 path within ~5e-3 on probabilities
