"""Opt-in split-precision network ("f32x3", net_precision 3): every conv operand of the tower is carried as fp16
hi + lo, three of the four partial products of (w_hi + w_lo)(a_hi + a_lo) (lo*lo is dropped) run on the fp16 matrix cores with f32 accumulation, the residual
stream, the narrow head conv and the dense heads stay f32.  ~22 significand bits per operand: the outputs must sit within 1e-5 of
the fp32 torch tower (north_star allows 1e-3), and searches driven by it must agree with the oracle driving the torch network about
as well as the exact-f32 path does.  The default path is untouched (tests/test_gpu_net.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _positions(S, n, seed):
    rng = np.random.RandomState(seed)
    x = np.zeros((n, 10, S, S), np.float32)
    occ = rng.rand(n, S, S)
    for c in range(3):
        x[:, c] = (occ < 0.1 * (c + 1)) & (occ >= 0.1 * c)
        x[:, 3 + c] = (occ > 1 - 0.1 * (c + 1)) & (occ <= 1 - 0.1 * c)
    x[:, 6:] = rng.rand(n, 4, S, S) < 0.05
    return x


@pytest.mark.parametrize("S,F,NB,n", [(9, 128, 6, 600), (19, 256, 20, 4), (9, 256, 3, 40), (19, 128, 2, 3)])
def test_split_precision_tower_within_1e5_of_torch_f32(S, F, NB, n):
    """BASELINE configs[1]'s net (6 x 128 @ 9x9, several row tiles incl. a partial one) and configs[3]'s (20 x 256 @ 19x19)."""
    import torch
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(8)
    net = seeded_tower(S, 10, F, NB, seed=4000 + NB)
    x = _positions(S, n, 8)
    with torch.no_grad():
        p, v, o = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
    h = HipNetwork(S, 10, F, NB, rows_cap=max(8, n), precision="f32x3")
    h.set_weights(net.get_weights())
    hp, hv, ho = h.main_prediction(x)
    e = [float(np.abs(a - b).max()) for a, b in ((hp, p), (hv, v), (ho, o))]
    # the exact-f32 path on the same input, for scale
    h32 = HipNetwork(S, 10, F, NB, rows_cap=max(8, n))
    h32.set_weights(net.get_weights())
    e32 = [float(np.abs(a - b).max()) for a, b in zip(h32.main_prediction(x), (p, v, o))]
    print(f"f32x3 {NB}x{F}@{S}x{S}: max abs err vs torch f32 policy {e[0]:.2e} value {e[1]:.2e} own {e[2]:.2e} "
          f"(exact-f32 MFMA path: {e32[0]:.2e} {e32[1]:.2e} {e32[2]:.2e})")
    assert max(e) < TOL
    assert np.allclose(hp.sum(1), 1.0, atol=1e-5)


@pytest.mark.parametrize("F,n", [(128, 300), (256, 40)])
def test_split_precision_mainnetwork_with_attention(F, n):
    """The reference's shipped MainNetwork ("RARRRARRRRAR+P", model.py:49-76) under net_precision 3: its nine residual blocks on
    the split-precision convs, the Self_Attention layers (trunk and policy head) on the f32 kernels, which hand a residual
    block that follows its input already split (k_attention_mfma<..., X2O>).  Against the torch restatement that
    tests/golden/net_transgo_f32.npz pins, and against the exact-f32 HIP path on the same input."""
    import torch
    from oracle.net import TransGoMain
    from transgo_amd.model import HipNetwork, transgo_arch
    torch.manual_seed(5); torch.set_num_threads(8)
    net = TransGoMain(9, 10, F).eval()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            if hasattr(m, "gamma"):
                m.gamma.copy_(0.5 + torch.rand(1, generator=g))
    x = _positions(9, n, 12)
    with torch.no_grad():
        p, v, o = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
    sd = {k: t.numpy() for k, t in net.state_dict().items()}
    h = HipNetwork(9, 10, F, rows_cap=max(8, n), arch=transgo_arch(), precision="f32x3")
    h.set_weights(sd)
    hp, hv, ho = h.main_prediction(x)
    e = [float(np.abs(a - b).max()) for a, b in ((hp, p), (hv, v), (ho, o))]
    h32 = HipNetwork(9, 10, F, rows_cap=max(8, n), arch=transgo_arch())
    h32.set_weights(sd)
    e32 = [float(np.abs(a - b).max()) for a, b in zip(h32.main_prediction(x), (p, v, o))]
    print(f"f32x3 MainNetwork F={F}: max abs err vs torch f32 policy {e[0]:.2e} value {e[1]:.2e} own {e[2]:.2e} "
          f"(exact-f32 MFMA path: {e32[0]:.2e} {e32[1]:.2e} {e32[2]:.2e})")
    assert max(e) < TOL
    assert np.allclose(hp.sum(1), 1.0, atol=1e-5)
    # a batch that is not a multiple of the attention kernel's four boards per workgroup, and a single row
    for k in (1, 7):
        qp, qv, qo = h.main_prediction(x[:k])
        assert np.array_equal(qp, hp[:k]) and np.array_equal(qv, hv[:k]) and np.array_equal(qo, ho[:k])
    if F == 128:
        # more boards than the fused attention kernel has waves (256 workgroups x 4): every wave walks several boards, with the next
        # board's rows requested while the current one is stored -- against the exact-f32 HIP path on the same batch (both sit
        # ~1e-7 from torch), and row for row against the small batch above (a board's result does not depend on its neighbours)
        nb = 2 * 1024 + 131
        xb = np.concatenate([x, _positions(9, nb - n, 13)])
        hb = HipNetwork(9, 10, F, rows_cap=nb, arch=transgo_arch(), precision="f32x3"); hb.set_weights(sd)
        hb32 = HipNetwork(9, 10, F, rows_cap=nb, arch=transgo_arch()); hb32.set_weights(sd)
        bp, bv, bo = hb.main_prediction(xb)
        eb = [float(np.abs(a - b).max()) for a, b in zip((bp, bv, bo), hb32.main_prediction(xb))]
        print(f"f32x3 MainNetwork F={F}, {nb} boards: max abs difference to the exact-f32 HIP path {eb[0]:.2e} {eb[1]:.2e} {eb[2]:.2e}")
        assert max(eb) < TOL
        assert np.array_equal(bp[:n], hp) and np.array_equal(bv[:n], hv) and np.array_equal(bo[:n], ho)


def test_split_precision_background_refresh_and_refusals():
    """The weight hand-off (tg_net_load_async -> switch at a boundary) restages the split copies too; widths the fp16 kernels are not
    built for are refused loudly."""
    import ctypes
    from transgo_amd import model
    from transgo_amd._lib import TransgoError
    from transgo_amd.model import HipNetwork
    x = _positions(9, 20, 3)
    sds = [model.random_weights(9, 10, 128, 2, seed=s) for s in (11, 12)]
    want = []
    for sd in sds:
        h = HipNetwork(9, 10, 128, 2, rows_cap=32, precision="f32x3")
        h.set_weights(sd)
        want.append(h.main_prediction(x))
        h.ctx.close()
    h = HipNetwork(9, 10, 128, 2, rows_cap=32, precision="f32x3")
    h.set_weights(sds[0])
    blob = model.pack_weights(sds[1], 9, 10, 128, arch=h.arch)
    h.ctx.call("tg_net_load_async", h.arch.code.encode(), blob.ctypes.data_as(ctypes.c_void_p), blob.size)
    pend = ctypes.c_int(-1)
    h.ctx.call("tg_net_load_poll", 1, ctypes.byref(pend))
    assert pend.value == 0
    assert all(np.array_equal(a, b) for a, b in zip(h.main_prediction(x), want[1]))
    with pytest.raises(TransgoError):
        HipNetwork(9, 10, 64, 2, rows_cap=8, precision="f32x3").set_weights(model.random_weights(9, 10, 64, 2))


def test_split_precision_search_agrees_with_the_oracle():
    """Whole engine with the split-precision tower against the CPU oracle driving the fp32 torch tower (same weights, same seeds),
    as tests/test_gpu_selfplay.py::test_end_to_end_visit_counts_with_the_real_network does for the exact-f32 path: every game's
    first move identical, and the great majority of all (game, move) visit-count vectors."""
    import torch
    from oracle.go_oracle import OracleGoEnv
    from oracle.net import TowerNetwork
    from oracle.wp_mcts import OracleSearch
    from transgo_amd import model
    from transgo_amd.engine import SelfPlayEngine
    torch.set_num_threads(4)
    G, sims, moves, F, NB = 12, 48, 4, 128, 2
    sd = model.random_weights(9, 10, F, NB, seed=78)
    net = TowerNetwork(9, 10, F, NB).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})

    def ev(obs):
        with torch.no_grad():
            p, v, _ = net.main_prediction(torch.from_numpy(obs))
        return p.numpy(), v.numpy()
    eng = SelfPlayEngine(G, num_simulation=sims, net_blocks=NB, net_filters=F, net_precision="f32x3")
    model.load_into(eng.ctx, sd, 9, 10, F, NB)
    seeds = np.arange(600, 600 + G)
    eng.reset(seeds)
    orcs = [OracleSearch(OracleGoEnv(), ev, np.random.RandomState(int(s)), num_simulation=sims) for s in seeds]
    same, total, alive = 0, 0, np.ones(G, bool)
    for m in range(moves):
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        acts, pis = eng.choose_moves(vis, st)
        for g, o in enumerate(orcs):
            if not alive[g]:
                continue
            a, pi, obs, info = o.search_move()
            raw = np.array([o.root.kids[i].n if i in o.root.kids else 0 for i in range(82)])
            ok = (raw == vis[g]).all() and a == acts[g]
            total += 1; same += int(ok)
            if m == 0:
                assert ok, (g, "first move must match exactly")
            if not ok:
                alive[g] = False
            else:
                o.advance(a)
        eng.play(acts)
    print(f"f32x3: identical visit-count vectors {same}/{total}")
    assert same >= 0.85 * total


def test_fp16_range_guard_counts_overflows_in_every_fp16_carrying_mode():
    """VERDICT r3 item 5.  The reference runs f32 end to end (model.py:79-114) and takes any trained checkpoint (model.py:23-27); the
    modes that carry fp16 lose f32's range.  A tower whose second block's pre-activation BatchNorm scale is multiplied by 1e5 pushes
    relu(bn1(x)) far beyond 65504: f16 / f16r / f32x3 must raise the sticky counter (tg_net_range), exact f32 must not (and still
    stays finite); the same weights with the scale left alone leave the counter at 0 in every mode."""
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    S, F, NB, n = 9, 128, 2, 40
    net = seeded_tower(S, 10, F, NB, seed=77)
    good = {k: np.array(v) for k, v in net.get_weights().items()}
    bad = dict(good)
    bad["main_network.res_blocks.1.batchnormlize_1.weight"] = good["main_network.res_blocks.1.batchnormlize_1.weight"] * np.float32(1e5)
    x = _positions(S, n, 3)
    for prec in ("f32", "f16", "f16r", "f32x3"):
        h = HipNetwork(S, 10, F, NB, rows_cap=n, precision=prec)
        rng = h.set_weights(good)
        assert rng["finite"] and 0 < rng["weight_absmax"] < 100
        h.main_prediction(x)
        r0 = h.net_range()
        assert r0["fp16_overflows"] == 0 and abs(r0["weight_absmax"] - rng["weight_absmax"]) <= 1e-6 * rng["weight_absmax"]
        rng = h.set_weights(bad)
        assert rng["finite"] and rng["weight_absmax"] > 1e4                  # reported when the weights are handed over
        p, v, o = h.main_prediction(x)
        r1 = h.net_range()
        print(f"{prec}: overflow events {r1['fp16_overflows']}, |w|max {r1['weight_absmax']:.3g}")
        if prec == "f32":
            assert r1["fp16_overflows"] == 0 and np.isfinite(p).all() and np.isfinite(v).all() and np.isfinite(o).all()
        else:
            assert r1["fp16_overflows"] > 0
            msg = h.ctx.lib.tg_last_error(h.ctx.h).decode()
            assert "fp16 overflow" in msg and "65504" in msg
        h.ctx.close()


def test_fp16_range_guard_in_the_fused_attention_block():
    """The same for the shipped MainNetwork under f32x3: an attention block's BatchNorm scale x 1e5 makes k_attention_x3 round values
    beyond fp16 into the next residual block's split input -- counted; untouched weights are not."""
    from transgo_amd.model import HipNetwork, random_transgo_weights, transgo_arch
    n = 24
    good = random_transgo_weights(9, 10, 128, seed=9)
    bad = dict(good)
    bad["main_network.res_conv3.bn.weight"] = good["main_network.res_conv3.bn.weight"] * np.float32(1e5)
    x = _positions(9, n, 4)
    h = HipNetwork(9, 10, 128, rows_cap=n, arch=transgo_arch(), precision="f32x3")
    h.set_weights(good); h.main_prediction(x)
    assert h.net_range()["fp16_overflows"] == 0
    h.set_weights(bad); h.main_prediction(x)
    assert h.net_range()["fp16_overflows"] > 0


def _oracle_two_moves(args):
    """(worker process, CPU only) the oracle driving the fp32 torch tower for one game: raw root visits and move of each ply."""
    seed, sims, moves, F, NB, wseed = args
    import torch
    torch.set_num_threads(1)
    from oracle.go_oracle import OracleGoEnv
    from oracle.net import TowerNetwork
    from oracle.wp_mcts import OracleSearch
    from transgo_amd import model
    net = TowerNetwork(9, 10, F, NB).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in model.random_weights(9, 10, F, NB, seed=wseed).items()})

    def ev(obs):
        with torch.no_grad():
            p, v, _ = net.main_prediction(torch.from_numpy(obs))
        return p.numpy(), v.numpy()
    o = OracleSearch(OracleGoEnv(), ev, np.random.RandomState(int(seed)), num_simulation=sims)
    out = []
    for _ in range(moves):
        a, pi, _, _ = o.search_move()
        out.append((np.array([o.root.kids[i].n if i in o.root.kids else 0 for i in range(82)]), int(a)))
        o.advance(a)
    return out


def test_split_precision_at_c2_size_vs_oracle():
    """BASELINE configs[1] at its own size under the split-precision network -- 4096 boards x 400 simulations x
    the real 6 x 128 tower in f32x3, two moves: root-visit window, no tree error, no truncated block, no fp16 overflow, and on 32
    sampled games the visit vectors against the oracle driving the torch f32 tower (a game is followed while it agrees)."""
    import multiprocessing as mp
    from transgo_amd import model
    from transgo_amd.engine import SelfPlayEngine
    G, sims, moves, F, NB, wseed = 4096, 400, 2, 128, 6, 1234
    sample = np.random.RandomState(1).choice(G, 32, replace=False)
    sample[:4] = [0, 1, G - 2, G - 1]
    seeds = np.arange(G).astype(np.uint32)
    ctx = mp.get_context("spawn")
    with ctx.Pool(min(16, len(os.sched_getaffinity(0)))) as pool:
        fut = pool.map_async(_oracle_two_moves, [(int(seeds[g]), sims, moves, F, NB, wseed) for g in sample])
        eng = SelfPlayEngine(G, num_simulation=sims, net_blocks=NB, net_filters=F, net_precision="f32x3")
        model.load_into(eng.ctx, model.random_weights(9, 10, F, NB, seed=wseed), 9, 10, F, NB)
        eng.reset(seeds)
        got, n0 = [], np.zeros(G, np.int64)
        for m in range(moves):
            eng.search()
            vis, rn, pl, st, _ = eng.root_info(obs=False)
            assert (rn >= n0 + sims).all() and (rn < n0 + sims + 4).all(), m
            acts, _ = eng.choose_moves(vis, st)
            got.append((vis[sample].copy(), acts[sample].copy()))
            eng.play(acts)
            n0 = np.array([vis[g, acts[g]] for g in range(G)], np.int64)
        s = eng.stats()
        assert s["errors"] == 0 and s["truncated_blocks"] == 0 and s["fp16_overflows"] == 0 and s["sims"] >= G * sims * moves
        eng.close()
        want = fut.get(timeout=600)
    same = total = 0
    for k in range(len(sample)):
        for m in range(moves):
            total += 1
            ok = bool((want[k][m][0] == got[m][0][k]).all() and want[k][m][1] == got[m][1][k])
            same += int(ok)
            if not ok:
                total += moves - 1 - m                       # the rest of a diverged game counts as different
                break
    print(f"f32x3 at C2 size: identical visit-count vectors {same}/{total} on {len(sample)} sampled games")
    assert same >= 0.85 * total
