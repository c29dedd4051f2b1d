"""HIP network forward (net.hip through tg_net_predict) against the fp32 torch oracle tower on real encoded positions.
Tolerance 1e-3 absolute on policy probabilities, value and ownership (BASELINE.json north_star); observed ~1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3


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
            if rng.rand() < 0.5:
                obs.append(env.encode(s))
    return np.stack(obs)


@pytest.mark.parametrize("S,F,NB,n", [(9, 32, 2, 300), (9, 128, 6, 257), (9, 64, 3, 5), (19, 128, 2, 9),
                                       (9, 256, 2, 131), (19, 256, 2, 7)])
def test_tower_matches_torch(S, F, NB, n):
    import torch
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(4)
    net = seeded_tower(S, 10, F, NB, seed=1234 + F)
    x = _positions(S, n, 3)
    with torch.no_grad():
        p, v, o = net.main_prediction(torch.from_numpy(x))
    h = HipNetwork(S, 10, F, NB, rows_cap=128)          # smaller than n: exercises chunking
    h.set_weights(net.get_weights())
    hp, hv, ho = h.main_prediction(x)
    ep, ev, eo = np.abs(hp - p.numpy()).max(), np.abs(hv - v.numpy()).max(), np.abs(ho - o.numpy()).max()
    print(f"S={S} F={F} N={NB}: max abs err policy {ep:.2e} value {ev:.2e} own {eo:.2e}")
    assert ep < TOL and ev < TOL and eo < TOL
    assert np.allclose(hp.sum(1), 1.0, atol=1e-5)
    # single-row call equals the batched rows (tile boundaries / halo rows do not leak between positions)
    k = min(7, n - 1)
    hp1, hv1, _ = h.main_prediction(x[k:k + 1])
    assert np.array_equal(hp1[0], hp[k]) and np.array_equal(hv1[0], hv[k])


def test_engine_with_network_smoke():
    """Whole loop with the network as evaluator: every game completes its simulations, pi is a distribution."""
    from oracle.net import seeded_tower
    from transgo_amd.engine import SelfPlayEngine
    from transgo_amd import model
    G, sims = 48, 32
    eng = SelfPlayEngine(G, num_simulation=sims, net_blocks=2, net_filters=32)
    model.load_into(eng.ctx, seeded_tower(9, 10, 32, 2).get_weights(), 9, 10, 32, 2)
    eng.reset(np.arange(G))
    for m in range(3):
        eng.search()
        vis, rn, pl, st, ob = eng.root_info()
        assert (vis.sum(1) >= sims - 1).all() and (st == m + 1).all()
        acts, pis = eng.choose_moves(vis, st)
        assert np.allclose(pis.sum(1), 1.0)
        eng.play(acts)
    s = eng.stats()
    assert s["errors"] == 0 and s["sims"] >= 3 * G * sims


def test_reference_mainnetwork_with_attention(golden_dir):
    """The shipped MainNetwork layout (9 residual + 3 attention blocks, attention policy head, model.py:49-76) with the
    reference's own state_dict, against outputs recorded from the imported reference model."""
    import os
    from transgo_amd.model import HipNetwork, transgo_arch
    with np.load(os.path.join(golden_dir, "net_transgo_f32.npz")) as z:
        b = {k: z[k] for k in z.files}
    sd = {k[3:]: v for k, v in b.items() if k.startswith("sd/")}
    h = HipNetwork(9, 10, 32, rows_cap=8, arch=transgo_arch())
    h.set_weights(sd)
    hp, hv, ho = h.main_prediction(b["x"])
    ep, ev, eo = np.abs(hp - b["policy"]).max(), np.abs(hv - b["value"]).max(), np.abs(ho - b["own"]).max()
    print(f"MainNetwork F=32: max abs err policy {ep:.2e} value {ev:.2e} own {eo:.2e}")
    assert ep < TOL and ev < TOL and eo < TOL


@pytest.mark.parametrize("F", [128, 256, 64])
def test_mainnetwork_default_width_vs_oracle(F):
    """F=128 (the reference default, configure.py:37) against the torch restatement that the fixture above pins; 128 and 256 run
    the residual blocks on the DMA-fed k_conv3x3_sg chain and Self_Attention on the matrix cores (k_attention_mfma), 64 the
    general kernels."""
    import torch
    from oracle.net import TransGoMain
    from transgo_amd.model import HipNetwork, transgo_arch
    torch.manual_seed(3); torch.set_num_threads(4)
    net = TransGoMain(9, 10, F).eval()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            if hasattr(m, "gamma"):
                m.gamma.copy_(0.5 + torch.rand(1, generator=g))
    x = _positions(9, 70, 11)
    with torch.no_grad():
        p, v, o = net.main_prediction(torch.from_numpy(x))
    h = HipNetwork(9, 10, F, rows_cap=64, arch=transgo_arch())
    h.set_weights({k: t.numpy() for k, t in net.state_dict().items()})
    hp, hv, ho = h.main_prediction(x)
    ep, ev, eo = np.abs(hp - p.numpy()).max(), np.abs(hv - v.numpy()).max(), np.abs(ho - o.numpy()).max()
    print(f"MainNetwork F={F}: max abs err policy {ep:.2e} value {ev:.2e} own {eo:.2e}")
    assert ep < TOL and ev < TOL and eo < TOL


@pytest.mark.parametrize("variant", ["0", "1"])
def test_dma_conv_path_matches_torch(monkeypatch, variant):
    """Both F->F conv paths of the f32 F=128 tower: TG_DMA_CONV=1 (default) the DMA-fed chain (k_conv3x3_sg: weight ring by LDS-DMA,
    B fragments straight from L2, slice-major conv inputs, prologue-free); =0 the general register-staged kernel (k_conv3x3)."""
    import torch
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    monkeypatch.setenv("TG_DMA_CONV", variant)
    torch.set_num_threads(4)
    for S, NB, n in ((9, 3, 150), (19, 1, 5)):
        net = seeded_tower(S, 10, 128, NB, seed=99)
        x = _positions(S, n, 8)
        with torch.no_grad():
            p, v, o = net.main_prediction(torch.from_numpy(x))
        h = HipNetwork(S, 10, 128, NB, rows_cap=64)
        h.set_weights(net.get_weights())
        hp, hv, ho = h.main_prediction(x)
        assert np.abs(hp - p.numpy()).max() < TOL and np.abs(hv - v.numpy()).max() < TOL and np.abs(ho - o.numpy()).max() < TOL


@pytest.mark.parametrize("S,F,NB,n", [(9, 256, 3, 70), (19, 256, 2, 5), (9, 128, 4, 40), (19, 128, 1, 3)])
def test_fp16_chain_matches_half_storage_oracle(S, F, NB, n):
    """BASELINE config 5 ("fp16 policy/value inference"): fp16 weights/activations, f32 accumulate (k_conv3x3_h).  Checked
    against the oracle's emulation with the same rounding points (only the summation order inside a conv is free: 1e-3),
    and reported / bounded against the plain f32 network (expect <~5e-3 on the probabilities)."""
    import torch
    from oracle.net import half_storage_forward, seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(4)
    net = seeded_tower(S, 10, F, NB, seed=77 + F)
    x = _positions(S, n, 5)
    p16, v16, o16 = [t.numpy() for t in half_storage_forward(net, torch.from_numpy(x))]
    with torch.no_grad():
        p32, v32, o32 = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
    h = HipNetwork(S, 10, F, NB, rows_cap=32, precision="f16")        # smaller than n: chunking; 32 boards < one 256-row tile at 9x9
    h.set_weights(net.get_weights())
    hp, hv, ho = h.main_prediction(x)
    e16 = [np.abs(a - b).max() for a, b in ((hp, p16), (hv, v16), (ho, o16))]
    e32 = [np.abs(a - b).max() for a, b in ((hp, p32), (hv, v32), (ho, o32))]
    print(f"fp16 S={S} F={F} N={NB}: vs half-storage oracle policy {e16[0]:.2e} value {e16[1]:.2e} own {e16[2]:.2e}; "
          f"vs f32 network policy {e32[0]:.2e} value {e32[1]:.2e} own {e32[2]:.2e}")
    assert max(e16) < TOL
    assert e32[0] < 5e-3 and e32[1] < 2e-2 and e32[2] < 2e-2
    assert np.allclose(hp.sum(1), 1.0, atol=1e-5)


def test_fp16_refused_where_not_built():
    from transgo_amd._lib import TransgoError
    from transgo_amd.model import HipNetwork, random_weights
    h = HipNetwork(9, 10, 64, 2, rows_cap=8, precision="f16")
    with pytest.raises(TransgoError):
        h.set_weights(random_weights(9, 10, 64, 2))


@pytest.mark.parametrize("precision,F", [("f32", 128), ("f16", 128), ("f32", 32)])
def test_background_weight_refresh_switches_sets_cleanly(precision, F):
    """tg_net_load_async (trainer.py:76-79 -> self_play.py:913): the refresh goes into the idle weight set on a
    side stream; until it is adopted every forward still computes with the old weights, afterwards with exactly the new ones
    (bit-identical to a context that loaded them synchronously).  Three refreshes in a row reuse both sets."""
    import ctypes
    from transgo_amd import model
    from transgo_amd.model import HipNetwork
    x = _positions(9, 40, 17)
    sds = [model.random_weights(9, 10, F, 2, seed=s) for s in (1, 2, 3, 4)]
    want = []
    for sd in sds:
        h = HipNetwork(9, 10, F, 2, rows_cap=64, precision=precision)
        h.set_weights(sd)
        want.append(h.main_prediction(x))
        h.ctx.close()
    h = HipNetwork(9, 10, F, 2, rows_cap=64, precision=precision)
    h.set_weights(sds[0])
    same = lambda a, b: all(np.array_equal(p, q) for p, q in zip(a, b))
    assert same(h.main_prediction(x), want[0])
    for k in (1, 2, 3):
        blob = model.pack_weights(sds[k], 9, 10, F, arch=h.arch)
        h.ctx.call("tg_net_load_async", h.arch.code.encode(), blob.ctypes.data_as(ctypes.c_void_p), blob.size)
        got = h.main_prediction(x)                        # may run before or after the switch, never on a half-written set
        assert same(got, want[k - 1]) or same(got, want[k])
        pend = ctypes.c_int(-1)
        h.ctx.call("tg_net_load_poll", 1, ctypes.byref(pend))
        assert pend.value == 0
        assert same(h.main_prediction(x), want[k])


@pytest.mark.parametrize("S,F,NB,n", [(9, 128, 4, 40), (19, 256, 3, 4), (9, 256, 40, 8)])
def test_fp16_residual_stream_mode(S, F, NB, n):
    """net_precision 2 ("f16r"): the residual stream is fp16 too (a third less HBM traffic per block).  Checked against the oracle's
    emulation with the same rounding points (1e-3; only the order of summation inside a conv is free) and bounded against the plain
    f32 tower (fp16 inference within ~5e-3 on probabilities) -- including a 40-block tower, where the rounding of the
    stream accumulates."""
    import torch
    from oracle.net import half_storage_forward, seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(8)
    net = seeded_tower(S, 10, F, NB, seed=900 + NB)
    x = _positions(S, n, 6)
    p16, v16, o16 = [t.numpy() for t in half_storage_forward(net, torch.from_numpy(x), half_residual=True)]
    with torch.no_grad():
        p32, v32, o32 = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
    h = HipNetwork(S, 10, F, NB, rows_cap=16, precision="f16r")
    h.set_weights(net.get_weights())
    hp, hv, ho = h.main_prediction(x)
    e16 = [float(np.abs(a - b).max()) for a, b in ((hp, p16), (hv, v16), (ho, o16))]
    e32 = [float(np.abs(a - b).max()) for a, b in ((hp, p32), (hv, v32), (ho, o32))]
    print(f"fp16-residual S={S} F={F} N={NB}: vs emulation policy {e16[0]:.2e} value {e16[1]:.2e} own {e16[2]:.2e}; "
          f"vs f32 tower policy {e32[0]:.2e} value {e32[1]:.2e} own {e32[2]:.2e}")
    assert max(e16) < 1e-3
    assert e32[0] < 5e-3 and e32[1] < 2e-2 and e32[2] < 2e-2


def test_f32_conv_both_tile_shapes():
    """k_conv3x3_sg at F = 128 is built with 192-row and 128-row tiles and the host picks per launch by the batch's round count:
    1700 positions (137 700 rows: one round of 768 x 192, two of 1024 x 128) take the large tiles, 1500 (121 500 rows) the small
    ones; both against fp32 torch, and the rows they share must agree bit for bit (the tile shape does not change the k order)."""
    import torch
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(8)
    net = seeded_tower(9, 10, 128, 2, seed=31)
    x = _positions(9, 1700, 12)
    with torch.no_grad():
        p, v, o = [t.numpy() for t in net.main_prediction(torch.from_numpy(x))]
    h = HipNetwork(9, 10, 128, 2, rows_cap=1700)
    h.set_weights(net.get_weights())
    big = h.main_prediction(x)                       # M = 137700 -> 192-row tiles
    small = h.main_prediction(x[:1500])              # M = 121500 -> 128-row tiles
    for got, n in ((big, 1700), (small, 1500)):
        assert np.abs(got[0] - p[:n]).max() < TOL and np.abs(got[1] - v[:n]).max() < TOL and np.abs(got[2] - o[:n]).max() < TOL
    assert np.array_equal(big[0][:1500], small[0]) and np.array_equal(big[1][:1500], small[1])


@pytest.mark.parametrize("S,F,NB,n", [(9, 128, 2, 1), (9, 128, 2, 213), (9, 32, 2, 70), (9, 256, 1, 45), (19, 128, 1, 5), (19, 256, 1, 3)])
def test_head_gemm_equals_the_implicit_gemm_head(monkeypatch, S, F, NB, n):
    """Round 3: the 16-wide head conv (6 real couts) runs as GEMM + col2im (k_head_gemm: every input row times all 9 taps' weights,
    then the shifted sum with the board mask) instead of an implicit GEMM that fills 6 of 16 MFMA columns.  Same products, another
    summation order: outputs within 2e-6 of the kernel it replaced (TG_HEAD_GEMM=0) on batches that end inside a tile, on one
    board, at every filter count built and at 19x19, and within 1e-3 of fp32 torch."""
    import torch
    from oracle.net import seeded_tower
    from transgo_amd.model import HipNetwork
    torch.set_num_threads(8)
    net = seeded_tower(S, 10, F, NB, seed=500 + F)
    x = _positions(S, n, 21)
    h = HipNetwork(S, 10, F, NB, rows_cap=max(8, n))
    h.set_weights(net.get_weights())
    monkeypatch.setenv("TG_HEAD_GEMM", "0")
    old = h.main_prediction(x)
    monkeypatch.setenv("TG_HEAD_GEMM", "1")
    new = h.main_prediction(x)
    e = [float(np.abs(a - b).max()) for a, b in zip(old, new)]
    print(f"head GEMM {NB}x{F}@{S}x{S} n={n}: max abs difference to the implicit-GEMM head policy {e[0]:.2e} value {e[1]:.2e} own {e[2]:.2e}")
    assert max(e) < 2e-6
    k = min(n, 32)
    with torch.no_grad():
        p, v, o = [t.numpy() for t in net.main_prediction(torch.from_numpy(x[:k]))]
    assert np.abs(new[0][:k] - p).max() < TOL and np.abs(new[1][:k] - v).max() < TOL and np.abs(new[2][:k] - o).max() < TOL
