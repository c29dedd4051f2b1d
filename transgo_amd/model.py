"""Weights for the HIP network: state_dict -> BatchNorm-folded, kernel-layout float32 blob, and the reference model
surface (`main_prediction / get_weights / set_weights`, model.py:11-27) on top of libtransgo_hip.so.

Architecture = the tower BASELINE.json parametrises as "N-block x F-filter", assembled from the reference's own building
blocks (CNNBlock model.py:317-324, pre-activation ResidualBlock model.py:238-248, tail BN model.py:62/94 and the two heads
model.py:65-76/97-111).  state_dict key names follow MainNetwork's (model.py:49-76), residual blocks numbered
`res_blocks.{i}`:

    main_network.conv1.conv.0.{weight,bias}            conv3x3 C->F           main_network.conv1.conv.1.*   BatchNorm
    main_network.res_blocks.{i}.batchnormlize_1.*      BN1 (pre-activation)   ...conv_1.{weight,bias}       conv3x3 F->F
    main_network.res_blocks.{i}.batchnormlize_2.*      BN2                    ...conv_2.{weight,bias}       conv3x3 F->F
    main_network.bn_res_end.*                          tail BN
    main_network.conv_val_own.conv.{0,1}.*             conv3x3 F->2 + BN      main_network.fc_val_own / fc_val / fc_own
    main_network.conv_act.conv.{0,1}.*                 conv3x3 F->4 + BN      main_network.fc_act

Blob layout (float32, in this order; `t` = ky*3+kx, convs stored [t][cout][cin] so a cout row is contiguous in cin):
    stem   W[9][F][16] (cin zero-padded to 16, BN folded)      b[F]
    block  s1[F] t1[F] | W1[9][F][F] (BN2 folded) b1[F] | W2[9][F][F] b2[F]        x N
    tail   s_end[F] t_end[F]
    [+P]   policy-head attention: Wqkv[1.5F][F] (rows q|k|v) b[1.5F] gamma[1] s[F] t[F] | policy conv W[9][16][F] (cout 2-5) b[16]
    head   W[9][16][F] (cout 0-1 value/own conv, 2-5 policy conv unless +P, rest zero; their BNs folded)  b[16]
    ('A' layers in the trunk are stored like the policy-head attention block, in place of a residual block)
    fc     W_vo^T[2P][64] b[64] | w_v[64] b[1] | W_own^T[64][P] b[P] | W_act^T[4P][A] b[A]
"""
import ctypes

import numpy as np

from . import _lib

EPS = 1e-5   # torch.nn.BatchNorm2d default


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64) if hasattr(t, "detach") else np.asarray(t, np.float64)


def _np32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=False) if hasattr(t, "detach") else np.asarray(t, np.float32)


def _bn(sd, prefix):
    g, b = _np(sd[prefix + ".weight"]), _np(sd[prefix + ".bias"])
    m, v = _np(sd[prefix + ".running_mean"]), _np(sd[prefix + ".running_var"])
    s = g / np.sqrt(v + EPS)
    return s, b - m * s


def _conv_t(w):
    """torch [co][ci][ky][kx] -> [t][co][ci]"""
    co, ci = w.shape[:2]
    return np.transpose(w.reshape(co, ci, 9), (2, 0, 1))


class Arch:
    """Layer program of the trunk: kinds 'R' (pre-activation ResidualBlock, model.py:238-248) / 'A' (Self_Attention,
    model.py:288-315) with the state_dict module name of each layer, and the name of the policy-head attention if any."""

    def __init__(self, kinds, names, policy_attention=None):
        assert len(kinds) == len(names) and set(kinds) <= {"R", "A"}
        self.kinds, self.names, self.policy_attention = kinds, list(names), policy_attention

    @property
    def code(self):
        return self.kinds + ("+P" if self.policy_attention else "")

    @property
    def blocks(self):
        return self.kinds.count("R")


def tower_arch(blocks):
    """BASELINE.json's N-block tower."""
    return Arch("R" * blocks, [f"res_blocks.{i}" for i in range(blocks)])


def transgo_arch():
    """The shipped MainNetwork (model.py:49-76): res_conv2 .. res_conv13 with Self_Attention at 3, 7, 12 and in the policy head."""
    return Arch("RARRRARRRRAR", [f"res_conv{i}" for i in range(2, 14)], policy_attention="attention_act")


def pack_weights(sd, board_size, encode_dim, filters, blocks=None, prefix="main_network.", arch=None):
    S, C, F = board_size, encode_dim, filters
    arch = arch or tower_arch(blocks)
    P, A = S * S, S * S + 1
    out = []
    g = lambda k: _np(sd[prefix + k])
    g32 = lambda k: _np32(sd[prefix + k])            # tensors that are copied unchanged skip the float64 round trip

    def attention(name):
        w = np.concatenate([g(name + f".{k}_conv.weight").reshape(-1, F) for k in ("query", "key", "value")], 0)
        b = np.concatenate([g(name + f".{k}_conv.bias") for k in ("query", "key", "value")])
        s, t = _bn(sd, prefix + name + ".bn")
        return [w, b, g(name + ".gamma").reshape(1), s, t]          # [1.5F][F] rows q|k|v, bias, gamma, BN scale/shift

    def head_conv(pairs):
        wh = np.zeros((9, 16, F)); bh = np.zeros(16)
        for name, lo, n in pairs:
            s, t = _bn(sd, prefix + name + ".conv.1")
            w = g(name + ".conv.0.weight") * s[:, None, None, None]
            wh[:, lo:lo + n, :] = _conv_t(w)
            bh[lo:lo + n] = g(name + ".conv.0.bias") * s + t
        return [wh, bh]

    # stem: conv + BN folded
    s, t = _bn(sd, prefix + "conv1.conv.1")
    w = g("conv1.conv.0.weight") * s[:, None, None, None]
    b = g("conv1.conv.0.bias") * s + t
    wt = np.zeros((9, F, 16)); wt[:, :, :C] = _conv_t(w)
    out += [wt, b]
    for kind, name in zip(arch.kinds, arch.names):
        if kind == "A":
            out += attention(name)
            continue
        pb = name + "."
        s1, t1 = _bn(sd, prefix + pb + "batchnormlize_1")
        s2, t2 = _bn(sd, prefix + pb + "batchnormlize_2")
        w1 = g(pb + "conv_1.weight") * s2[:, None, None, None]
        b1 = g(pb + "conv_1.bias") * s2 + t2
        out += [s1, t1, _conv_t(w1), b1, _conv_t(g32(pb + "conv_2.weight")), g(pb + "conv_2.bias")]    # conv_2 is stored as is
    se, te = _bn(sd, prefix + "bn_res_end")
    out += [se, te]
    if arch.policy_attention:
        out += attention(arch.policy_attention)
        out += head_conv([("conv_act", 2, 4)])                      # policy conv on the attention output
        out += head_conv([("conv_val_own", 0, 2)])                  # value/ownership conv on relu(bn_end(x))
    else:
        out += head_conv([("conv_val_own", 0, 2), ("conv_act", 2, 4)])
    out += [g("fc_val_own.weight").T, g("fc_val_own.bias"), g("fc_val.weight")[0], g("fc_val.bias"),
            g("fc_own.weight").T, g("fc_own.bias"), g("fc_act.weight").T, g("fc_act.bias")]
    # folds are evaluated in float64 and rounded once, on the way into the float32 blob (assignment casts)
    blob = np.empty(sum(int(np.size(a)) for a in out), np.float32)
    o = 0
    for a in out:
        n = int(np.size(a))
        blob[o:o + n] = np.asarray(a).reshape(-1)
        o += n
    want = _lib.load().tg_net_blob_floats_arch(S, C, F, arch.code.encode())
    if blob.size != want:
        raise ValueError(f"packed {blob.size} floats, library expects {want} for S={S} C={C} F={F} arch={arch.code}")
    return blob


def load_into(ctx, sd, board_size, encode_dim, filters, blocks=None, rows_cap=0, arch=None):
    """pack + tg_net_load_arch.  Returns the range of what was loaded, {"weight_absmax", "finite"} of the BN-folded blob: the
    reference hands over arbitrary trained checkpoints (model.py:23-27) and runs them in f32; net_precision 1 / 2 store weights as
    fp16 (|w| > 65504 -> inf) and every fp16-carrying mode can overflow its activations (tg_net_range counts that at run time), so
    a blob that is not finite, or too large for the context's precision, is reported at once (warnings.warn) instead of surfacing
    later as NaN priors in the tree."""
    arch = arch or tower_arch(blocks)
    blob = pack_weights(sd, board_size, encode_dim, filters, arch=arch)
    ctx.call("tg_net_load_arch", arch.code.encode(), blob.ctypes.data_as(ctypes.c_void_p), blob.size, rows_cap)
    return weight_range(blob, int(ctx.cfg.net_precision))


def weight_range(blob, net_precision=0):
    import warnings
    amax = float(np.max(np.abs(blob))) if blob.size else 0.0
    rng = {"weight_absmax": amax, "finite": bool(np.isfinite(amax))}
    if not rng["finite"]:
        warnings.warn("network weights contain inf/NaN after BatchNorm folding: every evaluation will be NaN", RuntimeWarning)
    elif net_precision in (1, 2) and amax > 65504.0:
        warnings.warn(f"BN-folded weights reach |w| = {amax:.4g}, beyond fp16 (net_precision {net_precision} stores them as fp16): "
                      "use inference_dtype 'f32' or 'f32x3' for this checkpoint", RuntimeWarning)
    return rng


# tg_config.net_precision: "f16" = fp16 weights/activations in HBM and LDS with f32 accumulation and an f32 residual stream
# (BASELINE config 5); attention-free towers with 128 or 256 filters only -- anything else is refused by tg_net_load.  "f32x3" also
# takes attention layers at 9x9 (the shipped MainNetwork; its attention blocks run as one fused kernel each at 128 filters).
PRECISIONS = {"f32": 0, "f16": 1, "f16r": 2, "f32x3": 3}     # f16r: fp16 residual stream as well; f32x3: split precision (both opt-in)


class HipNetwork:
    """TransGoNetwork surface (model.py:11-27) for inference: main_prediction(x) -> (policy, value, own) as NumPy."""

    def __init__(self, board_size=9, encode_dim=10, filters=128, blocks=6, rows_cap=1024, device=0, arch=None,
                 precision="f32"):
        cfg = _lib.default_config()
        cfg.board_size, cfg.encode_dim, cfg.net_filters, cfg.net_blocks, cfg.n_games, cfg.device = \
            board_size, encode_dim, filters, blocks, 0, device
        cfg.net_precision = PRECISIONS[precision]
        self.ctx = _lib.Context(cfg)
        self.S, self.C, self.F, self.NB, self.rows_cap = board_size, encode_dim, filters, blocks, rows_cap
        self.arch = arch or tower_arch(blocks)
        self._weights = None

    def set_weights(self, weights):                      # model.py:26-27
        self._weights = weights
        self.weight_range = load_into(self.ctx, weights, self.S, self.C, self.F, rows_cap=self.rows_cap, arch=self.arch)
        return self.weight_range

    def net_range(self):
        """tg_net_range: {"fp16_overflows": sticky count of output tiles that rounded a value beyond +-65504 to fp16 (0 with
        precision "f32"), "weight_absmax": largest |w| of the live BN-folded blob as the device sees it}."""
        n = ctypes.c_uint64(); w = ctypes.c_float()
        self.ctx.call("tg_net_range", ctypes.byref(n), ctypes.byref(w))
        return {"fp16_overflows": int(n.value), "weight_absmax": float(w.value)}

    def get_weights(self):                               # model.py:23-24
        return self._weights

    def main_prediction(self, x):                        # model.py:17-20
        x = np.ascontiguousarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, np.float32)
        n, P = x.shape[0], self.S * self.S
        pol = np.empty((n, P + 1), np.float32); val = np.empty((n,), np.float32); own = np.empty((n, P), np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.ctx.call("tg_net_predict", p(x), n, p(pol), p(val), p(own))
        return pol, val.reshape(n, 1), own


def random_weights(board_size=9, encode_dim=10, filters=128, blocks=6, seed=1234):
    """Synthetic random-init weights of the tower (no checkpoint exists offline): PyTorch's default bounds
    (uniform +-1/sqrt(fan_in) for conv/linear weights and biases), BatchNorm gamma ~ 1+0.1N, beta ~ 0.1N,
    running_mean ~ 0.1N, running_var ~ U(0.5, 1.5).  Returns a state_dict of float32 NumPy arrays with the key
    names documented at the top of this file."""
    rs = np.random.RandomState(seed)
    S, C, F = board_size, encode_dim, filters
    P, A = S * S, S * S + 1
    sd = {}

    def conv(name, cin, cout):
        b = 1.0 / np.sqrt(cin * 9)
        sd[name + ".weight"] = rs.uniform(-b, b, (cout, cin, 3, 3)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-b, b, (cout,)).astype(np.float32)

    def lin(name, cin, cout):
        b = 1.0 / np.sqrt(cin)
        sd[name + ".weight"] = rs.uniform(-b, b, (cout, cin)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-b, b, (cout,)).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.1 * rs.randn(c)).astype(np.float32)
        sd[name + ".bias"] = (0.1 * rs.randn(c)).astype(np.float32)
        sd[name + ".running_mean"] = (0.1 * rs.randn(c)).astype(np.float32)
        sd[name + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.int64(0)

    p = "main_network."
    conv(p + "conv1.conv.0", C, F); bn(p + "conv1.conv.1", F)
    for i in range(blocks):
        q = p + f"res_blocks.{i}."
        bn(q + "batchnormlize_1", F); conv(q + "conv_1", F, F); bn(q + "batchnormlize_2", F); conv(q + "conv_2", F, F)
    bn(p + "bn_res_end", F)
    conv(p + "conv_val_own.conv.0", F, 2); bn(p + "conv_val_own.conv.1", 2)
    lin(p + "fc_val_own", 2 * P, 64); lin(p + "fc_val", 64, 1); lin(p + "fc_own", 64, P)
    conv(p + "conv_act.conv.0", F, 4); bn(p + "conv_act.conv.1", 4)
    lin(p + "fc_act", 4 * P, A)
    return sd


def random_transgo_weights(board_size=9, encode_dim=10, filters=128, seed=1234):
    """Synthetic random-init state_dict of the reference's shipped MainNetwork (model.py:41-114: res_conv2..res_conv13 with
    Self_Attention at 3, 7, 12, attention policy head), same distributions as random_weights; gamma ~ U(0.5, 1.5) so the
    attention branch is live (the reference initialises gamma to 0, which would make every attention block a no-op)."""
    rs = np.random.RandomState(seed)
    S, C, F = board_size, encode_dim, filters
    P, A = S * S, S * S + 1
    sd = {}

    def conv(name, cin, cout, k=3):
        b = 1.0 / np.sqrt(cin * k * k)
        sd[name + ".weight"] = rs.uniform(-b, b, (cout, cin, k, k)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-b, b, (cout,)).astype(np.float32)

    def lin(name, cin, cout):
        b = 1.0 / np.sqrt(cin)
        sd[name + ".weight"] = rs.uniform(-b, b, (cout, cin)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-b, b, (cout,)).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.1 * rs.randn(c)).astype(np.float32)
        sd[name + ".bias"] = (0.1 * rs.randn(c)).astype(np.float32)
        sd[name + ".running_mean"] = (0.1 * rs.randn(c)).astype(np.float32)
        sd[name + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.int64(0)

    def att(name):
        conv(name + ".query_conv", F, F // 4, 1); conv(name + ".key_conv", F, F // 4, 1); conv(name + ".value_conv", F, F, 1)
        sd[name + ".gamma"] = rs.uniform(0.5, 1.5, 1).astype(np.float32)
        bn(name + ".bn", F)

    p = "main_network."
    conv(p + "conv1.conv.0", C, F); bn(p + "conv1.conv.1", F)
    for i in range(2, 14):
        q = p + f"res_conv{i}"
        if i in (3, 7, 12):
            att(q)
        else:
            bn(q + ".batchnormlize_1", F); conv(q + ".conv_1", F, F); bn(q + ".batchnormlize_2", F); conv(q + ".conv_2", F, F)
    bn(p + "bn_res_end", F)
    conv(p + "conv_val_own.conv.0", F, 2); bn(p + "conv_val_own.conv.1", 2)
    lin(p + "fc_val_own", 2 * P, 64); lin(p + "fc_val", 64, 1); lin(p + "fc_own", 64, P)
    att(p + "attention_act")
    conv(p + "conv_act.conv.0", F, 4); bn(p + "conv_act.conv.1", 4)
    lin(p + "fc_act", 4 * P, A)
    return sd
