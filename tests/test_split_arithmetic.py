"""The arithmetic of the opt-in split-precision mode (net_precision 3, transgo_amd/csrc/net.hip: k_conv3x3_h2<..., X2>,
k_attention_x3), restated in NumPy: operands carried as fp16 hi + lo = half(x - hi), weights scaled by a power of two so that their
lo halves stay out of the fp16 subnormals, THREE of the four partial products (w_hi a_hi + w_hi a_lo + w_lo a_hi; w_lo a_lo is
dropped) accumulated in f32.  Pins the claim the GPU tests then check on the real kernels: the result sits within a few f32 roundings of the
exact dot product (measured here: 1.3e-7 of sum|w||a| against 4.9e-8 for an f32 dot product), the dropped term being one of them (5e-8).  No GPU, no library."""
import numpy as np


def split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def scale_exp(w):
    """k_restage_split's scale: 2^s with the largest |w| brought to [2^14, 2^15)."""
    m = float(np.abs(w).max())
    return 0 if m == 0.0 else 14 - int(np.floor(np.log2(m)))


def dot3(w, a):
    s = scale_exp(w)
    wh, wl = split(np.ldexp(w, s).astype(np.float32))
    ah, al = split(a)
    f = lambda v: v.astype(np.float32).astype(np.float64)          # fp16 x fp16 products are exact in f32; sum them in f64 here
    acc = (f(wh) * f(ah)).sum() + (f(wh) * f(al)).sum() + (f(wl) * f(ah)).sum()
    dropped = (f(wl) * f(al)).sum()
    return np.ldexp(acc, -s), np.ldexp(dropped, -s)


def test_three_products_are_f32_accurate():
    rng = np.random.RandomState(7)
    worst3, worst32, worst_drop = 0.0, 0.0, 0.0
    for k in (16, 144, 1152, 2304):                                # K of a stem tap set, a 128- and a 256-filter conv
        for _ in range(40):
            w = (rng.randn(k) * 0.05).astype(np.float32)
            a = np.maximum(rng.randn(k) * 1.5 + 0.3, 0).astype(np.float32)        # post-ReLU activations
            exact = float((w.astype(np.float64) * a.astype(np.float64)).sum())
            scale = float((np.abs(w).astype(np.float64) * np.abs(a)).sum()) + 1e-30
            got, dropped = dot3(w, a)
            f32dot = float(np.dot(w, a))                                           # what an f32 accumulation gives
            worst3 = max(worst3, abs(got - exact) / scale)
            worst32 = max(worst32, abs(f32dot - exact) / scale)
            worst_drop = max(worst_drop, abs(dropped) / scale)
    print(f"three-product split dot: {worst3:.2e} of sum|w||a| (f32 dot product: {worst32:.2e}); dropped lo*lo term {worst_drop:.2e}")
    assert worst3 < 2.0 ** -20                                     # ~22 significand bits per operand
    assert worst_drop < 2.0 ** -21                                 # what dropping w_lo a_lo costs: below the bound above


def test_split_reconstructs_operands_to_22_bits_and_scaling_keeps_lo_normal():
    rng = np.random.RandomState(8)
    w = (rng.randn(4096) * 0.02).astype(np.float32)
    s = scale_exp(w)
    ws = np.ldexp(w, s).astype(np.float32)
    hi, lo = split(ws)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    rel = np.abs(rec - ws) / np.maximum(np.abs(ws), 1e-30)
    big = np.abs(ws) > 2.0 ** 2                                    # elements whose lo half is a normal fp16 number
    assert rel[big].max() < 2.0 ** -21
    assert np.abs(ws).max() < 65504 and np.isfinite(hi.astype(np.float32)).all()
    # an unscaled small weight would lose its lo half to the subnormals: the reason for the per-conv scale
    h0, l0 = split(w)
    rel0 = np.abs(h0.astype(np.float64) + l0.astype(np.float64) - w) / np.maximum(np.abs(w), 1e-30)
    assert np.median(rel0) > np.median(rel) * 4
