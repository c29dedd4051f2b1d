"""Deterministic stand-in evaluators for search-parity tests.  TEST INFRASTRUCTURE ONLY.

The network cannot be bit-identical between torch and the HIP kernels, so tree parity (visit counts, chosen moves,
RNG consumption) is tested with evaluators whose outputs are exact functions of the integer feature planes: every
intermediate is an exactly representable integer and the only roundings are one f32 division per output element, so
NumPy reproduces them bit-for-bit on any machine.  `flat` produces many equal priors (stresses the tie-break RNG path
of self_play.py:709-713); `sharp` produces peaked priors (deep trees, large inherited sub-trees on re-rooting).
"""
import numpy as np


def _hash_planes(obs):
    k, C, S, _ = obs.shape
    P = S * S
    o = obs.reshape(k, C, P).astype(np.int64)
    w = np.arange(1, C + 1, dtype=np.int64)
    s = (o * w[None, :, None]).sum(1)
    tot = o.sum((1, 2))
    idx = np.arange(P, dtype=np.int64)
    h = s * 7 + idx[None, :] * 13 + tot[:, None] * 5
    v = ((s * idx[None, :]).sum(1) + tot * 3) % 17 - 8
    return h, v


def _finish(raw, v):
    k = raw.shape[0]
    raw = np.concatenate([raw, np.ones((k, 1), np.int64)], 1).astype(np.float32)
    denom = raw.sum(1, dtype=np.float64).astype(np.float32)      # integer sums < 2**24: exact
    policy = raw / denom[:, None]
    value = v.astype(np.float32) / np.float32(10)
    return policy, value.reshape(k, 1)


def flat(obs):
    h, v = _hash_planes(np.asarray(obs))
    return _finish(h % 32 + 1, v)


def sharp(obs):
    h, v = _hash_planes(np.asarray(obs))
    return _finish(np.left_shift(1, h % 11), v)


def spike(obs):
    """Almost all prior mass on the point(s) with the largest plane hash -- the shape of a trained, confident policy: visits
    pile onto one child, so nearly the whole tree is inherited on every re-rooting (arena sizing / truncation tests)."""
    h, v = _hash_planes(np.asarray(obs))
    return _finish(np.where(h == h.max(axis=1, keepdims=True), 1 << 16, 1), v)


BY_NAME = {"flat": flat, "sharp": sharp, "spike": spike}
