"""libtransgo_hip's host MT19937 helpers (transgo_amd/csrc/rng_host.cpp) against NumPy's legacy RandomState:
golden draw sequences recorded by tests/golden/gen_fixtures.py rng, plus the live NumPy of this box.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

from transgo_amd import _lib


def _state(seed):
    lib = _lib.load()
    s = _lib.TgMt19937()
    lib.tg_host_mt_seed(ctypes.byref(s), seed)
    return lib, s


def _pick(lib, s, p):
    """RandomState.choice(A, p=p) internals (mtrand.pyx choice): cdf = cumsum(p); cdf /= cdf[-1];
    searchsorted(random_sample(), 'right')."""
    cdf = p.cumsum(); cdf /= cdf[-1]
    u = lib.tg_host_mt_random_sample(ctypes.byref(s))
    return int(cdf.searchsorted(u, side="right"))


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**31 - 1])
def test_golden_sequences(golden_dir, seed):
    with np.load(os.path.join(golden_dir, "rng_mt19937.npz")) as z:
        kinds, ns, vals = z[f"s{seed}_kinds"], z[f"s{seed}_n"], z[f"s{seed}_vals"]
        final_pos, final_key = int(z[f"s{seed}_final_pos"]), z[f"s{seed}_final_key"]
    lib, s = _state(seed)
    off = 0
    for kind, n in zip(kinds, ns):
        if kind == 0:
            out = np.zeros(n)
            assert lib.tg_host_mt_dirichlet(ctypes.byref(s), 0.03, int(n), out.ctypes.data_as(ctypes.c_void_p)) == 0
            assert (out == vals[off:off + n]).all(), (seed, "dirichlet", n)
            off += n
        elif kind == 1:
            assert lib.tg_host_mt_choice_index(ctypes.byref(s), int(n)) == int(vals[off]), (seed, "tie", n)
            off += 1
        else:
            p = np.zeros(82)
            assert lib.tg_host_mt_dirichlet(ctypes.byref(s), 0.5, 82, p.ctypes.data_as(ctypes.c_void_p)) == 0
            assert (p == vals[off:off + 82]).all()
            assert _pick(lib, s, p) == int(vals[off + 82])
            off += 83
    assert s.pos == final_pos
    assert (np.frombuffer(s.key, np.uint32) == final_key).all()


def test_against_live_numpy():
    for seed in (7, 99, 4242):
        lib, s = _state(seed)
        rs = np.random.RandomState(seed)
        st = rs.get_state()
        assert (np.frombuffer(s.key, np.uint32) == st[1]).all() and s.pos == st[2]
        for rep in range(200):
            n = int(rs.randint(1, 83)); k = lib.tg_host_mt_choice_index(ctypes.byref(s), 82) + 1
            assert n == k
            a = rs.dirichlet([0.03] * n)
            b = np.zeros(n); lib.tg_host_mt_dirichlet(ctypes.byref(s), 0.03, n, b.ctypes.data_as(ctypes.c_void_p))
            assert (a == b).all()
            assert rs.random_sample() == lib.tg_host_mt_random_sample(ctypes.byref(s))
            assert rs.choice(list(range(n))) == lib.tg_host_mt_choice_index(ctypes.byref(s), n)
        assert s.pos == rs.get_state()[2]
