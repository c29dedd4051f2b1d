"""The C-ABI library loads without a GPU and exports every symbol include/transgo_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from transgo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "transgo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_header():
    assert set(declared_symbols()) <= set(_lib.SIGNATURES) | {"tg_config_default"}, \
        set(declared_symbols()) - set(_lib.SIGNATURES)


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(_lib.TransgoError):
        _lib.Context(_lib.default_config())


def test_config_struct_matches_header_defaults():
    cfg = _lib.default_config()
    assert (cfg.board_size, cfg.encode_dim, cfg.max_step, cfg.komi) == (9, 10, 120, 7.5)
    assert (cfg.num_simulation, cfg.parallel_readouts, cfg.wu_loss, cfg.c_puct1, cfg.c_puct2) == (210, 4, 2, 3.0, 0.05)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "transgo_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "go_oracle" not in txt, f
