"""The C-ABI library loads without a GPU and exports every symbol include/transgo_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

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


def test_reference_go_env_symbols_exported():
    """The 15 names of GoEnv/cpp_src/go_env.h:24-70, so the reference's own environment.py can bind this library."""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in ("Init Reset Step Step_ checkAction isTerminated Encode getScore getTerritory getLegalAction getLegalNoEye "
              "Show getPlayer getStep getSubEncode").split():
        assert hasattr(lib, n), n


def test_get_sub_encode_crops_like_the_reference():
    """getSubEncode is host-side word movement (board.cc:1166-1271): 4 corner windows then the centre."""
    import numpy as np
    lib = ctypes.CDLL(_lib.LIB_PATH)
    enc = np.arange(10 * 81, dtype=np.float32).reshape(10, 9, 9)
    sub = np.zeros((5, 10, 7, 7), np.float32)
    lib.getSubEncode(enc.ctypes.data_as(ctypes.c_void_p), sub.ctypes.data_as(ctypes.c_void_p), 7, 10, 5)
    assert (sub[0] == enc[:, :7, :7]).all() and (sub[1] == enc[:, :7, 2:]).all() and (sub[2] == enc[:, 2:, :7]).all()
    assert (sub[3] == enc[:, 2:, 2:]).all() and (sub[4] == enc[:, 1:8, 1:8]).all()


def test_get_sub_encode_equals_the_compiled_reference():
    """Same call into oracle/_ref (the reference engine compiled in place by `make -C oracle ref`; it travels to the GPU box as a
    built artefact) and into this library, for every cut count and two window sizes."""
    import numpy as np
    ref_so = os.path.join(ROOT, "oracle", "_ref", "GoEnv", "go_env.so")
    if not os.path.exists(ref_so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref, lib = ctypes.CDLL(ref_so), ctypes.CDLL(_lib.LIB_PATH)
    rs = np.random.RandomState(3)
    for sub_size in (7, 5):
        for cuts in (1, 4, 5):
            enc = (rs.rand(10, 9, 9) < 0.3).astype(np.float32)
            a = np.full((5, 10, sub_size, sub_size), -1, np.float32); b = a.copy()
            for L, out in ((ref, a), (lib, b)):
                L.getSubEncode(enc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), sub_size, 10, cuts)
            assert np.array_equal(a, b), (sub_size, cuts)


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: with the shared library absent the package raises at first use instead of computing anything."""
    import subprocess, sys
    code = ("import sys; from transgo_amd import _lib; _lib.LIB_PATH = sys.argv[1]\n"
            "try:\n    _lib.load()\nexcept _lib.TransgoError as e:\n    print('RAISED', 'not found' in str(e))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, str(tmp_path / "nope.so")], capture_output=True, text=True, cwd=root, timeout=120)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_creating_a_context_without_a_gpu_fails_loudly():
    """On a machine without a GPU tg_create returns an error that the Python wrapper raises (skipped where a GPU exists)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from transgo_amd import _lib
    with pytest.raises(_lib.TransgoError):
        _lib.Context(_lib.default_config())


def test_device_code_has_no_out_of_line_calls(tmp_path):
    """Round 3 (DESIGN.md, 19x19 fault): one out-of-line device function call (k_play<19> -> the ballot-packed encode_bits<19>)
    corrupted state silently and made the next kernel fault; the same body inlined is fine.  Since then every device helper is
    __forceinline__ and this test keeps it that way: no s_swappc_b64 (call) in any gfx950 code object of the shipped library."""
    import glob
    import shutil
    import struct
    import subprocess
    # llvm-objdump through ROCM_PATH / hipconfig / PATH; a library that holds device code but cannot be disassembled FAILS the
    # test (this check is the only guard against the fault's return: it must not turn itself off silently)
    cands = [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump")]
    try:
        hp = subprocess.run(["hipconfig", "--rocmpath"], stdout=subprocess.PIPE, text=True, timeout=30).stdout.strip()
        if hp:
            cands.append(os.path.join(hp, "lib", "llvm", "bin", "llvm-objdump"))
    except (OSError, subprocess.SubprocessError):
        pass
    cands += sorted(glob.glob("/opt/rocm*/lib/llvm/bin/llvm-objdump")) + [shutil.which("llvm-objdump") or ""]
    objdump = next((c for c in cands if c and os.path.exists(c)), None)
    data = open(_lib.LIB_PATH, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    assert magic in data, "the shipped library holds no device code bundle"
    assert objdump, "llvm-objdump not found (ROCM_PATH, hipconfig --rocmpath, /opt/rocm*, PATH): the library's device code cannot be checked"
    pos, n_objs, calls, targets = 0, 0, 0, set()
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        nb = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(nb):
            o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + tl].decode(); off += tl
            if "amdgcn" in triple and sz:                    # EVERY device target of the bundle, whatever ARCH the Makefile was given
                f = tmp_path / f"co{n_objs}.o"
                f.write_bytes(data[i + o:i + o + sz])
                out = subprocess.run([objdump, "-d", str(f)], stdout=subprocess.PIPE, text=True, check=True).stdout
                assert "s_endpgm" in out, f"{triple}: disassembly shows no kernel"
                calls += out.count("s_swappc_b64")
                n_objs += 1
                targets.add(triple.split("--")[-1])
        pos = i + 24
    assert targets, "no amdgcn code object in the library"
    assert n_objs >= 4 * len(targets), f"expected one code object per .hip source and target, found {n_objs} for {sorted(targets)}"
    assert calls == 0, f"{calls} out-of-line device calls in the library"
