"""CPU-side sanitizer runs of the native code that has a CPU build: the oracle's C restatement and the C ABI's host RNG helpers,
each compiled with -fsanitize=address,undefined and driven over seeded inputs (GPU AddressSanitizer is not available)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_c_restatement_is_clean_under_asan_ubsan():
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "no sanitizer report" in out.stdout, out.stdout + out.stderr


def test_host_rng_helpers_are_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_rng_host")
    src = [os.path.join(ROOT, "tests", "native", "san_rng_host.cpp"), os.path.join(ROOT, "transgo_amd", "csrc", "rng_host.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe] + src)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "no sanitizer report" in out.stdout, out.stdout + out.stderr
