"""bench.py as the driver runs it.  `python bench.py --gpus N` with N > 1 and no rank environment is a LAUNCHER: a GPU-free parent
that times the CPU baseline, starts N ranks of its own and forwards rank 0's line (the reference's driver
spawns its workers itself, transgo.py:92-107).  On the one-GPU box the two ranks share the card and gloo carries the collectives."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout, **env):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_launches_its_own_two_ranks():
    r = _run("--gpus 2 --games 1024 --steps 3 --warmup 1".split(), 900, TRANSGO_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    print(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    rk = line["ranks"]
    assert rk["world"] == 2 and rk["backend"] == "gloo" and rk["launcher"] == "bench.py"
    assert sorted(d["rank"] for d in rk["devices"]) == [0, 1] and len({d["pid"] for d in rk["devices"]}) == 2
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["achieved"] > 0 and 0 < rf["frac"] <= 1 and rf["launches"] > 0
    assert rf["launches_not_timed"] == 0                  # the event pool is drained every step: every conv launch is in the figure
    cb = line["cpu_baseline"]
    assert cb and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and line["cpu_baseline_c1"]["value"] > 0
    assert line["value"] > 0 and line["ms_per_step"] > 0
    sg = line["selfplay_games"]
    assert sg["finished_and_stored"] == sg["finished_all_ranks"] > 0       # both ranks' finished games reached rank 0's store


def test_bench_launcher_reports_a_failing_rank():
    """Any rank that fails ends the job with a non-zero exit (no line, no hang): 48 filters is a width the library does not build."""
    r = _run("--gpus 2 --games 8 --steps 1 --warmup 0 --filters 48 --sims 8 --stagger 0 --no-cpu-baseline".split(), 300,
             TRANSGO_DIST_BACKEND="gloo")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "launcher" in r.stderr


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _run("--gpus 2 --no-cpu-baseline".split(), 120, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
