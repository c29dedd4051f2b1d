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


def test_bench_rehearses_more_than_two_ranks():
    """BASELINE configs[2]/[4] are world 8.  The rehearsal of the launcher + per-move exchange with MORE than two ranks on the one
    card: four ranks over gloo (this pool's GPU boxes end a job with more than 6 processes on the card, and pytest itself holds a
    context; the eight-rank exchange itself runs over gloo on the CPU, tests/test_host_logic.py)."""
    r = _run("--gpus 4 --games 512 --steps 2 --warmup 1 --cpu-seconds 5".split(), 900, TRANSGO_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    print(lines[0])
    rk = line["ranks"]
    assert line["n_gpus"] == 4 and rk["world"] == 4 and rk["backend"] == "gloo" and rk["launcher"] == "bench.py"
    assert len({d["pid"] for d in rk["devices"]}) == 4 and sorted(d["rank"] for d in rk["devices"]) == [0, 1, 2, 3]
    assert rk["transport"].startswith("all_gather")                        # the plain collective is the default transport
    assert rk["seeds_disjoint"] is True
    pr = rk["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1, 2, 3] and all(p["sims_per_s"] > 0 and p["ms_per_step"] > 0 for p in pr)
    assert all(p["games_finished"] > 0 for p in pr)                         # every rank finished games inside the timed steps ...
    sg = line["selfplay_games"]
    assert sg["finished_and_stored"] == sg["finished_all_ranks"] == sum(p["games_finished"] for p in pr)   # ... all in rank 0's store
    assert abs(sum(p["sims"] for p in pr) / (line["ms_per_step"] * line["steps"] * 1e-3) - line["value"]) <= 1e-3 * line["value"]
    assert "secondary" not in line                                          # N > 1: the headline leg only


def test_bench_one_gpu_line_carries_the_secondary_legs():
    """`python bench.py --gpus 1` is the same launcher: headline leg (exact-f32 tower) in a fresh child, then the f32x3 tower and the
    f32 MainNetwork on the same workload in fresh children of their own, ONE line (tiny sizes here)."""
    r = _run("--gpus 1 --games 256 --sims 32 --steps 2 --warmup 1 --stagger 8 --cpu-seconds 3".split(), 900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    print(lines[0])
    assert line["dtype"] == "f32" and line["n_gpus"] == 1 and "6-block x 128-filter tower" in line["config"]["workload"]
    assert line["ranks"]["launcher"] == "bench.py" and line["ranks"]["world"] == 1 and line["cpu_baseline"]["value"] > 0
    sec = line["secondary"]
    x3, mn, g4 = sec["f32x3"], sec["mainnetwork_f32"], sec["f32_groups4"]
    for leg in (x3, mn, g4):
        assert "error" not in leg, leg
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and leg["steps"] == 2 and leg["warmup"] == 1 and leg["tree_errors"] == 0
        assert leg["roofline"]["achieved"] > 0 and 0 < leg["roofline"]["frac"] <= 1 and leg["roofline"]["launches_not_timed"] == 0
        assert leg["ms_per_step"] * leg["steps"] * 1e-3 < leg["leg_wall_s"]
    assert x3["dtype"] == "f32x3" and "fp16 MFMA peak / 3" in x3["roofline"]["peak_note"] and x3["fp16_overflows"] == 0
    assert mn["dtype"] == "f32" and "MainNetwork" in mn["workload"] and mn["roofline"]["peak"] == line["roofline"]["peak"]
    assert g4["dtype"] == "f32" and g4["groups_per_gpu"] == 4 and g4["roofline_exclusive"] is False and line["roofline"]["exclusive"] is True
    assert line["extra"]["fp16_overflows"] == 0
    tp = line["extra"]["tree_pool"]                                          # one shared tree pool per context: size, fill, never empty
    assert tp["slots"] > 0 and 0 < tp["high_water_slots"] <= tp["slots"] and tp["ran_empty"] == 0
    assert x3["tree_pool"]["ran_empty"] == 0 and mn["tree_pool"]["ran_empty"] == 0


def test_bench_refuses_rccl_ranks_that_share_a_gpu():
    """Two RCCL ranks need two GPUs: on a one-GPU box `--gpus 2` over nccl ends non-zero without a line (rank 1 has no device; had
    it been given rank 0's, the distinct-GPU check refuses the job), never a 2-GPU line measured on one card."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    r = _run("--gpus 2 --games 8 --steps 1 --warmup 0 --sims 8 --stagger 0 --no-cpu-baseline --launch-timeout 120".split(), 300,
             TRANSGO_PG_TIMEOUT="60")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


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
