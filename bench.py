#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the batched self-play hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        no RANK/WORLD_SIZE in the environment (any N, 1 included): this process is a GPU-free LAUNCHER (as the reference's driver
        spawns its own workers, transgo.py:92-107): it times the CPU baseline, starts N fresh child processes (one rank per GPU,
        RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), takes rank 0's JSON line and exits non-zero if any rank fails.  At N = 1
        it then runs the SECONDARY legs -- the same workload under the opt-in split-precision network (`--dtype f32x3`) and under
        the reference's shipped MainNetwork in exact f32 (`--network transgo`) -- each in a fresh child of its own, one after the
        other, and prints ONE line: the exact-f32 tower as metric / value / dtype / config, the legs under "secondary".
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
        (ranks started by somebody else: rank 0 times the CPU baseline before it touches the GPU, the others wait in
        init_process_group)

Workload (BASELINE.json configs[1]): 9x9 Go, 400 simulations/move, 6-block x 128-filter tower (random-init weights),
4096 concurrent boards per GPU, self-play from empty boards with Dirichlet root noise; the k-th game in slot g of rank r is
seeded (k*world + r)*G + g (transgo_amd.self_play.default_seed: disjoint streams for every game of a job; BASELINE.md 4's
1000*rank + g would collide from 1000 boards per rank on and is knowingly not followed -- the line says so in config.seeds).
A step = one move of every board: root noise, ~100 search waves (tree kernels + network forward on each leaf batch),
visit counts -> pi and sampled move on the host, the move's record entry + re-rooting on the device, and for the games that
move finished: scoring + target generation on the device (tg_sp_harvest), the gather to rank 0 (RCCL, device buffers) and
the append into the device-resident replay store (tg_replay_append_dev), then the restart of their slots.
The boards are staggered before the warm-up (slot g is g mod max_step plies into its game, reached by real self-play with
16-simulation searches; the games the timed steps finish therefore played their early plies under those cheap searches), so the timed steps see the steady state of a running pipeline: ~G/max_step games finish on every
step.  Boards live in HBM throughout; per step the host receives G*(A+1) int32 visit counts and sends G actions.
`value` = completed simulations (root visit increments) of all ranks / max-over-ranks wall time; `games_per_hour` =
games finished (and stored) inside the timed region / the same wall time.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

# the hosts of this pool share GPU memory between processes through dmabuf handles only (RCCL needs it for N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense fp32 matrix peak
PEAK_F16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense F16/BF16 MFMA (v_mfma_f32_16x16x32_f16), ~2.5 PFLOP/s


def net_source_hash():
    """Identity of the network kernels' source for the PMC traffic figure: sha256 of transgo_amd/csrc/net.hip with comments and
    blank space removed (a comment edit does not make a collected figure stale, a code edit does).  None without the source."""
    import hashlib
    import re
    src = os.path.join(ROOT, "transgo_amd", "csrc", "net.hip")
    if not os.path.exists(src):
        return None
    txt = open(src, encoding="utf-8", errors="replace").read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    txt = "".join(txt.split())
    return hashlib.sha256(txt.encode()).hexdigest()[:16]


def flops_per_leaf(S, C, F, N):
    P = S * S
    return 2 * 9 * P * (C * F + 2 * N * F * F) + 2 * 9 * P * F * 6 + 2 * 4 * P * (P + 1) + 2 * 2 * P * 64 + 2 * 64 * (1 + P)


# ---- CPU baseline leg (the ONLY part of this file that touches oracle/) -------------------------------------------------------
def usable_cores():
    """Host cores this process may actually use: scheduler affinity, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_worker(args):
    seed, budget_s, sims, F, N, max_step = args
    import torch
    torch.set_num_threads(1)
    from oracle.go_oracle import OracleGoEnv
    from oracle.net import TowerNetwork
    from oracle.wp_mcts import OracleSearch
    from transgo_amd import model
    net = TowerNetwork(9, 10, F, N).eval()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in model.random_weights(9, 10, F, N, seed=1234).items()}
    net.load_state_dict(sd)

    def ev(obs):
        with torch.no_grad():
            p, v, _ = net.main_prediction(torch.from_numpy(obs))
        return p.numpy(), v.numpy()
    rng = np.random.RandomState(seed)
    s = OracleSearch(OracleGoEnv(max_step=max_step), ev, rng, num_simulation=sims)
    t_end = time.time() + budget_s
    moves = games = ply = 0
    sims_timed, t_timed = 0, 0.0
    while time.time() < t_end:
        t0, n0 = time.time(), s.sims_done
        a, _, _, _ = s.search_move()
        done = s.advance(a)
        if ply >= 10:                                    # the first 10 plies of each game are not timed
            sims_timed += s.sims_done - n0; t_timed += time.time() - t0
        moves += 1; ply += 1
        if done:
            s.reset_root(); games += 1; ply = 0
    return sims_timed, t_timed, moves, games


def cpu_baseline(budget_s=60.0, sims=400, F=128, N=6, max_step=120, label="C2"):
    """P = every usable host core, one oracle self-play game per process (leaf batch 4, torch CPU pinned
    to 1 thread), a fixed window per process, game seeds 1000*rank + g with rank = 0."""
    import multiprocessing as mp
    procs = usable_cores()
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(g, budget_s, sims, F, N, max_step) for g in range(procs)])
    total = sum(r[0] / r[1] for r in res if r[1] > 0)
    moves = sum(r[2] for r in res)
    return {"value": round(total, 1), "unit": "sims/s", "cores": procs, "os_cpu_count": os.cpu_count(), "kind": "port",
            "per_core": round(total / procs, 1), "config": label,
            "moves_per_hour": round(moves / budget_s * 3600.0, 1),
            "games_per_hour_equiv": round(moves / budget_s * 3600.0 / max_step, 2),
            "sample": f"{procs} processes x {budget_s:.0f}s of oracle WP_MCTS self-play (oracle/wp_mcts.py + go_oracle.c + torch "
                      f"CPU 1 thread each), 9x9, {sims} sims/move, {N}x{F} tower, leaf batch 4, first 10 plies of a game untimed; "
                      f"{moves} moves, {sum(r[3] for r in res)} games finished"}


def tree_roofline(S, C, sims, evals, depth_sum, children_scored, tree_ms, waves):
    """Second regime of SURVEY.md 8(d): selection / leaf step / expansion / backup are scan-and-graph work priced in HBM bytes.
    Algorithmic bytes per simulation with the mean depth d and mean fan-out a measured in this run (rank 0's games):
    selection d*(a+1)*32 (one 32-B record per child scored + the parent) + path updates 2*(d+1)*16*2 (pending then backup, read
    and write) + board state 2*state + legal mask 2*ceil(A/8) + feature planes written bit-packed ceil(C*P/32)*4 (the evaluation
    batch; round 1 wrote float planes, C*P*4) + policy/value read (A+1)*4 + expansion a*32; time = HIP events around every
    k_collect and k_absorb launch."""
    if sims <= 0 or tree_ms <= 0:
        return None
    P, A = S * S, S * S + 1
    d = depth_sum / sims
    a = children_scored / max(1.0, depth_sum)
    state = 48 if S == 9 else 112
    per_eval = ((C * P + 31) // 32) * 4 + (A + 1) * 4 + a * 32 + 2 * ((A + 7) // 8)
    bytes_per_sim = d * (a + 1) * 32 + 2 * (d + 1) * 16 * 2 + 2 * state + per_eval * (evals / sims)
    gbs = sims * bytes_per_sim / (tree_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 5),
            "bytes_per_sim": round(bytes_per_sim, 1), "mean_depth": round(d, 3), "mean_fanout": round(a, 2),
            "tree_ms_per_wave": round(tree_ms / max(1, waves), 4), "share_of_step": None,
            "kernels": "k_collect + k_absorb (one 64-lane workgroup per game; latency/occupancy-bound, see DESIGN.md)"}


def kernel_name(S, filters, dtype):
    """The F->F 3x3 conv kernel net.hip dispatches for this configuration (transgo_amd/csrc/net.hip: forward_t)."""
    if dtype == "f32x3":
        return (f"k_conv3x3_h2<{S},{2 * filters},{filters},X2> (split precision: operands as fp16 hi + lo, three partial products "
                "(hi*hi, hi*lo, lo*hi) on v_mfma_f32_16x16x32_f16, f32 accumulate, f32 residual stream; LDS-DMA fed)")
    if dtype != "f32":
        return (f"k_conv3x3_h2<{S},{filters}> (fp16 operands, v_mfma_f32_16x16x32_f16, f32 accumulate, LDS-DMA fed"
                + (", fp16 residual stream)" if dtype == "f16r" else ")"))
    if filters in (128, 256) and os.environ.get("TG_DMA_CONV", "1") != "0":
        return (f"k_conv3x3_sg<{S},{filters}> (fp32 MFMA 16x16x4 implicit GEMM; weights by LDS-DMA, activations straight from L2"
                + ("; 192- or 128-row tiles chosen per launch)" if filters == 128 else ")"))
    return f"k_conv3x3<{S},{filters},{filters}> (fp32 MFMA 16x16x4 implicit GEMM)"


def stagger(sp, period, sims=16):
    """Put slot g (g mod period) plies into its game -- slots with g mod period = 0 are never restarted and end up period - 1
    plies in -- by real self-play with cheap searches: `period - 1` untimed moves of every board, slot g restarted (fresh seed,
    empty board, empty record) just before the move that leaves it at its offset.  Every record entry the timed steps later harvest was written by the engine's own tg_sp_play."""
    T0 = period - 1
    offs = np.arange(sp.G) % period
    for s_ in range(T0):
        m = offs == (T0 - s_)
        if s_ > 0 and m.any():
            sp._reset(m)
        sp.advance(num_simulation=sims)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent boards per GPU")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--board", type=int, default=9, help="board edge (9 = BASELINE configs[1]; 19 = configs[3])")
    ap.add_argument("--max-step", type=int, default=0, help="ply limit (default 120 at 9x9, 450 at 19x19)")
    ap.add_argument("--dtype", choices=["f32", "f16", "f16r", "f32x3"], default="f32",
                    help="network arithmetic: f32 (BASELINE metric, exact-fp32 MFMA), f16 / f16r = fp16 storage + f32 accumulate "
                         "(configs[4]; f16r: fp16 residual stream too), f32x3 = opt-in split-precision convs (operands as fp16 hi + lo, "
                         "three of their four partial products on the fp16 MFMA, f32 accumulate; ~1e-7 of f32)")
    ap.add_argument("--network", choices=["tower", "transgo"], default="tower",
                    help="tower = BASELINE.json's N-block x F-filter net; transgo = the reference's shipped MainNetwork (model.py:41-114)")
    ap.add_argument("--stagger", type=int, default=-1,
                    help="spread the boards over this many ply offsets before the warm-up (default: max_step at 9x9, 0 = all "
                         "boards start together, at 19x19)")
    ap.add_argument("--groups", type=int, default=1,
                    help="run the boards of a GPU as this many independent groups (own context + HIP stream each, advanced by as many "
                         "host threads): the groups' kernels overlap on the GPU.  Default 1: one serial chain, the configuration the "
                         "roofline figures are defined on (with K > 1 launches of different groups share the chip, so per-launch "
                         "durations are no longer exclusive and `roofline` says so)")
    ap.add_argument("--arena-slots", type=int, default=0,
                    help="most 32-byte tree slots ONE game's tree may hold (per-game cap; default: (4*sims + 256) * (header + actions)); "
                         "memory is set by --pool-slots")
    ap.add_argument("--pool-slots", type=int, default=0,
                    help="tree memory: 32-byte slots provisioned per game ON AVERAGE in the pool all boards of a GPU share (default: "
                         "(2*sims + 128) * (header + actions)); the line reports the pool's size, its high-water mark, how often it "
                         "ran empty and any truncated tree blocks, so a run can be sized for more boards per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=0.0,
                    help="window of each CPU-baseline leg (default: 60 s at N = 1 as BASELINE.md 4 says, 30 s at N > 1)")
    ap.add_argument("--cpu-json", default="", help="(set by the launcher) file holding the CPU-baseline legs the launcher timed")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="launcher: seconds before the ranks are given up on")
    ap.add_argument("--no-launcher", action="store_true",
                    help="N = 1 only: be rank 0 of a world of one in THIS process (no children, no secondary legs) -- what a profiler "
                         "needs (`rocprofv3 ... -- python3 bench.py --no-launcher --no-cpu-baseline`): its preloaded library has "
                         "initialised the GPU before Python starts, and such a process must not start GPU children")
    ap.add_argument("--secondary", choices=["auto", "none"], default="auto",
                    help="auto: after the headline leg of the default family (N = 1, tower, f32, one group) the launcher times the "
                         "f32x3 tower and the f32 MainNetwork on the same workload in fresh child processes and reports them under "
                         "\"secondary\" (never as `value`)")
    ap.add_argument("--secondary-timeout", type=float, default=240.0, help="launcher: seconds one secondary leg may take")
    return ap.parse_args(argv)


def cpu_legs(a):
    """The two CPU-baseline legs (C2's net for the like-for-like ratio, C1 = the plumbing config beside it), run where no GPU
    context exists: the launcher, or rank 0 before it touches the GPU."""
    if a.no_cpu_baseline or a.board != 9 or a.network != "tower":
        return None, None
    secs = a.cpu_seconds if a.cpu_seconds > 0 else (60.0 if a.gpus <= 1 else 30.0)
    c2 = cpu_baseline(secs, a.sims, a.filters, a.blocks,
                      label="C2 net" if (a.sims, a.filters, a.blocks) == (400, 128, 6) else "bench net")
    c1 = cpu_baseline(secs, 64, 32, 2, label="C1")
    return c2, c1


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawn_ranks(argv, n, tmp, tag, timeout_s):
    """Start n fresh children of this file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), wait for them, end every rank (exact
    PIDs) as soon as one fails or the timeout passes.  Returns (rc, rank 0's stdout, why)."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port),
                   TRANSGO_BENCH_LAUNCHER="bench.py")
        out = open(os.path.join(tmp, f"{tag}_rank{r}.out"), "w+")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out, stderr=None))
    deadline = time.time() + timeout_s
    rc, why = 0, ""
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = bad[0][1] if bad[0][1] > 0 else 1
                why = f"rank {bad[0][0]} exited with {bad[0][1]}; stopping the other ranks"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                rc, why = 124, f"ranks still running after {timeout_s:.0f} s; giving up"
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(20)
            except subprocess.TimeoutExpired:
                p.kill()
    txt0 = ""
    for r, out in enumerate(outs):
        out.seek(0)
        txt = out.read()
        out.close()
        if r == 0:
            txt0 = txt
        elif txt.strip():
            sys.stderr.write(txt)
    return rc, txt0, why


def _result_line(txt):
    for l in reversed(txt.splitlines()):
        if l.startswith("{"):
            try:
                return json.loads(l)
            except ValueError:
                pass
    return None


# What the launcher times after the headline leg at N = 1: the round's opt-in modes where the driver's clock
# sees them.  (name, arguments replacing the headline's, cap on steps, cap on warm-up)
SECONDARY_LEGS = (
    ("f32x3", ["--dtype", "f32x3"], 10, 3),
    ("mainnetwork_f32", ["--network", "transgo", "--dtype", "f32"], 3, 1),
    # the headline's own arithmetic with the boards as four independent groups on four HIP streams (same games bit for bit): the
    # best exact-f32 rate of the build; not the headline because per-launch durations stop being exclusive (roofline.exclusive false)
    ("f32_groups4", ["--groups", "4"], 8, 2),
)


def _secondary_summary(line, wall_s):
    rf, ex = line.get("roofline") or {}, line.get("extra") or {}
    return {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"],
            "warmup": line["warmup"], "dtype": line["dtype"], "games_per_hour": line.get("games_per_hour"),
            "workload": (line.get("config") or {}).get("workload"),
            "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "peak_note", "kernel", "launches",
                                                "launches_not_timed", "avg_launch_ms")},
            "net_tflops_end_to_end": ex.get("net_tflops_end_to_end"), "leaves_per_s": ex.get("leaves_per_s"),
            "tree_errors": ex.get("tree_errors"), "fp16_overflows": ex.get("fp16_overflows"),
            "truncated_tree_blocks": ex.get("truncated_tree_blocks"), "arena_high_water_slots": ex.get("arena_high_water_slots"),
            "tree_pool": ex.get("tree_pool"), "groups_per_gpu": (line.get("config") or {}).get("groups_per_gpu"),
            "roofline_exclusive": rf.get("exclusive"),
            "leg_wall_s": round(wall_s, 1)}


def launcher(a, argv):
    """`python bench.py --gpus N` with no rank environment: the GPU-free parent (nothing here imports torch or touches HIP).
    CPU legs first, then N fresh children -- this same file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- whose rank 0
    writes the JSON line with the CPU legs merged in.  Any failing rank, or the join timeout, ends the others (exact PIDs) and
    the launcher exits non-zero.  At N = 1 the secondary legs follow, each in its own fresh child (never an exec, never a second
    network inside a process that already holds one): a leg that fails leaves {"error": ...} under its name and does not take
    the headline with it."""
    import tempfile
    cpu, cpu_c1 = cpu_legs(a)
    tmp = tempfile.mkdtemp(prefix="transgo_bench_")
    cpu_json = os.path.join(tmp, "cpu.json")
    with open(cpu_json, "w") as f:
        json.dump({"cpu_baseline": cpu, "cpu_baseline_c1": cpu_c1}, f)
    args = [x for x in argv] + ["--no-cpu-baseline", "--cpu-json", cpu_json]
    t0 = time.time()
    rc, txt, why = _spawn_ranks(args, a.gpus, tmp, "main", a.launch_timeout)
    line = _result_line(txt)
    if rc != 0:
        print(f"bench.py launcher: {why}", file=sys.stderr)
        sys.stdout.write("".join(l + "\n" for l in txt.splitlines() if not l.startswith("{")))
        return rc
    if line is None:
        print("bench.py launcher: rank 0 printed no result line", file=sys.stderr)
        return 1
    line["launcher_wall_s"] = {"cpu_legs_and_headline": round(time.time() - t0, 1)}
    if a.gpus == 1 and a.secondary == "auto" and (a.dtype, a.network, a.groups) == ("f32", "tower", 1):
        sec = {}
        for name, extra, max_steps, max_warm in SECONDARY_LEGS:
            leg = [x for x in argv] + extra + ["--steps", str(min(a.steps, max_steps)), "--warmup", str(min(a.warmup, max_warm)),
                                               "--no-cpu-baseline"]
            t1 = time.time()
            rc2, txt2, why2 = _spawn_ranks(leg, 1, tmp, name, a.secondary_timeout)
            l2 = _result_line(txt2)
            if rc2 == 0 and l2 is not None:
                sec[name] = _secondary_summary(l2, time.time() - t1)
            else:
                sec[name] = {"error": why2 or "no result line", "rc": rc2, "leg_wall_s": round(time.time() - t1, 1)}
                print(f"bench.py launcher: secondary leg {name} failed ({sec[name]['error']})", file=sys.stderr)
        line["secondary"] = dict(sec, note="same workload, same launcher, fresh process each, timed after the headline leg; "
                                           "opt-in modes reported beside the exact-f32 tower, never as `value`")
        line["launcher_wall_s"]["secondary"] = round(sum(v.get("leg_wall_s", 0) for v in sec.values()), 1)
    print(json.dumps(line), flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    have_ranks = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not have_ranks and not (a.no_launcher and a.gpus == 1):
        raise SystemExit(launcher(a, argv))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")

    cpu = cpu_c1 = None
    if a.cpu_json:
        if rank == 0:
            with open(a.cpu_json) as f:
                legs = json.load(f)
            cpu, cpu_c1 = legs.get("cpu_baseline"), legs.get("cpu_baseline_c1")
    elif rank == 0:
        # before any GPU context exists in this process; with ranks started by somebody else (torchrun) the other ranks wait for
        # rank 0 inside init_process_group, idle
        cpu, cpu_c1 = cpu_legs(a)

    import datetime
    import torch
    import torch.distributed as dist
    backend = os.environ.get("TRANSGO_DIST_BACKEND", "nccl")        # "gloo": rehearsal of the N>1 path with several ranks on one GPU
    # TRANSGO_BENCH_STANDIN=1 (with gloo): a CPU rehearsal of the N-rank control flow -- launcher, process group, per-move exchange,
    # reductions, the line -- with tests/bench_standin.py in place of the HIP engine: no GPU is touched and nothing it prints is a
    # measurement (the line says so in `data`).  How BASELINE configs[2]/[4]'s world of 8 is rehearsed where 8 GPU processes cannot run.
    standin = os.environ.get("TRANSGO_BENCH_STANDIN", "0") == "1"
    if standin and backend == "nccl":
        raise SystemExit("TRANSGO_BENCH_STANDIN=1 is a CPU rehearsal: set TRANSGO_DIST_BACKEND=gloo")
    dev = torch.device("cuda", local if backend == "nccl" else 0)
    if not standin:
        torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("TRANSGO_PG_TIMEOUT", "1800")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
    cdev = dev if backend == "nccl" else torch.device("cpu")          # where collective payloads live
    # what the process group really is: answered from the group, not from the arguments
    from types import SimpleNamespace
    props = SimpleNamespace(name="stand-in (no GPU)") if standin else torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": int(dev.index), "name": props.name, "pid": os.getpid(),
          "bus": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks = {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": seen,
                 "distinct_gpus": len({(d["device"], d["bus"], d["uuid"]) for d in seen}),
                 "launcher": os.environ.get("TRANSGO_BENCH_LAUNCHER", "external (torch.distributed.run)")}
        if ranks["backend"] == "nccl" and ranks["distinct_gpus"] < ranks["world"]:
            # every rank sees the same list and leaves together: a job whose ranks share a card is not the job --gpus names
            raise SystemExit(f"bench.py: {ranks['world']} RCCL ranks on {ranks['distinct_gpus']} distinct GPU(s) "
                             f"({[(d['rank'], d['device'], d['bus']) for d in seen]}): refusing to report it as --gpus {a.gpus}")
    else:
        ranks = {"world": 1, "backend": None, "devices": [me], "distinct_gpus": 1,
                 "launcher": os.environ.get("TRANSGO_BENCH_LAUNCHER", "none")}

    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.distributed import gather_harvest, transport_name
    from transgo_amd.replay_buffer import DeviceReplayMemory
    from transgo_amd.self_play import BatchedSelfPlay, GroupedSelfPlay

    S = a.board
    cfg = Config(num_simulation=a.sims, num_features=a.filters, num_blocks=a.blocks, board_size=S,
                 max_step=a.max_step or (120 if S == 9 else 450), inference_dtype=a.dtype, network=a.network)
    gpu = local if backend == "nccl" else 0
    if a.network == "transgo" and not a.arena_slots:
        # the shipped MainNetwork's policies are peaked even at random init: at the default arena 0.5 % of the game-moves lose kept
        # sub-tree blocks at re-rooting (counted in truncated_tree_blocks; DESIGN.md 3 "Sizing") -- twice the default holds them all
        a.arena_slots = 2 * (4 * a.sims + 256) * ((2 if S == 9 else 4) + S * S + 1)
    if standin:
        from tests.bench_standin import StandInReplay, StandInSelfPlay
        sp = StandInSelfPlay(cfg, a.games, rank=rank, world=world)
        parts = [sp]
    elif a.groups > 1:
        sp = GroupedSelfPlay(cfg, a.games, groups=a.groups, device=gpu, rank=rank, world=world, arena_slots=a.arena_slots,
                             pool_slots=a.pool_slots)
        parts = sp.parts
    else:
        sp = BatchedSelfPlay(cfg, a.games, device=gpu, rank=rank, world=world, arena_slots=a.arena_slots, pool_slots=a.pool_slots)
        parts = [sp]
    if a.network == "transgo":
        sp.set_weights(model.random_transgo_weights(S, 10, a.filters, seed=1234))
    else:
        sp.set_weights(model.random_weights(S, 10, a.filters, a.blocks, seed=1234))
    # rank 0 owns the replay store (north_star: "RCCL gather of (s, pi, z) tuples into the replay buffer"); sized for every
    # position the run can produce
    mem = None
    if rank == 0:
        mem = StandInReplay() if standin else DeviceReplayMemory(cfg, capacity_positions=max(1024, world * a.games * (a.steps + a.warmup + 2)), device=gpu)
    sp.start()
    period = 0 if standin else (a.stagger if a.stagger >= 0 else (cfg.max_step if S == 9 else 0))
    t_st = time.perf_counter()
    if period > 1:
        if a.groups > 1:
            sp._each(lambda part: stagger(part, period))
        else:
            stagger(sp, period)
    stagger_s = time.perf_counter() - t_st

    def barrier():
        if world > 1:
            dist.barrier()
        if not standin:
            torch.cuda.synchronize()

    tally = {"games": 0, "positions": 0}

    def one_step():
        hs = sp.advance(device=True)
        for h in (hs if a.groups > 1 else [hs]):                          # one gather per group, the same number on every rank
            for hb in gather_harvest(h, S, 10, dst=0, device_index=gpu):  # rank 0: every rank's finished games
                mem.append_harvest(hb)
                tally["games"] += hb.n_games; tally["positions"] += hb.n_positions

    for _ in range(a.warmup):
        one_step()
    tally["games"] = tally["positions"] = 0
    engs = [part.engine for part in parts]
    # event pools sized for ONE step; they are drained into running totals after every step (the stream is idle there: the step
    # ended with the visit-count read-back), so the roofline covers every conv launch of the timed region
    def all_stats():
        tot = {}
        for e in engs:
            for k, v in e.stats().items():
                tot[k] = max(tot.get(k, 0), v) if k == "max_slots" else tot.get(k, 0) + v
        return tot

    def children_scored():
        n = 0
        for e in engs:
            c = ctypes.c_uint64()
            e.ctx.call("tg_prof_read_tree", None, None, None, ctypes.byref(c))
            n += c.value
        return n
    for e in engs:
        e.ctx.call("tg_prof_enable", 1, 8192)
        e.ctx.call("tg_prof_enable_tree", 1, 8192)
    cs0 = children_scored()
    st0 = all_stats()
    fin0, drop0 = sp.games_finished, sp.games_dropped
    for part in parts:
        part.phase_s = {}
    for e in engs:
        e.begin_move_s = 0.0
    prof = [dict(ms=ctypes.c_double(), nl=ctypes.c_int64(), fl=ctypes.c_double()) for _ in engs]
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
        for e, pr in zip(engs, prof):
            e.ctx.call("tg_prof_read", ctypes.byref(pr["ms"]), ctypes.byref(pr["nl"]), ctypes.byref(pr["fl"]))
            e.ctx.call("tg_prof_read_tree", None, None, None, None)
    barrier()
    dt = time.perf_counter() - t0
    st1 = all_stats()
    ms_v = nl_v = fl_v = nskip_v = cms_v = ams_v = nw_v = 0
    for e, pr in zip(engs, prof):
        e.ctx.call("tg_prof_read", ctypes.byref(pr["ms"]), ctypes.byref(pr["nl"]), ctypes.byref(pr["fl"]))
        k = ctypes.c_int64(); e.ctx.call("tg_prof_skipped", ctypes.byref(k))
        c1, a1, w1 = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        e.ctx.call("tg_prof_read_tree", ctypes.byref(c1), ctypes.byref(a1), ctypes.byref(w1), None)
        ms_v += pr["ms"].value; nl_v += pr["nl"].value; fl_v += pr["fl"].value; nskip_v += k.value
        cms_v += c1.value; ams_v += a1.value; nw_v += w1.value
    cs1 = children_scored()
    from types import SimpleNamespace as _V          # summed over the groups; the line below reads `.value`
    ms, nl, fl, nskip, cms, ams, nw, cs0, cs1 = (_V(value=v) for v in (ms_v, nl_v, fl_v, nskip_v, cms_v, ams_v, nw_v, cs0, cs1))
    eng = engs[0]
    phase_s = {}
    for part in parts:
        for k, v in part.phase_s.items():
            phase_s[k] = phase_s.get(k, 0.0) + v / len(parts)
    begin_move_s = sum(e.begin_move_s for e in engs) / len(engs)

    sims = st1["sims"] - st0["sims"]; evals = st1["evals"] - st0["evals"]; depth = st1["depth_sum"] - st0["depth_sum"]
    t = torch.tensor([dt], dtype=torch.float64, device=cdev)
    s = torch.tensor([float(sims), float(evals), float(depth), float(sp.games_finished - fin0), float(sp.games_dropped - drop0)],
                     dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    own_dt = dt
    dt = float(t.item()); sims_all, evals_all, depth_all, fin_all, drop_all = [float(x) for x in s.tolist()]
    # every rank's own clock and count next to the job figure (value = sum of sims / slowest rank's time hides a straggler)
    mine = {"rank": rank, "device": int(dev.index), "sims": int(sims), "seconds": round(own_dt, 4),
            "ms_per_step": round(own_dt / a.steps * 1e3, 2), "sims_per_s": round(sims / own_dt, 1),
            "games_finished": int(sp.games_finished - fin0)}
    seeds_now = np.concatenate([np.asarray(part.seeds, np.int64) for part in parts])      # the games running in this rank's slots
    mine["seed_min"], mine["seed_max"] = int(seeds_now.min()), int(seeds_now.max())
    per_rank, all_seeds = [mine], [seeds_now]
    if world > 1:
        per_rank, all_seeds = [None] * world, [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.all_gather_object(all_seeds, seeds_now)
    ranks["per_rank"] = per_rank
    cat = np.concatenate(all_seeds)
    ranks["seeds_disjoint"] = bool(len(np.unique(cat)) == len(cat))       # no two running games of the job share an RNG stream
    ranks["transport"] = transport_name() if world > 1 else "none (one rank: finished games go device -> device into the store)"

    if rank == 0:
        # HBM bytes per launch of the dominant kernel come from PMC passes (separate rocprofv3 runs, committed under profiles/);
        # they describe one configuration AND one build: the file records the hash of net.hip it was collected on, and a figure
        # from another build is not reported (traffic = null, the note says why) instead of going stale silently
        traffic, tnote = None, "PMC traffic was collected for the 9x9 / 128-filter / 4096-board tower workload only"
        tname = {"f32": "pmc_traffic.json", "f32x3": "pmc_f32x3.json"}.get(a.dtype)
        tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if tname and f.endswith("_" + tname)) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if tfiles and (S, a.filters, a.games, a.network) == (9, 128, 4096, "tower"):
            tfile = os.path.join(ROOT, "profiles", tfiles[-1])                # the latest round's
            with open(tfile) as f:
                pj = json.load(f)
            now = net_source_hash()
            if pj.get("net_hip_sha16") and pj["net_hip_sha16"] == now:
                traffic = pj.get("hbm_bytes_per_launch_mean")
                tnote = f"HBM bytes per full-batch launch, FETCH_SIZE x2 + WRITE_SIZE (profiles/{tfiles[-1]}, collected on this build of net.hip)"
            else:
                tnote = (f"profiles/{tfiles[-1]} was collected on another build of net.hip ({pj.get('net_hip_sha16')} vs {now}): not reported; "
                         "re-collect with scripts/collect_profiles.sh")
        value = sims_all / dt
        conv_tflops = (fl.value / (ms.value * 1e-3)) / 1e12 if ms.value > 0 else 0.0
        fpl = flops_per_leaf(S, 10, a.filters, a.blocks) if a.network == "tower" else None
        peak = {"f32": PEAK_F32_MATRIX_TFLOPS, "f32x3": PEAK_F16_MATRIX_TFLOPS / 3.0}.get(a.dtype, PEAK_F16_MATRIX_TFLOPS)
        tree = tree_roofline(S, 10, sims, evals, depth, cs1.value - cs0.value, cms.value + ams.value, nw.value)
        if tree:
            tree["share_of_step"] = round((cms.value + ams.value) / (dt * 1e3), 4)
        games_stored = tally["games"]
        mean_len = tally["positions"] / games_stored if games_stored else None
        info = mem.info()
        net_name = (f"{a.blocks}-block x {a.filters}-filter tower" if a.network == "tower"
                    else f"reference MainNetwork (RARRRARRRRAR+P, {a.filters} filters)")
        gph_step = round(world * a.games * 3600.0 / (dt / a.steps * mean_len), 1) if mean_len else None
        line = {
            "metric": "REHEARSAL with a stand-in engine (not MCTS simulations/sec)" if standin else "MCTS simulations/sec",
            "value": round(value, 1), "unit": "stand-in steps/s" if standin else "sims/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
            "data": "stand-in engine (CPU rehearsal of the N-rank control flow: NOT a measurement)" if standin else "synthetic",
            "games_per_hour": round(games_stored / dt * 3600.0, 1),
            "ranks": ranks,
            "config": {"workload": f"{S}x{S} Go self-play, {a.sims} sims/move, {net_name}, "
                                   f"{a.games} concurrent boards per GPU", "boards_per_gpu": a.games,
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective" + (
                           f"; on each GPU {a.groups} independent groups of {a.games // a.groups} boards on their own HIP streams" if a.groups > 1 else ""),
                       "groups_per_gpu": a.groups,
                       "seeds": "k-th game of slot g on rank r: (k*world + r)*G + g (disjoint per job; not BASELINE.md 4's "
                                "1000*rank + g, which collides beyond 1000 boards per rank)",
                       "step": "one move of every board (search + move selection + record + re-root) and, for the games it "
                               "finishes, device-side target generation, gather to rank 0 and append to the device replay store",
                       "stagger": (f"slot g starts the warm-up g mod {period} plies into its game (slots with g mod {period} = 0: "
                                   f"{period - 1} plies); {period - 1} untimed moves with 16-simulation searches, {stagger_s:.1f} s") if period > 1 else "none: all boards start together"},
            "roofline": {"bound": "mfma", "achieved": round(conv_tflops, 2), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(conv_tflops / peak, 4), "traffic": traffic,
                         "traffic_note": tnote,
                         "peak_note": ("dense fp16 MFMA peak / 3: every F->F conv runs three partial products (w_hi a_hi, w_hi a_lo, w_lo a_hi; "
                                       "w_lo a_lo, 2^-22 of a product, is dropped) on the fp16 MFMA; achieved counts the conv's 2*9*F*F FLOP "
                                       "per row once") if a.dtype == "f32x3" else None,
                         "kernel": kernel_name(S, a.filters, a.dtype),
                         "launches": int(nl.value), "launches_not_timed": int(nskip.value),
                         "avg_launch_ms": round(ms.value / max(1, nl.value), 4),
                         "rank": 0,
                         "exclusive": a.groups == 1,
                         "exclusive_note": None if a.groups == 1 else
                         (f"{a.groups} groups launch on {a.groups} streams: conv launches of different groups share the chip, so avg_launch_ms "
                          "is not an exclusive duration and `achieved` understates the kernel; the groups = 1 line carries the roofline")},
            "roofline_tree": tree,
            "cpu_baseline": cpu,
            "cpu_baseline_c1": cpu_c1,
            "selfplay_games": {"finished_and_stored": games_stored, "finished_all_ranks": int(fin_all),
                               "positions_stored": tally["positions"], "mean_game_length": round(mean_len, 2) if mean_len else None,
                               "games_per_hour": round(games_stored / dt * 3600.0, 1),
                               "games_per_hour_from_step_time": gph_step,
                               "note": ("the games counted here were started before the timed region and played their first plies "
                                        "under the stagger's 16-simulation searches (only the plies inside the timed steps are 400-"
                                        "simulation searches); games_per_hour_from_step_time = boards x 3600 / (step time x mean "
                                        "length) is the steady-state rate of full-strength games") if period > 1 else None,
                               "dropped_arena_overflow": int(drop_all),
                               "replay_entries": info["entries"], "consumer": "DeviceReplayMemory on rank 0 (tg_replay_append_dev)"},
            "extra": {"leaves_per_s": round(evals_all / dt, 1), "mean_depth": round(depth_all / max(1.0, sims_all), 3),
                      "net_tflops_end_to_end": round(evals_all * fpl / dt / 1e12, 2) if fpl else None,
                      "tree_errors": st1["errors"], "fp16_overflows": st1["fp16_overflows"],
                      "arena_high_water_slots": st1["max_slots"],
                      "arena_cap_slots_per_game": int(eng.ctx.cfg.arena_slots) or (4 * a.sims + 256) * ((2 if S == 9 else 4) + S * S + 1),
                      "tree_pool": {"slots": st1["pool_slots"], "gib": round(st1["pool_slots"] * 32 / 2 ** 30, 2),
                                    "high_water_slots": st1["pool_high_water"], "high_water_frac": round(st1["pool_high_water"] / max(1, st1["pool_slots"]), 4),
                                    "ran_empty": st1["pool_exhausted"],
                                    "note": "one pool of 32-byte slots per engine context, shared by its boards (chunks of 1024 slots at 9x9, "
                                            "4096 at 19x19); arena_high_water_slots = the largest single tree"},
                      "truncated_tree_blocks": st1["truncated_blocks"],
                      "step_phases_ms": dict({k: round(v / a.steps * 1e3, 2) for k, v in phase_s.items()},
                                             begin_move_inside_search=round(begin_move_s / a.steps * 1e3, 2)),
                      "step_phases_note": "rank 0 wall clock per step: search = root noise (host Dirichlet) + all waves; select = visit "
                                          "counts D2H + pi/move sampling on the host; play = record + re-root kernel + new-root evaluation; "
                                          "game_end = harvest + restart of finished slots (gather/append are outside these four)"},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
