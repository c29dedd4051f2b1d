#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the batched self-play hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 9x9 Go, 400 simulations/move, 6-block x 128-filter tower (random-init weights),
4096 concurrent boards per GPU, self-play from empty boards with Dirichlet root noise; game seeds 1000*rank + g.
A step = one move of every board: root noise, ~100 search waves (tree kernels + network forward on each leaf batch),
visit counts -> pi and sampled move on the host, re-rooting, and the gather of finished games to rank 0.
Boards live in HBM throughout; nothing is staged from the host inside the timed region except G actions per step.
`value` = completed simulations (root visit increments) of all ranks / max-over-ranks wall time.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense fp32 matrix peak
PEAK_F16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense F16/BF16 MFMA (v_mfma_f32_16x16x32_f16), ~2.5 PFLOP/s


def flops_per_leaf(S, C, F, N):
    P = S * S
    return 2 * 9 * P * (C * F + 2 * N * F * F) + 2 * 9 * P * F * 6 + 2 * 4 * P * (P + 1) + 2 * 2 * P * 64 + 2 * 64 * (1 + P)


# ---- CPU baseline leg (the ONLY part of this file that touches oracle/) -------------------------------------------------------
def _cpu_worker(args):
    seed, budget_s, sims, F, N = args
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
    s = OracleSearch(OracleGoEnv(), ev, rng, num_simulation=sims)
    t0 = time.time()
    moves = 0
    while time.time() - t0 < budget_s:
        a, _, _, _ = s.search_move()
        moves += 1
        if s.advance(a):
            s.reset_root()
    return s.sims_done, time.time() - t0, moves


def cpu_baseline(budget_s=12.0, sims=400, F=128, N=6):
    import multiprocessing as mp
    procs = max(1, min(os.cpu_count() or 1, 16))
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(9000 + i, budget_s, sims, F, N) for i in range(procs)])
    total = sum(r[0] / r[1] for r in res)
    return {"value": round(total, 1), "unit": "sims/s", "cores": procs, "kind": "port",
            "sample": f"{procs} processes x {budget_s:.0f}s of oracle WP_MCTS self-play (oracle/wp_mcts.py + go_oracle.c + torch "
                      f"CPU 1 thread each), 9x9, {sims} sims/move, {N}x{F} tower, leaf batch 4; {sum(r[2] for r in res)} moves"}


def tree_roofline(S, C, sims, evals, depth_sum, children_scored, tree_ms, waves):
    """Second regime of SURVEY.md 8(d): selection / leaf step / expansion / backup are scan-and-graph work priced in HBM bytes.
    Algorithmic bytes per simulation with the mean depth d and mean fan-out a measured in this run (rank 0's games):
    selection d*(a+1)*32 (one 32-B record per child scored + the parent) + path updates 2*(d+1)*16*2 (pending then backup, read
    and write) + board state 2*state + legal mask 2*ceil(A/8) + feature planes written C*P*4 + policy/value read (A+1)*4 +
    expansion a*32; time = HIP events around every k_collect and k_absorb launch."""
    if sims <= 0 or tree_ms <= 0:
        return None
    P, A = S * S, S * S + 1
    d = depth_sum / sims
    a = children_scored / max(1.0, depth_sum)
    state = 48 if S == 9 else 112
    per_eval = C * P * 4 + (A + 1) * 4 + a * 32 + 2 * ((A + 7) // 8)
    bytes_per_sim = d * (a + 1) * 32 + 2 * (d + 1) * 16 * 2 + 2 * state + per_eval * (evals / sims)
    gbs = sims * bytes_per_sim / (tree_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 5),
            "bytes_per_sim": round(bytes_per_sim, 1), "mean_depth": round(d, 3), "mean_fanout": round(a, 2),
            "tree_ms_per_wave": round(tree_ms / max(1, waves), 4), "share_of_step": None,
            "kernels": "k_collect + k_absorb (one 64-lane workgroup per game; latency/occupancy-bound, see DESIGN.md)"}


def kernel_name(S, filters, dtype):
    """The F->F 3x3 conv kernel net.hip dispatches for this configuration (transgo_amd/csrc/net.hip: forward_t)."""
    if dtype == "f16":
        return f"k_conv3x3_h2<{S},{filters}> (fp16 operands, v_mfma_f32_16x16x32_f16, f32 accumulate, LDS-DMA fed)"
    if filters in (128, 256) and os.environ.get("TG_DMA_CONV", "1") != "0":
        return f"k_conv3x3_sg<{S},{filters}> (fp32 MFMA 16x16x4 implicit GEMM; weights by LDS-DMA, activations straight from L2)"
    return f"k_conv3x3<{S},{filters},{filters}> (fp32 MFMA 16x16x4 implicit GEMM)"


def main():
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
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f32",
                    help="network arithmetic: f32 (BASELINE metric, parity 1e-3) or f16 storage + f32 accumulate (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.board == 9:
        cpu = cpu_baseline(a.cpu_seconds, a.sims, a.filters, a.blocks)      # before any GPU context exists

    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local if os.environ.get("TRANSGO_DIST_BACKEND", "nccl") == "nccl" else 0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("TRANSGO_DIST_BACKEND", "nccl")        # "gloo": rehearsal of the N>1 path with several ranks on one GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")          # where collective payloads live

    from transgo_amd import model
    from transgo_amd.configure import Config
    from transgo_amd.distributed import gather_records
    from transgo_amd.self_play import BatchedSelfPlay

    S = a.board
    cfg = Config(num_simulation=a.sims, num_features=a.filters, num_blocks=a.blocks, board_size=S,
                 max_step=a.max_step or (120 if S == 9 else 450), inference_dtype=a.dtype)
    sp = BatchedSelfPlay(cfg, a.games, device=local if backend == "nccl" else 0, rank=rank, world=world)
    sp.set_weights(model.random_weights(S, 10, a.filters, a.blocks, seed=1234))
    sp.start()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = 0
    for _ in range(a.warmup):
        fin = sp.step()
        gathered += len(gather_records(fin, S, 10, 0, cdev))
    eng = sp.engine
    eng.ctx.call("tg_prof_enable", 1, 8192)
    eng.ctx.call("tg_prof_enable_tree", 1, 8192)
    cs0 = ctypes.c_uint64()
    eng.ctx.call("tg_prof_read_tree", None, None, None, ctypes.byref(cs0))
    st0 = eng.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fin = sp.step()
        gathered += len(gather_records(fin, S, 10, 0, cdev))
    barrier()
    dt = time.perf_counter() - t0
    st1 = eng.stats()
    ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    eng.ctx.call("tg_prof_read", ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl))
    cms, ams, nw, cs1 = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_uint64()
    eng.ctx.call("tg_prof_read_tree", ctypes.byref(cms), ctypes.byref(ams), ctypes.byref(nw), ctypes.byref(cs1))

    sims = st1["sims"] - st0["sims"]; evals = st1["evals"] - st0["evals"]; depth = st1["depth_sum"] - st0["depth_sum"]
    t = torch.tensor([dt], dtype=torch.float64, device=cdev)
    s = torch.tensor([float(sims), float(evals), float(depth)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dt = float(t.item()); sims_all, evals_all, depth_all = [float(x) for x in s.tolist()]

    if rank == 0:
        # HBM bytes per launch of the dominant kernel come from PMC passes (separate rocprofv3 runs, committed under profiles/);
        # they only describe the configuration they were collected on
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json" if a.dtype == "f32" else "r1_pmc_traffic_f16.json")
        if os.path.exists(tfile) and (S, a.filters, a.games) == (9, 128, 4096):
            with open(tfile) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch_mean")
        value = sims_all / dt
        conv_tflops = (fl.value / (ms.value * 1e-3)) / 1e12 if ms.value > 0 else 0.0
        fpl = flops_per_leaf(S, 10, a.filters, a.blocks)
        peak = PEAK_F16_MATRIX_TFLOPS if a.dtype == "f16" else PEAK_F32_MATRIX_TFLOPS
        tree = tree_roofline(S, 10, sims, evals, depth, cs1.value - cs0.value, cms.value + ams.value, nw.value)
        if tree:
            tree["share_of_step"] = round((cms.value + ams.value) / (dt * 1e3), 4)
        line = {
            "metric": "MCTS simulations/sec", "value": round(value, 1), "unit": "sims/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{S}x{S} Go self-play, {a.sims} sims/move, {a.blocks}-block x {a.filters}-filter tower, "
                                   f"{a.games} concurrent boards per GPU", "boards_per_gpu": a.games,
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective",
                       "step": "one move of every board (search + move selection + re-root + gather of finished games)"},
            "roofline": {"bound": "mfma", "achieved": round(conv_tflops, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(conv_tflops / peak, 4), "traffic": traffic,
                         "traffic_note": (f"HBM bytes per full-batch launch, FETCH_SIZE x2 + WRITE_SIZE (profiles/{os.path.basename(tfile)})"
                                          if traffic is not None else "PMC traffic was collected for the 9x9 / 128-filter / 4096-board workload only"),
                         "kernel": kernel_name(S, a.filters, a.dtype),
                         "launches": int(nl.value), "avg_launch_ms": round(ms.value / max(1, nl.value), 4)},
            "roofline_tree": tree,
            "cpu_baseline": cpu,
            "extra": {"leaves_per_s": round(evals_all / dt, 1), "mean_depth": round(depth_all / max(1.0, sims_all), 3),
                      "net_tflops_end_to_end": round(evals_all * fpl / dt / 1e12, 2),
                      "games_per_hour_est": round(world * a.games * 3600.0 / (dt / a.steps * cfg.max_step), 1),
                      "finished_games_gathered": gathered, "tree_errors": st1["errors"],
                      "arena_high_water_slots": st1["max_slots"]},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
