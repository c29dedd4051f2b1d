"""RCCL on real hardware.  A second RCCL rank needs a second GPU, so on the one-GPU box the RCCL code paths run as a process group
of ONE rank (TRANSGO_DIST_SINGLE_RANK=1): torch.distributed's nccl backend (= RCCL) initialised with the device bound next to the
library's own HIP streams, the actor loop's control word / size exchange / weight broadcast on DEVICE tensors, and the device ->
device weight load (tg_net_load_async_dev) from the buffer the broadcast filled.  The 2-rank send/recv of payloads is covered over
gloo (test_gpu_records.py, test_host_logic.py) and by test_actor_loop_two_ranks_rccl where two GPUs exist."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(port, q):
    import ctypes
    os.environ["TRANSGO_DIST_SINGLE_RANK"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from transgo_amd import distributed, model
    from transgo_amd.configure import Config
    from transgo_amd.replay_buffer import DeviceReplayMemory
    from transgo_amd.self_play import SelfPlay
    from transgo_amd.shared_storage import SharedStorage
    distributed.init_process_group("nccl", 0, 1, device_index=0, timeout_s=120.0, init_method=f"tcp://127.0.0.1:{port}")
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    # control word and size exchange on device tensors
    out["ctl"] = distributed.control_exchange([1, 2], src=0, device_index=0)
    # broadcast of a packed blob: comes back as a device tensor, complete
    sd = [model.random_weights(9, 10, 32, 2, seed=s) for s in (1, 2)]
    blob = model.pack_weights(sd[1], 9, 10, 32, 2)
    got = distributed.broadcast_weights(blob, src=0, device=torch.device("cuda", 0))
    out["bcast_device"] = bool(got.is_cuda) and bool(np.array_equal(got.cpu().numpy(), blob))
    # the actor loop over RCCL: weights v1 at the first move, v2 published later -> broadcast -> tg_net_load_async_dev
    cfg = Config(num_simulation=8, max_step=5, num_features=32, num_blocks=2, buffer_size=4096)
    actor = SelfPlay(cfg, n_games=4, device=0)
    st = SharedStorage({"weights": sd[0], "now_play_steps": 0, "now_play_games": 0, "now_train_steps": 1, "train_play_ratio": 0.075,
                        "adjust_train_play_ratio": False, "game_total_num": 1e8, "adjust_lr": False, "learn_rate": 1e-4}, cfg)
    mem = DeviceReplayMemory(cfg, capacity_positions=256, device=0)
    actor.continuous_self_play(st, mem, max_moves=3)
    st.set_info({"weights": sd[1], "now_train_steps": 2})
    actor.continuous_self_play(st, mem, max_moves=4)                 # 7 moves: one generation (5 plies) finished and stored
    pend = ctypes.c_int(-1)
    actor.worker.engine.ctx.call("tg_net_load_poll", 1, ctypes.byref(pend))
    h = model.HipNetwork(9, 10, 32, 2, rows_cap=4, device=0)
    h.set_weights(sd[1])
    probe = (np.random.RandomState(5).rand(3, 10, 9, 9) < 0.2).astype(np.float32)
    want = h.main_prediction(probe)
    pol = np.empty((3, 82), np.float32); val = np.empty(3, np.float32)
    actor.worker.engine.ctx.call("tg_net_predict", probe.ctypes.data_as(ctypes.c_void_p), 3, pol.ctypes.data_as(ctypes.c_void_p),
                                 val.ctypes.data_as(ctypes.c_void_p), None)
    out["weights_v2_on_gpu"] = bool(np.array_equal(pol, want[0]) and np.array_equal(val, want[1].reshape(-1)))
    out["games"] = st.get_info("now_play_games"); out["steps"] = st.get_info("now_play_steps")
    out["entries"] = mem.info()["entries"]
    q.put(out)
    dist.destroy_process_group()


def test_rccl_single_rank_device_paths():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["ctl"] == [1, 2] and out["bcast_device"] and out["weights_v2_on_gpu"]
    assert out["games"] == 4 and out["steps"] == 7 * 4 and out["entries"] == 4 * 5 * 8
