"""Fixture for the storage interfaces: the reference's ReplayMemory_Random (replay_buffer.py:16-94) and
SharedStorage (shared_storage.py:6-43) driven through a scripted scenario with the `ray` stub of ref_harness.py; the observable
results (info() after every phase, the z-values of seeded samples with and without replacement, the save()/load() bookkeeping,
the learning-rate and train/play-ratio schedules of configure.py:88-103) are recorded as data in tests/golden/storage.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402


def scenario(ReplayCls, StorageCls, cfg):
    out = {}
    cfg.buffer_size = 50
    mem = ReplayCls(cfg)
    infos, samples = [], []
    tup = lambda i: (np.full((cfg.encode_state_channels, cfg.board_size, cfg.board_size), i, np.float32),
                     np.full(cfg.board_size ** 2 + 1, 1.0 / (cfg.board_size ** 2 + 1)), float(i), np.full(cfg.board_size ** 2, -float(i)))
    n = 0
    for phase, count in enumerate((7, 30, 40)):                     # under-full (with replacement), partial, wrapped
        for _ in range(count):
            mem.append(*tup(n)); n += 1
        i = mem.info(); infos.append([i["capacity"], i["index"], int(i["full"])])
        np.random.seed(100 + phase)
        for bs in (16, 5):
            batch = mem.sample(bs)
            samples.append(np.array([t[2] for t in batch] + [-1.0] * (16 - bs)))
    out["infos"] = np.array(infos); out["samples"] = np.array(samples)
    sv = mem.save()
    out["save_meta"] = np.array([sv["buffer_capacity"], sv["index"], int(sv["full"]), sv["save_len"]])
    # the pickled layout other tools (and the other implementation) see: data is a 2-D object array, one column per tuple field
    out["save_data_shape"] = np.array(np.shape(sv["data"]))
    out["save_z"] = np.array([float(row[2]) for row in sv["data"]])
    out["save_obs00"] = np.array([float(row[0][0, 0, 0]) for row in sv["data"]])
    cfg.buffer_size = 10
    part = ReplayCls(cfg)                                           # part-filled: the unfilled rows are blank tuples
    for i in range(4):
        part.append(*tup(i))
    pv = part.save()
    out["partial_shape"] = np.array(np.shape(pv["data"]))
    out["partial_z"] = np.array([float(row[2]) for row in pv["data"]])
    out["partial_pi_len"] = np.array([len(row[1]) for row in pv["data"]])
    # a save() dict spelt out in that layout by hand loads into the class
    hand = {"buffer_capacity": 10, "index": 3, "full": False, "save_len": 3,
            "data": np.array([tup(20), tup(21), tup(22)], dtype=object)}
    out["hand_shape"] = np.array(hand["data"].shape)
    tgt = ReplayCls(cfg)
    tgt.load(hand)
    out["hand_loaded_z"] = np.array([float(row[2]) for row in tgt.data[:3]])
    cfg.buffer_size = 120
    big = ReplayCls(cfg)
    fulls = []
    for _ in range(3):                                              # 50 + 50 < 120, then the third load wraps
        fulls.append(int(bool(big.load(sv))))
        i = big.info(); fulls += [i["index"], int(i["full"])]
    out["load_trace"] = np.array(fulls)
    st = StorageCls({"weights": None, "now_play_steps": 0, "now_play_games": 0, "learn_rate": 6.5e-5, "adjust_lr": True,
                     "train_play_ratio": 0.075, "adjust_train_play_ratio": True, "now_train_steps": 0}, cfg)
    lr, ratio = [], []
    for g in range(3200):
        st.set_info("now_play_games")
        for _ in range(3):
            st.set_info("now_play_steps")
        if g % 100 == 99 or g in (1498, 1499, 2998, 2999):
            lr.append(st.get_info("learn_rate")); ratio.append(st.get_info("train_play_ratio"))
    out["lr"] = np.array(lr, np.float64); out["ratio"] = np.array(ratio, np.float64)
    d = st.get_info(["now_play_games", "now_play_steps"])
    out["counters"] = np.array([d["now_play_games"], d["now_play_steps"]])
    st.set_info({"weights": 5, "now_train_steps": 9}); st.set_info("learn_rate", 1e-3)
    out["after_set"] = np.array([st.get_info("weights"), st.get_info("now_train_steps"), st.get_info("learn_rate")], np.float64)
    return out


def main():
    R = ref_harness.load_reference()
    import replay_buffer, shared_storage                           # the reference modules (ray.remote = identity)
    out = scenario(replay_buffer.ReplayMemory_Random, shared_storage.SharedStorage, R.Config())
    np.savez_compressed(os.path.join(HERE, "storage.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
