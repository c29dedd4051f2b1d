"""ReplayMemory_Random with the reference's interface (replay_buffer.py:16-94): same method names, argument order and
tuple layout `(obs f32[C,S,S], pi f64[A], z float, own f64[S*S])`, so trainer.py:46-54 (`map(np.stack, zip(*batch))`)
consumes it unchanged.  Host-side container, not a Ray actor: wrap it with `ray.remote(ReplayMemory_Random)` where Ray
exists; transgo_amd.self_play calls `.remote()` when the object offers it."""
import numpy as np


class ReplayMemory_Random:
    def __init__(self, config):
        self.capacity = int(config.buffer_size)
        self.full = False
        self.index = 0
        self.last_save_index = 0
        self.load_index = 0
        self.data = np.empty(self.capacity, dtype=object)          # replay_buffer.py:25-27 (blank tuples on demand)
        self._blank = (np.zeros((config.encode_state_channels, config.board_size, config.board_size)),
                       np.zeros((config.board_size ** 2 + 1)), 0.0, np.zeros((config.board_size ** 2)))
        self.data[:] = [self._blank] * self.capacity if self.capacity <= 4096 else None

    def append(self, observation, act_prob, win_z, own_z):          # replay_buffer.py:30-34
        self.data[self.index] = (observation, act_prob, win_z, own_z)
        self.index = (self.index + 1) % self.capacity
        self.full = self.full or self.index == 0

    def sample(self, batch_size):                                   # replay_buffer.py:36-47
        buffer_len = self.capacity if self.full else self.index
        if buffer_len < batch_size:
            idx = np.random.choice(buffer_len, batch_size)
        else:
            idx = np.random.choice(buffer_len, batch_size, replace=False)
        return self.data[idx]

    def save(self):                                                 # replay_buffer.py:49-72
        if self.capacity <= 1500000:
            return {"buffer_capacity": self.capacity, "index": self.index, "full": self.full, "data": self.data,
                    "save_len": self.capacity}
        if 0 < self.index - self.last_save_index < 300000:
            return False
        if self.index - self.last_save_index < 0 and self.index < 300000:
            return False
        save_len = min(1000000, self.index)
        out = {"buffer_capacity": self.capacity, "data": self.data[self.index - save_len:self.index],
               "save_len": save_len, "index": self.index}
        self.last_save_index = self.index
        return out

    def load(self, infos):                                          # replay_buffer.py:74-87
        save_len = infos["save_len"]
        if self.load_index + save_len < self.capacity:
            self.data[self.load_index:self.load_index + save_len] = infos["data"]
            self.load_index += save_len
        else:
            end_len = self.capacity - self.load_index
            self.data[self.load_index:] = infos["data"][:end_len]
            self.load_index = 0
            self.full = True
        self.index = self.load_index
        return self.full

    def info(self):                                                 # replay_buffer.py:89-94
        return {"capacity": self.capacity, "index": self.index, "full": self.full}


def ReplayMemory(config):                                           # replay_buffer.py:7-10
    return ReplayMemory_Random(config)
