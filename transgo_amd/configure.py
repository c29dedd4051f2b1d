"""Config -- the hot-path parameters of the reference's configure.py:5-103 (same attribute names and defaults), plus the
tower size and the number of concurrent boards of the batched engine."""
import math


class Config:
    def __init__(self, **over):
        self.board_size = 9                      # configure.py:9
        self.encode_state_channels = 10          # configure.py:11
        self.komi = 7.5                          # configure.py:13
        self.black, self.white = 1, 2
        self.max_step = 120                      # configure.py:16
        self.buffer_size = 1500000               # configure.py:19
        self.game_total_num = 1e8                # configure.py:24
        self.c_puct1 = 3                         # configure.py:26
        self.c_puct2 = 0.05                      # configure.py:27
        self.num_simulation = 210                # configure.py:29
        self.wu_loss = 2                         # configure.py:32
        self.parallel_readouts = 4               # configure.py:33
        self.input_dim = self.encode_state_channels
        self.num_features = 128                  # configure.py:37
        self.num_blocks = 6                      # tower depth (BASELINE.json "N-block x F-filter")
        self.network = "tower"                   # "tower" | "transgo" (the shipped MainNetwork with attention, model.py:49-76)
        self.concurrent_games = 4096             # boards resident on one GPU
        self.game_groups = 1                     # K > 1: the boards of a GPU as K independent groups on their own HIP streams (GroupedSelfPlay)
        self.stagger_games = 0                   # T > 1: slot g starts its first game at step g mod T (BatchedSelfPlay.start): games end
                                                 # spread over T steps instead of all on one; 0 = all slots start together
        self.inference_dtype = "f32"             # "f16": fp16 weights/activations, f32 accumulate (BASELINE config 5; towers of 128/256
                                                 # filters); "f16r": the residual stream in fp16 as well (+4-6 %, error < 4e-4 over 40 blocks);
                                                 # "f32x3": split precision -- conv operands as fp16 hi + lo, f32 accumulate, ~1e-6 of f32
        self.batch_size = 2048
        self.train_play_ratio = 7500 / 100000    # configure.py:61
        self.adjust_train_play_ratio = True
        self.adjust_lr = True
        self.learn_rate = 6.5e-5
        for k, v in over.items():
            setattr(self, k, v)

    def epsilon_by_frame(self, game_step):       # configure.py:75-79
        return 0.65 + (1.0 - 0.65) * math.exp(-1. * game_step / 10)

    def ad_lr(self, now_play_games, current_lr):                      # configure.py:90-93
        if (now_play_games + 1) % 1500 == 0 and now_play_games < 3100 and current_lr > 0.5 * 0.5 * 6.5e-5:
            return current_lr * 0.5
        return current_lr

    def ad_train_play_ratio(self, now_play_steps, current):           # configure.py:97-103
        if (now_play_steps + 1) % 6 == 0 and current < 2.6 / 10:
            return (current * 100000 + 1) / 100000
        return current
