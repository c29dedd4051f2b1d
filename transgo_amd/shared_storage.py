"""SharedStorage with the reference's interface and key semantics (shared_storage.py:6-43): get_info(key | list),
set_info(key, value) / set_info(dict) / the increment form set_info("now_play_steps" | "now_play_games") that also
drives the learning-rate and train/play-ratio schedules (configure.py:90-103)."""
import copy


class SharedStorage:
    def __init__(self, checkpoint, config):
        self.config = config
        self.current_checkpoint = copy.deepcopy(checkpoint)

    def get_info(self, keys):                                       # shared_storage.py:13-19
        if isinstance(keys, str):
            return self.current_checkpoint[keys]
        if isinstance(keys, list):
            return {k: self.current_checkpoint[k] for k in keys}
        raise TypeError

    def set_info(self, keys, values=None):                          # shared_storage.py:21-43
        cp = self.current_checkpoint
        if keys == "now_play_steps" or keys == "now_play_games":
            cp[keys] += 1
        elif isinstance(keys, str) and values is not None:
            cp[keys] = values
        elif isinstance(keys, dict):
            cp.update(keys)
        else:
            raise TypeError(keys)
        if keys == "now_play_games" and cp.get("adjust_lr") is True:
            cp["learn_rate"] = self.config.ad_lr(cp["now_play_games"], cp["learn_rate"])
        if keys == "now_play_steps" and cp.get("adjust_train_play_ratio") is True and cp["now_play_games"] > 0:
            cp["train_play_ratio"] = self.config.ad_train_play_ratio(cp["now_play_steps"], cp["train_play_ratio"])

    def add_info(self, key, n):
        """n times the increment form set_info(key) in one call (one RPC per move of G games instead of G): the counter and
        both schedules end exactly where n separate calls would leave them -- the schedules only fire on particular counter
        values, so those are the only iterations evaluated."""
        cp = self.current_checkpoint
        if key not in ("now_play_steps", "now_play_games") or n < 0:
            raise TypeError(key)
        if key == "now_play_games":
            if cp.get("adjust_lr") is True:
                for _ in range(int(n)):                    # 1500-game periods, at most two halvings: cheap, keep it literal
                    cp[key] += 1
                    cp["learn_rate"] = self.config.ad_lr(cp[key], cp["learn_rate"])
            else:
                cp[key] += int(n)
            return
        end = cp[key] + int(n)
        if cp.get("adjust_train_play_ratio") is True and cp["now_play_games"] > 0:
            v = cp[key] + 1
            v += (5 - v) % 6                               # first counter value >= v with (value + 1) % 6 == 0
            while v <= end:
                cp["train_play_ratio"] = self.config.ad_train_play_ratio(v, cp["train_play_ratio"])
                v += 6
        cp[key] = end
