import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import evaluators
from transgo_amd.configure import Config
from transgo_amd.self_play import BatchedSelfPlay
cfg = Config(num_simulation=8, max_step=5, buffer_size=8 * 1024)
sp = BatchedSelfPlay(cfg, 4, rank=1, world=2, evaluator=evaluators.flat)
for i in range(11):
    h = sp.advance()
    print(i, "seeds", sp.seeds, "fin", None if h is None else (h.n_games, list(h.view("slot")), list(h.view("n_moves"))),
          "dropped", sp.games_dropped, "errored", sp.engine.errored, "finished", sp.engine.finished, "err", sp.engine.game_errors())
print(sp.games_finished, sp.engine.stats())
