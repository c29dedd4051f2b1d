#!/usr/bin/env python3
"""Self-play -> replay -> one training step, end to end on one GPU, with the trainer's own batch code.

The engine plays complete games (records kept in HBM), finished games go device -> device into DeviceReplayMemory, and a
stand-in trainer consumes it the way trainer.py:46-72 consumes the reference buffer: four float32 tensors (state, pi, z, own) per
mini-batch and the reference's loss (value MSE + 0.75 own MSE + 1.15 policy cross-entropy + 0.02 entropy term).  Two ways to get
the batch are shown: `mem.sample(B)` (host arrays, fed through the reference's np.stack / torch.FloatTensor lines unchanged) and
`mem.sample_device(B)` (the same tensors already on the GPU).  The network trained here is a throw-away torch module with the
tower's state_dict layout; its weights go back into the engine through the same `set_weights` the reference actor calls.

    python examples/selfplay_to_trainer.py            # needs an MI355X; ~10 s
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transgo_amd.configure import Config                         # noqa: E402
from transgo_amd.replay_buffer import DeviceReplayMemory         # noqa: E402
from transgo_amd.self_play import BatchedSelfPlay                # noqa: E402


class Block(nn.Module):                                           # pre-activation ResidualBlock (model.py:238-248)
    def __init__(self, f):
        super().__init__()
        self.batchnormlize_1, self.conv_1 = nn.BatchNorm2d(f), nn.Conv2d(f, f, 3, 1, 1)
        self.batchnormlize_2, self.conv_2 = nn.BatchNorm2d(f), nn.Conv2d(f, f, 3, 1, 1)

    def forward(self, x):
        return x + self.conv_2(F.relu(self.batchnormlize_2(self.conv_1(F.relu(self.batchnormlize_1(x))))))


class CBR(nn.Module):                                             # CNNBlock (model.py:317-324)
    def __init__(self, i, o):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(i, o, 3, 1, 1), nn.BatchNorm2d(o), nn.ReLU())

    def forward(self, x):
        return self.conv(x)


class Tower(nn.Module):                                           # the N-block x F-filter tower, MainNetwork's key names
    def __init__(self, S, C, f, n):
        super().__init__()
        P = S * S
        self.P = P
        self.conv1 = CBR(C, f)
        self.res_blocks = nn.ModuleList([Block(f) for _ in range(n)])
        self.bn_res_end = nn.BatchNorm2d(f)
        self.conv_val_own, self.fc_val_own, self.fc_val, self.fc_own = CBR(f, 2), nn.Linear(2 * P, 64), nn.Linear(64, 1), nn.Linear(64, P)
        self.conv_act, self.fc_act = CBR(f, 4), nn.Linear(4 * P, P + 1)

    def forward(self, x):
        x = self.conv1(x)
        for b in self.res_blocks:
            x = b(x)
        x = F.relu(self.bn_res_end(x))
        h = F.relu(self.fc_val_own(self.conv_val_own(x).flatten(1)))
        return torch.softmax(self.fc_act(self.conv_act(x).flatten(1)), -1), torch.tanh(self.fc_val(h)), torch.tanh(self.fc_own(h))


class Net(nn.Module):
    def __init__(self, *a):
        super().__init__()
        self.main_network = Tower(*a)

    def main_prediction(self, x):
        return self.main_network(x)


def loss_fn(net, state, pi, z, own_z):                            # trainer.py:57-70
    act_probs, value, own = net.main_prediction(state)
    value_loss = F.mse_loss(value.view(-1), z)
    own_loss = F.mse_loss(own, own_z)
    act_policy_loss = -torch.mean(torch.sum(pi * torch.log(act_probs), 1))
    entropy_loss = torch.mean(torch.sum(act_probs * torch.log(act_probs), 1))
    return value_loss + 0.75 * own_loss + 1.15 * act_policy_loss + 0.02 * entropy_loss


def main(games=64, sims=32, max_step=20, filters=32, blocks=2, batch=256, steps=3):
    dev = torch.device("cuda", 0)
    cfg = Config(num_simulation=sims, max_step=max_step, num_features=filters, num_blocks=blocks, buffer_size=8 * 65536)
    torch.manual_seed(0)
    net = Net(9, 10, filters, blocks).to(dev)
    sp = BatchedSelfPlay(cfg, games)
    sp.set_weights({k: v.detach().cpu() for k, v in net.eval().state_dict().items()})      # model.get_weights() -> actor (model.py:23-27)
    mem = DeviceReplayMemory(cfg, capacity_positions=65536)
    while sp.games_finished < games:                             # one generation of complete games
        h = sp.advance(device=True)
        if h is not None:
            mem.append_harvest(h)                                # HBM -> HBM
    print("self-play:", sp.games_finished, "games,", mem.info()["entries"], "replay entries (8 per position)")
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    net.train()
    losses = []
    for it in range(steps):
        if it == 0:                                              # the reference's own lines (trainer.py:46-54) on host tuples
            state, pi, z, own = (torch.FloatTensor(a).to(dev) for a in mem.sample(batch))
        else:                                                    # the same batch without the host round trip
            state, pi, z, own = mem.sample_device(batch)
        opt.zero_grad()
        loss = loss_fn(net, state, pi, z, own)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    print("losses:", [round(x, 4) for x in losses])
    sp.set_weights({k: v.detach().cpu() for k, v in net.eval().state_dict().items()})      # trainer.py:76-79 -> self_play.py:913
    sp.advance()
    print("selfplay_to_trainer ok")
    return losses


if __name__ == "__main__":
    main()
