"""F3 network fixture: the imported reference TransGoNetwork (model.py:11-114) at num_features=32 with seeded weights,
randomised BatchNorm statistics and non-zero attention gains, evaluated on real encoded positions.  Stores the
state_dict (as float32 arrays), the inputs and the three outputs; plus one ResidualBlock / CNNBlock in-out pair that pins
the building blocks the parametrised tower oracle (oracle/net.py) is assembled from."""
import os

import numpy as np
import torch


def _positions(R, n, seed):
    env = R.environment.GoEnv(R.cfg)
    rng = np.random.RandomState(seed)
    obs = []
    while len(obs) < n:
        s, done = env.reset()
        while not done and len(obs) < n:
            la = env.getLegalAction(s)
            s, done = env.step(s, int(la[rng.randint(len(la))]))
            if rng.rand() < 0.3:
                obs.append(env.encode(s))
    return np.stack(obs)


def run(R, outdir):
    torch.set_num_threads(1)
    torch.manual_seed(77)
    cfg = R.Config(); cfg.device = torch.device("cpu"); cfg.num_features = 32
    net = R.model.TransGoNetwork(cfg).eval()
    g = torch.Generator().manual_seed(78)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            if hasattr(m, "gamma"):
                m.gamma.copy_(0.5 + torch.rand(1, generator=g))
    x = _positions(R, 12, 5)
    with torch.no_grad():
        p, v, o = net.main_prediction(torch.from_numpy(x))
        blk = R.model.ResidualBlock(32, 32).eval(); cnn = R.model.CNNBlock(10, 32).eval()
        for m in list(blk.modules()) + list(cnn.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
        h = cnn(torch.from_numpy(x[:4]))
        hb = blk(h)
    blob = {"sd/" + k: t.numpy() for k, t in net.state_dict().items()}
    blob.update({"blk/" + k: t.numpy() for k, t in blk.state_dict().items()})
    blob.update({"cnn/" + k: t.numpy() for k, t in cnn.state_dict().items()})
    blob.update(x=x, policy=p.numpy(), value=v.numpy(), own=o.numpy(), cnn_out=h.numpy(), blk_out=hb.numpy())
    np.savez_compressed(os.path.join(outdir, "net_transgo_f32.npz"), **blob)
    print("net fixture:", sum(v.size for k, v in blob.items() if k.startswith("sd/")), "weights,", len(x), "positions")
