"""oracle/net.py against outputs recorded from the imported reference model (tests/golden/net_transgo_f32.npz): the full
MainNetwork restatement (attention included) and the building blocks the parametrised tower is assembled from."""
import os

import numpy as np
import torch

from oracle.net import ConvBnRelu, PreActBlock, TransGoMain


def _load(golden_dir):
    with np.load(os.path.join(golden_dir, "net_transgo_f32.npz")) as z:
        return {k: z[k] for k in z.files}


def test_transgo_main_matches_reference(golden_dir):
    b = _load(golden_dir)
    torch.set_num_threads(1)
    net = TransGoMain(9, 10, 32).eval()
    # the reference ResidualBlock also owns a conv_shortcut that its identity branch never uses (model.py:233,239-248)
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in b.items() if k.startswith("sd/") and "conv_shortcut" not in k})
    with torch.no_grad():
        p, v, o = net.main_prediction(torch.from_numpy(b["x"]))
    assert np.abs(p.numpy() - b["policy"]).max() < 1e-6
    assert np.abs(v.numpy() - b["value"]).max() < 1e-6 and np.abs(o.numpy() - b["own"]).max() < 1e-6


def test_tower_building_blocks_match_reference(golden_dir):
    b = _load(golden_dir)
    torch.set_num_threads(1)
    cnn = ConvBnRelu(10, 32).eval()
    cnn.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in b.items() if k.startswith("cnn/")})
    blk = PreActBlock(32).eval()
    sd = {k[4:]: torch.from_numpy(v) for k, v in b.items() if k.startswith("blk/") and "conv_shortcut" not in k}
    blk.load_state_dict(sd)
    with torch.no_grad():
        h = cnn(torch.from_numpy(b["x"][:4]))
        hb = blk(h)
    assert np.abs(h.numpy() - b["cnn_out"]).max() < 1e-6 and np.abs(hb.numpy() - b["blk_out"]).max() < 1e-6
