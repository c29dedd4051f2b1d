"""fp32 torch restatement of the policy/value/ownership tower.  TEST INFRASTRUCTURE ONLY (parity oracle for net.hip and
the CPU baseline's evaluator).  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.

Built from the reference's building blocks -- CNNBlock (model.py:317-324), pre-activation ResidualBlock identity branch
(model.py:238-248), tail BN+ReLU (model.py:62,94), value/ownership head (model.py:65-69, :97-102) and policy head
(model.py:73-76, :107-111) -- with the block count and width as parameters, because BASELINE.json's "N-block x
F-filter" nets cannot be expressed by the reference's hard-coded 9+3 layout (model.py:49-61).  `attention=True` inserts
the reference's Self_Attention (model.py:288-315) in the policy head as model.py:72,106 does.
Pinned against the imported reference modules by tests/test_oracle_net.py (golden: tests/golden/net_blocks.npz).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvBnRelu(nn.Module):                 # CNNBlock, model.py:317-324
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class PreActBlock(nn.Module):                # ResidualBlock with input_dim == output_dim, model.py:238-248
    def __init__(self, f):
        super().__init__()
        self.batchnormlize_1 = nn.BatchNorm2d(f)
        self.conv_1 = nn.Conv2d(f, f, 3, 1, 1)
        self.batchnormlize_2 = nn.BatchNorm2d(f)
        self.conv_2 = nn.Conv2d(f, f, 3, 1, 1)

    def forward(self, x):
        y = self.conv_1(F.relu(self.batchnormlize_1(x)))
        y = self.conv_2(F.relu(self.batchnormlize_2(y)))
        return x + y


class SelfAttention(nn.Module):              # Self_Attention, model.py:288-315
    def __init__(self, f):
        super().__init__()
        self.query_conv = nn.Conv2d(f, f // 4, 1)
        self.key_conv = nn.Conv2d(f, f // 4, 1)
        self.value_conv = nn.Conv2d(f, f, 1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.bn = nn.BatchNorm2d(f)

    def forward(self, x):
        b, c, w, h = x.size()
        q = self.query_conv(x).view(b, -1, w * h).permute(0, 2, 1)
        k = self.key_conv(x).view(b, -1, w * h)
        att = torch.softmax(torch.bmm(q, k), dim=-1)
        v = self.value_conv(x).view(b, -1, w * h)
        out = torch.bmm(v, att).view(b, c, w, h)          # sums over the softmaxed row index, as the reference does
        return F.relu(self.bn(self.gamma * out + x))


class TransGoMainBody(nn.Module):
    """MainNetwork (model.py:41-114) with its module names, so the reference state_dict loads unchanged."""

    def __init__(self, board_size, input_dim, f):
        super().__init__()
        self.S = board_size
        P = board_size * board_size
        self.conv1 = ConvBnRelu(input_dim, f)
        for i in range(2, 14):
            setattr(self, f"res_conv{i}", SelfAttention(f) if i in (3, 7, 12) else PreActBlock(f))
        self.bn_res_end = nn.BatchNorm2d(f)
        self.conv_val_own = ConvBnRelu(f, 2)
        self.fc_val_own = nn.Linear(2 * P, 64)
        self.fc_val = nn.Linear(64, 1)
        self.fc_own = nn.Linear(64, P)
        self.attention_act = SelfAttention(f)
        self.conv_act = ConvBnRelu(f, 4)
        self.fc_act = nn.Linear(4 * P, P + 1)

    def forward(self, x):
        P = self.S * self.S
        x = self.conv1(x)
        for i in range(2, 14):
            x = getattr(self, f"res_conv{i}")(x)
        x = F.relu(self.bn_res_end(x))
        h = F.relu(self.fc_val_own(self.conv_val_own(x).view(-1, 2 * P)))
        val = torch.tanh(self.fc_val(h))
        own = torch.tanh(self.fc_own(h))
        act = torch.softmax(self.fc_act(self.conv_act(self.attention_act(x)).view(-1, 4 * P)), -1)
        return act, val, own


class TransGoMain(nn.Module):
    """TransGoNetwork surface (model.py:11-27) around TransGoMainBody."""

    def __init__(self, board_size=9, input_dim=10, filters=128):
        super().__init__()
        self.main_network = TransGoMainBody(board_size, input_dim, filters)

    def main_prediction(self, state):
        return self.main_network(state)


class TowerBody(nn.Module):
    def __init__(self, board_size, input_dim, filters, blocks):
        super().__init__()
        self.S = board_size
        P = board_size * board_size
        self.conv1 = ConvBnRelu(input_dim, filters)
        self.res_blocks = nn.ModuleList([PreActBlock(filters) for _ in range(blocks)])
        self.bn_res_end = nn.BatchNorm2d(filters)
        self.conv_val_own = ConvBnRelu(filters, 2)
        self.fc_val_own = nn.Linear(2 * P, 64)
        self.fc_val = nn.Linear(64, 1)
        self.fc_own = nn.Linear(64, P)
        self.conv_act = ConvBnRelu(filters, 4)
        self.fc_act = nn.Linear(4 * P, P + 1)

    def forward(self, x):
        P = self.S * self.S
        x = self.conv1(x)
        for b in self.res_blocks:
            x = b(x)
        x = F.relu(self.bn_res_end(x))
        h = F.relu(self.fc_val_own(self.conv_val_own(x).view(-1, 2 * P)))
        val = torch.tanh(self.fc_val(h))
        own = torch.tanh(self.fc_own(h))
        act = torch.softmax(self.fc_act(self.conv_act(x).view(-1, 4 * P)), -1)
        return act, val, own


class TowerNetwork(nn.Module):
    """Same surface as TransGoNetwork (model.py:11-27)."""

    def __init__(self, board_size=9, input_dim=10, filters=128, blocks=6):
        super().__init__()
        self.main_network = TowerBody(board_size, input_dim, filters, blocks)

    def main_prediction(self, state):
        return self.main_network(state)

    def get_weights(self):
        return {k: v.cpu() for k, v in self.state_dict().items()}

    def set_weights(self, weights):
        self.load_state_dict(weights)


def seeded_tower(board_size=9, input_dim=10, filters=128, blocks=6, seed=1234):
    """SURVEY.md §8d synthetic weights: torch default init under manual_seed, BN running_mean ~ N(0, 0.1),
    running_var ~ U(0.5, 1.5)."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    net = TowerNetwork(board_size, input_dim, filters, blocks).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    return net


def half_storage_forward(net, x, half_residual=False):
    """What an fp16-storage / f32-accumulate evaluation of `net` (a TowerNetwork) computes, with the rounding points of the
    HIP fp16 chain (BASELINE config 5): every convolution (stem, tower, the two head convs) takes fp16 weights (BatchNorm
    folded in f64, stored as f32, then rounded to nearest even) and fp16 inputs, products accumulate in f32; the residual
    stream and the dense heads stay f32.  Conv inputs: the 0/1 planes (exact), relu(bn1(x)) and relu(bn2(conv1(.))) inside a
    PreActBlock, relu(bn_res_end(x)) for the head convs.  Only the accumulation order inside a convolution is left free.
    half_residual=True emulates net_precision 2: the residual stream is stored as fp16 as well -- a block's f32 result v feeds the
    next activation unrounded, is rounded once into the stream, and the next block adds its convolution to that rounded value."""
    body = net.main_network
    P = body.S * body.S
    q = lambda t: t.half().float()

    def fold(bn):
        s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        return s, bn.bias.double() - bn.running_mean.double() * s

    def conv_bn_relu(block, inp):                      # ConvBnRelu with its BN folded into the fp16 weights
        conv, bn = block.conv[0], block.conv[1]
        s, t = fold(bn)
        w = (conv.weight.double() * s[:, None, None, None]).float()
        b = (conv.bias.double() * s + t).float()
        return F.relu(F.conv2d(inp, q(w), b, 1, 1))

    with torch.no_grad():
        y = conv_bn_relu(body.conv1, x)
        res = q(y) if half_residual else y                 # what the next block reads back as its residual
        for b in body.res_blocks:
            s1, t1 = fold(b.batchnormlize_1)
            s2, t2 = fold(b.batchnormlize_2)
            w1 = (b.conv_1.weight.double() * s2[:, None, None, None]).float()
            b1 = (b.conv_1.bias.double() * s2 + t2).float()
            a = q(F.relu(y * s1.float()[None, :, None, None] + t1.float()[None, :, None, None]))
            h = q(F.relu(F.conv2d(a, q(w1), b1, 1, 1)))
            y = F.conv2d(h, q(b.conv_2.weight), b.conv_2.bias, 1, 1) + res
            res = q(y) if half_residual else y
        se, te = fold(body.bn_res_end)
        z = q(F.relu(y * se.float()[None, :, None, None] + te.float()[None, :, None, None]))
        hid = F.relu(body.fc_val_own(conv_bn_relu(body.conv_val_own, z).view(-1, 2 * P)))
        val = torch.tanh(body.fc_val(hid))
        own = torch.tanh(body.fc_own(hid))
        act = torch.softmax(body.fc_act(conv_bn_relu(body.conv_act, z).view(-1, 4 * P)), -1)
    return act, val, own
