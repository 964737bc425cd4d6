"""Host logic of the flat parameter buffer (trainer.FlatParams): every parameter becomes a view, groups a module declares
(`adjacent_parameter_groups`, MultiHeadAttention's W_Q|W_K|W_V) sit back to back so that `layers.stacked_view` forms one
matrix without a copy, and names / shapes / values of the state_dict stay the reference's (layers.py:102-148)."""
import torch
import torch.nn as nn

from nnr_amd.layers import MultiHeadAttention, grad_of, stacked_view
from nnr_amd.trainer import FlatParams


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.pre = nn.Linear(3, 5)                       # 15 + 5 floats: exercises the 4-float padding of the offsets
        self.mha = MultiHeadAttention(4, 12, 6, 6, 8, 8)
        self.post = nn.Linear(7, 2)
        self.frozen = nn.Parameter(torch.ones(3), requires_grad=False)


def test_groups_are_adjacent_and_values_survive():
    torch.manual_seed(0)
    net = _Net()
    net.mha.initialize()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    ws = [net.mha.W_Q.weight, net.mha.W_K.weight, net.mha.W_V.weight]
    bs = [net.mha.W_Q.bias, net.mha.W_K.bias, net.mha.W_V.bias]
    assert stacked_view(ws) is None                      # separate storages before re-homing: callers take the per-tensor path
    flat = FlatParams(net)
    after = net.state_dict()
    assert list(after) == list(before)
    for k in before:
        assert torch.equal(after[k], before[k]), k
    w, b = stacked_view(ws), stacked_view(bs)
    assert w.shape == (96, 12) and b.shape == (96,)
    assert w.data_ptr() == net.mha.W_Q.weight.data_ptr() and torch.equal(w[32:64], net.mha.W_K.weight) and torch.equal(b[64:], net.mha.W_V.bias)
    gw = stacked_view([grad_of(p) for p in ws])
    gw[64:] += 1.0                                       # the stacked gradient view aliases the per-parameter gradients
    assert float(net.mha.W_V.weight.grad.sum()) == 32 * 12 and float(net.mha.W_Q.weight.grad.abs().sum()) == 0.0
    # every trainable parameter is a 16-byte aligned view into the one buffer; frozen ones are left alone
    lo, hi = flat.flat.data_ptr(), flat.flat.data_ptr() + 4 * flat.numel
    for name, p in net.named_parameters():
        if p.requires_grad:
            assert lo <= p.data_ptr() < hi and p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0, name
        else:
            assert not (lo <= p.data_ptr() < hi), name
    flat.grad.fill_(2.0)
    assert all(float(p.grad.min()) == 2.0 for p in net.parameters() if p.requires_grad)
    flat.zero_grad()
    assert float(flat.grad.abs().sum()) == 0.0


def test_stacked_view_rejects_gaps_and_mismatches():
    buf = torch.arange(64, dtype=torch.float32)
    a, b, c = buf[0:8].view(2, 4), buf[8:16].view(2, 4), buf[16:24].view(2, 4)
    assert torch.equal(stacked_view([a, b, c]), buf[:24].view(6, 4))
    assert stacked_view([a, c]) is None                                  # a gap
    assert stacked_view([a, buf[8:20].view(3, 4)]) is None               # shape mismatch
    assert stacked_view([a, torch.zeros(2, 4)]) is None                  # another storage
    assert stacked_view([buf[0:16].view(4, 4)[:, :2], buf[16:24].view(4, 2)]) is None   # not contiguous
