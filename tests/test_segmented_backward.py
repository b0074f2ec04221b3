"""CPU: ddp.SegmentedBackward (the backward of the data-parallel graphed step, cut at runtime.cut_point tensors) gives the same gradients
as one loss.backward(), including a parameter shared by several branches / segments, and reports every parameter final exactly once."""
import torch

import tav_amd  # noqa: F401
from tav_amd import runtime
from tav_amd.ddp import SegmentedBackward


class _Branch(torch.nn.Module):
    def __init__(self, name, L):
        super().__init__()
        self.name = name
        self.ls = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(L)])

    def forward(self, x):
        for i, l in enumerate(self.ls):
            x = runtime.cut_point(self.name, i, len(self.ls), x)        # what the encoder stacks do
            x = torch.tanh(l(x))
        return x


def _toy():
    torch.manual_seed(0)
    bs = torch.nn.ModuleList([_Branch("a", 12), _Branch("b", 6), _Branch("c", 3)])
    emb = torch.nn.Parameter(torch.randn(8))                              # shared by all branches: reached in several segments
    head = torch.nn.Linear(24, 3)
    x0 = torch.randn(4, 8)

    def fwd():
        return head(torch.cat([b(x0 + emb) for b in bs], 1)).square().mean()
    return fwd, [emb] + list(bs.parameters()) + list(head.parameters())


def test_segments_equal_whole_backward():
    fwd, params = _toy()
    fwd().backward()
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    runtime.begin_cuts()
    loss = fwd()
    cuts = runtime.end_cuts()
    assert {k: len(v) for k, v in cuts.items()} == {"a": 3, "b": 3, "c": 2}     # depth fractions 1/12, 1/3, 2/3 of 12 / 6 / 3 layers -> layers {1,4,8} / {1,2,4} / {1,2}
    sb = SegmentedBackward(loss, cuts)
    assert sb.nseg == 4
    done = []
    for s in range(sb.nseg):
        fin = sb.run(s)
        assert all(p.grad is not None for p in fin)
        done += fin
    assert sorted(map(id, done)) == sorted(map(id, params))             # every parameter final exactly once
    assert id(params[0]) in {id(p) for p in sb.final[-1]}               # the shared one only after the last segment
    for p, r in zip(params, ref):
        assert torch.allclose(p.grad, r, rtol=1e-6, atol=1e-7)


def test_no_recorder_no_cut():
    fwd, params = _toy()
    loss = fwd()                                                        # cut_point is the identity outside begin_cuts()/end_cuts()
    sb = SegmentedBackward(loss, {})
    assert sb.nseg == 1
    assert sorted(map(id, sb.run(0))) == sorted(map(id, params))
