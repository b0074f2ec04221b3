"""Bitwise reproducibility of the step at the BENCHMARKED size (tests/test_model_gpu.py::test_backward_is_bitwise_deterministic runs 4 utterances
through 2-layer stacks): preset B, full depth, batch 32, the branches on their own streams -- forward + backward N times from the same state, logits
and every gradient compared bit for bit with the first run.  Written after the hunt of profiles/r04_experiments.md section 6 (a kernel with packed-f32
accumulators whose results moved when MFMA-heavy kernels of another stream shared its SIMDs): do the shipped kernels hold still under the concurrency of
the real step?  usage: python tools/gpu_determinism_full.py [runs=4] [batch=32] [preset=B]      (GPU box only)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import tav_amd  # noqa: E402,F401
from tav_amd import config as C  # noqa: E402
from tav_amd import runtime, synthetic  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.train_model.tav_train import TrainStep  # noqa: E402
from tav_amd.utils.global_functions import CrossEntropyLoss  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
preset = sys.argv[3] if len(sys.argv) > 3 else "B"
cfg = C.preset(preset)
runtime.set_precision("bf16")
torch.manual_seed(0)
pre, model = PreFormer(cfg), TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg)
synthetic.seeded_init_(pre, 1)
synthetic.seeded_init_(model, 2)
pre.cuda()
model.cuda()
st = TrainStep(model, pre, CrossEntropyLoss(), lr=0.0, weight_decay=0.0, clip=1.0)
s = torch.cuda.Stream()
ref, bad = None, 0
with torch.cuda.stream(s):
    inp, lab = synthetic.make_batch(cfg, batch, device="cuda")
    for r in range(runs):
        for p in st.params:
            p.grad = None
        loss = st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=104)
        torch.cuda.synchronize()
        got = [loss.detach().clone()] + [None if p.grad is None else p.grad.clone() for p in st.params]
        if ref is None:
            ref = got
            print(f"run 0: loss {loss.item():.6f}, {sum(g is not None for g in got) - 1} gradients, {sum(g.numel() for g in got if g is not None)} elements", flush=True)
            continue
        diff = [i for i, (a, b) in enumerate(zip(ref, got)) if (a is None) != (b is None) or (a is not None and not torch.equal(a, b))]
        bad += len(diff)
        print(f"run {r}: {len(diff)} tensors differ from run 0" + (f" (first: #{diff[0]}, max abs {(ref[diff[0]] - got[diff[0]]).abs().max().item():.3e})" if diff else ""), flush=True)
print("BITWISE REPRODUCIBLE" if bad == 0 else f"NOT reproducible: {bad} tensor comparisons failed")
sys.exit(0 if bad == 0 else 1)
