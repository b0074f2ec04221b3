"""GPU box: which ATen (non-libtavhip) device kernels does one training step launch, and from where?  torch.profiler over one eager step
(single stream), grouped by operator with the innermost repo frame of the caller."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import tav_amd
from tav_amd import config as C, runtime, synthetic
from tav_amd.models.tav import PreFormer, TAVForMAE
from tav_amd.train_model.tav_train import TrainStep
from tav_amd.utils.global_functions import CrossEntropyLoss
from torch.profiler import profile, ProfilerActivity

b = int(os.environ.get("TAV_B", "8"))
cfg = C.preset("B")
runtime.set_precision("bf16")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
torch.manual_seed(0)
pre, model = PreFormer(cfg), TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg)
pre.cuda(); model.cuda()
inp, lab = synthetic.make_batch(cfg, b, device="cuda")
st = TrainStep(model, pre, CrossEntropyLoss(), lr=1e-6, weight_decay=1e-4, clip=1.0)
for _ in range(2):
    st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=104); st.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=104); st.update()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    # only operators that launched a device kernel themselves
    if not any(k for k in e.kernels):
        continue
    frame = "?"
    for fr in (e.stack or []):
        if "/tav_amd/" in fr or "multi-modal-emotion_amd" in fr or "bench.py" in fr or "tools/" in fr:
            frame = fr.split("/")[-1][:80]
            break
    cnt[(e.name, frame)] += len(e.kernels)
print(f"ATen operators with device kernels in one step (batch {b}): {sum(cnt.values())} kernels")
for (name, frame), c in cnt.most_common(60):
    print(f"{c:5d}  {name:28s} {frame}")
