"""Which ATen ops (not libtavhip kernels) does one training step launch, and from where?  torch.profiler over one eager step at batch 32,
grouped by op name and Python call site.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import tav_amd  # noqa: F401,E402
from tav_amd import config as C, runtime, synthetic  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.train_model.tav_train import TrainStep  # noqa: E402
from tav_amd.utils.global_functions import CrossEntropyLoss  # noqa: E402

cfg = C.preset("B")
runtime.set_precision("bf16")
B = int(os.environ.get("TAV_B", "8"))
pre, model = PreFormer(cfg).cuda(), TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg).cuda()
inp, lab = synthetic.make_batch(cfg, B, device="cuda")
st = TrainStep(model, pre, CrossEntropyLoss(), lr=1e-6)
for _ in range(2):
    st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=104)
    st.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=104)
    st.update()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    stack = [s for s in e.stack if "multi-modal-emotion_amd" in s or "tav_amd" in s][:2]
    print(f"{e.count:5d} x {e.key:34s} dev {e.device_time_total / 1e3:8.3f} ms  {' <- '.join(s.split('/')[-1] for s in stack)}")
