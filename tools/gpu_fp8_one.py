"""Config 5 at full depth (24 video layers), batch 2, ONE policy setting from the environment (TAV_FP8_FWD_MASK / TAV_FP8_BWD_MASK / TAV_ATTN_PRESCALE /
TAV_FP8_DELAYED ...) against the fp32 oracle: logits / loss / grad-norm errors.  usage: python tools/gpu_fp8_one.py [policy] [label] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

torch.set_num_threads(16)
import test_model_gpu as T  # noqa: E402
from tav_amd import config as C  # noqa: E402
from tav_amd import runtime  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.optim import grad_norm  # noqa: E402

pol = sys.argv[1] if len(sys.argv) > 1 else "fp8"
label = sys.argv[2] if len(sys.argv) > 2 else ""
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = C.preset("B5")
cfg, batch, lab, (sdp, sdm), o_logits, o_loss, o_gn, o_grads = T._oracle_full("B5", seed=seed, batch_size=2, cfg=cfg, tag="B5-full")
label = f"{label} seed {seed}"
runtime.set_precision(pol)
pre, model = PreFormer(cfg), TAVForMAE(T.ARGS, cfg)
pre.load_state_dict(sdp)
model.load_state_dict(sdm)
pre.cuda()
model.cuda()
_, _, _, logits, loss = T._run_product(pre, model, batch, lab)
loss.backward()
torch.cuda.synchronize()
gn = grad_norm(list(pre.parameters()) + list(model.parameters())).item()
print(f"[config 5 full depth] {pol:5s} {label:40s}: logits {T.rel(logits, o_logits):.2e} loss {abs(loss.item() - o_loss) / abs(o_loss):.2e} grad-norm {abs(gn - o_gn) / o_gn:.2e}", flush=True)
