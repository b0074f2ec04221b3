"""Config 5 at full depth (24 video layers), batch 2: which e4m3 GEMMs cost how much of the 1e-2 parity budget?
Runs the product under Policy("fp8") with subsets of the layer's GEMMs on e4m3 (engine.Policy.fp8_fwd / fp8_bwd) against ONE oracle run
(~35 s of CPU).  Prints logits / loss / grad-norm errors per subset.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import test_model_gpu as T  # noqa: E402
from tav_amd import config as C  # noqa: E402
from tav_amd import runtime  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.optim import grad_norm  # noqa: E402

cfg = C.preset("B5")
rec = T._oracle_full("B5", seed=0, batch_size=2, cfg=cfg, tag="B5-full")
cfg, batch, lab, (sdp, sdm), o_logits, o_loss, o_gn, o_grads = rec
cases = [("bf16", None, None)] + [("fp8", f, b) for f, b in ((0, 15), (1, 0), (2, 0), (4, 0), (8, 0), (15, 0), (15, 15), (12, 15), (3, 15), (0, 3), (0, 12))]
for pol, fm, bm in cases:
    if fm is not None:
        os.environ["TAV_FP8_FWD_MASK"], os.environ["TAV_FP8_BWD_MASK"] = str(fm), str(bm)
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
    print(f"[config 5 full depth] {pol:5s} fwd-mask {fm} bwd-mask {bm}: logits {T.rel(logits, o_logits):.2e} loss {abs(loss.item() - o_loss) / abs(o_loss):.2e} "
          f"grad-norm {abs(gn - o_gn) / o_gn:.2e}", flush=True)
    del pre, model
    torch.cuda.empty_cache()
