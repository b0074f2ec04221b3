"""Config 5 at full depth (24 video layers), batch 2: would MX block scales (e4m3 elements, one power-of-two scale per 32 k-values: the native
operand format of v_mfma_scale_f32_16x16x128_f8f6f4) bring the out-proj / FFN forward GEMMs inside the 1e-2 parity budget that per-tensor e4m3
misses (profiles/r03_fp8_attribution.txt: 1.5e-2 .. 2.2e-2 each)?  The operands are rounded to MX-e4m3 by quantise / dequantise in torch
(engine._mx_roundtrip) and fed to the bf16 GEMM -- numerically what a block-scaled kernel would compute, without building it.  GPU box only."""
import os
import sys

os.environ["TAV_FP8_MX_EMULATE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

torch.set_num_threads(16)
import test_model_gpu as T  # noqa: E402
from tav_amd import config as C  # noqa: E402
from tav_amd import runtime  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.optim import grad_norm  # noqa: E402

cfg = C.preset("B5")
rec = T._oracle_full("B5", seed=0, batch_size=2, cfg=cfg, tag="B5-full")
cfg, batch, lab, (sdp, sdm), o_logits, o_loss, o_gn, o_grads = rec
for label, fm in (("QKV only (per-tensor e4m3, the shipped policy)", 1), ("+ out-proj MX", 3), ("+ FFN1 MX", 5), ("+ FFN2 MX", 9), ("+ all three MX", 15)):
    os.environ["TAV_FP8_FWD_MASK"], os.environ["TAV_FP8_BWD_MASK"] = str(fm), "15"
    runtime.set_precision("fp8")
    pre, model = PreFormer(cfg), TAVForMAE(T.ARGS, cfg)
    pre.load_state_dict(sdp)
    model.load_state_dict(sdm)
    pre.cuda()
    model.cuda()
    _, _, _, logits, loss = T._run_product(pre, model, batch, lab)
    loss.backward()
    torch.cuda.synchronize()
    gn = grad_norm(list(pre.parameters()) + list(model.parameters())).item()
    print(f"[config 5 full depth, MX-e4m3 emulation] {label:48s}: logits {T.rel(logits, o_logits):.2e} loss {abs(loss.item() - o_loss) / abs(o_loss):.2e} "
          f"grad-norm {abs(gn - o_gn) / o_gn:.2e}", flush=True)
    del pre, model
    torch.cuda.empty_cache()
