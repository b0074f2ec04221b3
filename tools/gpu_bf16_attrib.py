"""Which stack's bf16 rounding costs how much of the 1e-2 logits budget?  Full depth, full sizes, batch 2 (the oracle-checked case): the
bf16 policy with ONE stack's transformer layers at a time on the fp32 policy (TAV_F32_BRANCHES, a fresh process per case because the
switch is read at import).  usage: python tools/gpu_bf16_attrib.py [A|B] [seed]      (GPU box only)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
preset = sys.argv[1] if len(sys.argv) > 1 else "A"
seed = sys.argv[2] if len(sys.argv) > 2 else "0"
if os.environ.get("TAV_ATTRIB_CHILD") != "1":
    for br in ("", "fusion", "video", "audio", "text", "fusion,video,audio,text"):
        env = dict(os.environ, TAV_F32_BRANCHES=br, TAV_ATTRIB_CHILD="1")
        print(f"[bf16 attribution] case: fp32 stacks = {br or '(none)'}", flush=True)      # (a line a minute: silent runs are taken to be hung)
        subprocess.run([sys.executable, os.path.abspath(__file__), preset, seed], env=env, check=False)
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

torch.set_num_threads(int(os.environ.get("TAV_CPU_THREADS", "16")))      # the CPU oracle: a one-GPU box gives 16 threads, torch would take the host's count

import test_model_gpu as T  # noqa: E402
from tav_amd import runtime  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402

cfg, batch, lab, (sdp, sdm), o_logits, o_loss, o_gn, o_grads = T._oracle_full(preset, seed=int(seed))
runtime.set_precision("bf16")
pre, model = PreFormer(cfg), TAVForMAE(T.ARGS, cfg)
pre.load_state_dict(sdp)
model.load_state_dict(sdm)
pre.cuda()
model.cuda()
with torch.no_grad():
    _, _, _, logits, loss = T._run_product(pre, model, batch, lab)
print(f"[bf16 attribution, preset {preset} seed {seed}] fp32 stacks: {os.environ.get('TAV_F32_BRANCHES') or '(none)':28s} logits {T.rel(logits, o_logits):.2e} "
      f"loss {abs(loss.item() - o_loss) / abs(o_loss):.2e}", flush=True)
