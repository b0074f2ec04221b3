"""Target for the attention PMC passes (tools/gpu_pmc_attn.sh): the video-encoder attention at the benchmarked batch
(B = 32, S = 1464, 12 heads of 64, bf16), forward + backward, a few launches each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

dev, B, S, H, nh = "cuda", int(os.environ.get("TAV_B", "32")), int(os.environ.get("TAV_S", "1464")), 768, 12
qkv = torch.randn(B * S, 3 * H, device=dev).bfloat16()
do = torch.randn(B * S, H, device=dev).bfloat16()
q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
pre = os.environ.get("TAV_PRE", "1") == "1"      # tav_attn_args.q_prescaled: what the engine's layers launch
for _ in range(4):
    o, lse, _ = ops.attn_fwd(q, k, v, B, S, nh, q_prescaled=pre)
    ops.attn_bwd(q, k, v, o, do, lse, None, B, S, nh, q_prescaled=pre)
torch.cuda.synchronize()
