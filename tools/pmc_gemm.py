"""Target for PMC passes over the GEMM kernels (tools/gpu_pmc_attn.sh <tag> tools/pmc_gemm.py gemm_): three launches of a video-encoder
layer at the benchmarked batch -- the QKV projection (K = 768, bf16 out), the FFN1 dgrad (K = 3072, f32 out) and the layer's four weight
gradients in one grouped launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

dev, B = "cuda", int(os.environ.get("TAV_B", "32"))
M, H, F = B * 1464, 768, 3072


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


x, h, wqkv, w1t, b3 = rnd(M, H), rnd(M, F), rnd(3 * H, H), rnd(H, F), torch.randn(3 * H, device=dev)
dqkv, dy1, du, dy2 = rnd(M, 3 * H), rnd(M, H), rnd(M, F), rnd(M, H)
pairs = [(dqkv, x), (dy1, x), (du, x), (dy2, h)]
for _ in range(4):
    ops.gemm_nt(x, wqkv, bias=b3, tile_m=16)
    ops.gemm_nt(h, w1t, out_dtype=torch.float32, tile_m=16)
    ops.gemm_tn_grouped(pairs, want_bias=True)
torch.cuda.synchronize()
