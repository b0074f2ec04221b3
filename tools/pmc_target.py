"""Small fixed workload for rocprofv3 --pmc passes: the dominant kernels at the benchmark's video-encoder shapes (b=8)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

dev, B, S, H, nh = "cuda", 8, 1464, 768, 12
M = B * S
a = torch.randn(M, 768, device=dev).bfloat16()
w1 = torch.randn(3072, 768, device=dev).bfloat16()
bias = torch.randn(3072, device=dev)
dy = torch.randn(M, 3072, device=dev).bfloat16()
qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
x32 = torch.randn(M, H, device=dev)
g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
for _ in range(5):
    ops.gemm_nt(a, w1, bias=bias, act=1)                       # video FFN1: M=11712 N=3072 K=768 (+GELU epilogue)
    ops.gemm_tn(dy, a, want_bias=True)                         # video dW1: tokens=11712, 3072 x 768
    o, lse, _ = ops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nh)
    ops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, o, lse, None, B, S, nh)
    ops.ln_fwd(x32, g, b, 1e-12, want_f32=False, lp_dtype=torch.bfloat16)
torch.cuda.synchronize()
