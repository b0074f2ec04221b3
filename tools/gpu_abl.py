"""Timing-only ablation runs of the NT GEMM (TAV_LIB = a build from tools/ab_build.sh with -DTAV_ABL_*; results are wrong)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

tag = os.path.basename(os.environ.get("TAV_LIB", "libtavhip.so"))
out = []
for (name, M, N, K) in [("square 4096", 4096, 4096, 4096), ("video ffn1", 11712, 3072, 768), ("video ffn2", 11712, 768, 3072)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
    for tm in (4, 16):
        for _ in range(5):
            ops.gemm_nt(a, b, tile_m=tm)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.gemm_nt(a, b, tile_m=tm)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        out.append(f"{name} tm{tm} {best:6.1f}")
print(f"[{tag:12s}] " + " | ".join(out))
