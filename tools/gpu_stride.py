"""NT GEMM time vs operand row stride (L2 channel mapping experiment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import tav_amd.ops as ops

def t(fn):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    return best

for (M, N, K) in [(4096, 4096, 4096), (11712, 3072, 768), (11712, 768, 3072), (11712, 2304, 768)]:
    row = []
    for pad in (0, 64, 128, 192, 320):
        a = torch.randn(M, K + pad, device="cuda").bfloat16()[:, :K]
        b = torch.randn(N, K + pad, device="cuda").bfloat16()[:, :K]
        row.append(f"pad{pad} {t(lambda: ops.gemm_nt(a, b)):6.1f}")
    print(f"M={M} N={N} K={K}: " + "  ".join(row))
