"""Per-kernel timings at the bench shapes (GPU box only): TFLOP/s of the MFMA kernels, GB/s of the HBM-bound ones."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s, dtype=torch.bfloat16):
    return torch.randn(*s, device=dev).to(dtype)


B = int(os.environ.get("TAV_B", "32"))
for dtype in (torch.bfloat16, torch.float32):
    print(f"== {dtype}  (batch {B})")
    for (name, M, N, K) in [("video qkv", B * 1464, 2304, 768), ("video out", B * 1464, 768, 768), ("video ffn1", B * 1464, 3072, 768),
                            ("video ffn2", B * 1464, 768, 3072), ("fusion qkv", B * 481, 2304, 768), ("text ffn1", B * 128, 3072, 768),
                            ("audio ffn1", B * 249, 3072, 768), ("conv1", B * 7999, 512, 1536), ("square 4096", 4096, 4096, 4096)]:
        a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype)
        bias = torch.randn(N, device=dev)
        t = timeit(lambda: ops.gemm_nt(a, b, bias=bias))
        extra = ""
        if dtype == torch.bfloat16:
            extra = "   [tile_m 2/3/4: " + " ".join(f"{timeit(lambda tm=tm: ops.gemm_nt(a, b, bias=bias, tile_m=tm))*1e6:.1f}" for tm in (2, 3, 4)) + " us]"
        print(f"gemm_nt {name:12s} M={M:6d} N={N:5d} K={K:5d}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TFLOP/s{extra}")
    for (name, M, N1, N2) in [("video dWqkv", B * 1464, 2304, 768), ("video dWo", B * 1464, 768, 768), ("video dW1", B * 1464, 3072, 768),
                              ("video dW2", B * 1464, 768, 3072), ("text dW1", B * 128, 3072, 768)]:
        a, b = rnd(M, N1, dtype=dtype), rnd(M, N2, dtype=dtype)
        t = timeit(lambda: ops.gemm_tn(a, b))
        print(f"gemm_tn {name:12s} M={M:6d} N1={N1:5d} N2={N2:5d}: {t*1e6:9.1f} us  {2*M*N1*N2/t/1e12:8.1f} TFLOP/s")
    for (name, S, mode) in [("video", 1464, 0), ("fusion", 481, 2), ("audio", 249, 0), ("text", 128, 1)]:
        nh, H = 12, 768
        qkv = rnd(B * S, 3 * H, dtype=dtype)
        mask = torch.zeros(B, S, device=dev) if mode else None
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        t = timeit(lambda: ops.attn_fwd(q, k, v, B, S, nh, key_mask=mask, mask_mode=mode))
        fl = 4 * B * nh * S * S * 64
        print(f"attn_fwd {name:8s} S={S:5d}: {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TFLOP/s")
        o, lse, aux = ops.attn_fwd(q, k, v, B, S, nh, key_mask=mask, mask_mode=mode)
        do = rnd(B * S, H, dtype=dtype)
        t = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, aux if mode == 2 else None, B, S, nh, key_mask=mask, mask_mode=mode))
        print(f"attn_bwd {name:8s} S={S:5d}: {t*1e6:9.1f} us  {2.5*fl/t/1e12:8.1f} TFLOP/s (algorithmic 10*B*h*S^2*d)")
    rows = B * 1464
    x = torch.randn(rows, 768, device=dev)
    g, bb = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    lp = torch.bfloat16 if dtype == torch.bfloat16 else None
    t = timeit(lambda: ops.ln_fwd(x, g, bb, 1e-12, want_f32=(lp is None), lp_dtype=lp))
    byt = rows * 768 * (4 + (2 if lp else 4))
    print(f"ln_fwd rows={rows}: {t*1e6:9.1f} us  {byt/t/1e9:8.1f} GB/s")
    _, _, mean, rstd = ops.ln_fwd(x, g, bb, 1e-12, want_f32=True)
    dy = torch.randn(rows, 768, device=dev)
    t = timeit(lambda: ops.ln_bwd(dy, x, g, bb, mean, rstd, dx_add=dy, want_f32=True, lp_dtype=lp))
    byt = rows * 768 * (4 * 4 + (2 if lp else 0))
    print(f"ln_bwd rows={rows}: {t*1e6:9.1f} us  {byt/t/1e9:8.1f} GB/s")
