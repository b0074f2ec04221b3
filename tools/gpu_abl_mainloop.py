import os, sys
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo")]
import torch
import tav_amd.ops as ops
def timeit(fn, iters=30, reps=5):
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return min(ts)
tag = os.path.basename(os.environ.get("TAV_LIB", "libtavhip.so"))
for (name, M, N, K) in [("qkv", 46848, 2304, 768), ("dffn1-like", 46848, 768, 3072), ("4096^3", 4096, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    for tm in (16, 4):
        t = timeit(lambda: ops.gemm_nt(a, b, tile_m=tm))
        print(f"[{tag}] {name:11s} tm{tm:2d}: {t:8.1f} us  {2*M*N*K/t/1e6:7.1f} TF", flush=True)
