"""rocprofv3 --pmc target: the 256x256 NT GEMM tile (hint 16) and the 128x128 one (hint 4) at 4096^3 and at the video QKV shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

for (M, N, K) in [(4096, 4096, 4096), (11712, 2304, 768)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
    for tm in (4, 16):
        for _ in range(5):
            ops.gemm_nt(a, b, tile_m=tm)
torch.cuda.synchronize()
