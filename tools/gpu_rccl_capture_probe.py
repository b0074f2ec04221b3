"""GPU box probe (run under `timeout`): can a raw RCCL all-reduce (tav_allreduce_bucket, one rank) be captured into a hipGraph on a side
stream forked from the capture's origin, and replayed?  Prints one line per stage so that a hang shows where."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tav_amd  # noqa: F401,E402
from tav_amd import _lib, runtime  # noqa: E402
from tav_amd._lib import ptr  # noqa: E402

h = _lib.lib()
ver = ctypes.c_int32()
print("rccl version rc", h.tav_comm_rccl_version(ctypes.byref(ver)), ver.value, flush=True)
uid = ctypes.create_string_buffer(128)
print("unique id rc", h.tav_comm_unique_id(uid), flush=True)
comm = ctypes.c_void_p()
print("init rc", h.tav_comm_init_rank(ctypes.byref(comm), 1, uid, 0), flush=True)
x = torch.randn(1 << 22, device="cuda")
y = torch.empty_like(x)
origin, side = torch.cuda.Stream(), torch.cuda.Stream()
# eager first (RCCL's lazy setup must not happen inside a capture)
with torch.cuda.stream(side):
    print("eager rc", h.tav_allreduce_bucket(ptr(x), x.numel() * 4, 0, comm, side.cuda_stream), flush=True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(origin):
    with runtime.capture(g, origin, branches=[side], capture_error_mode="thread_local"):
        y.copy_(x).mul_(2.0)
        runtime.stream_wait(side, origin)
        rc = h.tav_allreduce_bucket(ptr(y), y.numel() * 4, 0, comm, side.cuda_stream)
        runtime.stream_wait(origin, side)
        y.add_(1.0)
print("captured rc", rc, flush=True)
for k in range(3):
    x.fill_(float(k))
    g.replay()
    torch.cuda.synchronize()
    print("replay", k, "ok" if torch.equal(y, torch.full_like(y, 2.0 * k + 1.0)) else "WRONG", flush=True)
print("destroy rc", h.tav_comm_destroy(comm), flush=True)
print("PROBE OK", flush=True)
