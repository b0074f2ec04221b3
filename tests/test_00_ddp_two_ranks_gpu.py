"""-m gpu, runs FIRST (file name): the data-parallel graphed step with TWO ranks on the one GPU of the box (ddp.GraphedStep had
only ever run with one rank).  The ranks are child processes started BEFORE this pytest process touches the GPU -- a process that has
initialised HIP must not exec another program on this pool -- so the test uses torch.cuda.device_count() (which does not initialise it)
instead of the `gpu` fixture.  Two ranks share cuda:0 over the "gloo" backend (RCCL refuses two ranks per device); see
tests/ddp_two_ranks_worker.py for what is compared."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graphed_ddp_step_two_ranks_one_gpu():
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "ddp_two_ranks_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    print(out[-3000:])
    assert r.returncode == 0, out[-3000:]
    assert out.count("] OK") == 2
