"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce of ddp.py (the N>1 path of bench.py / TrainStep).
Checks: gradients end up as the mean over ranks, bucket planning follows the observed ready order, the second step
(overlapped launches from the hooks) gives the same result as the first (everything reduced in finish())."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tav_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_mb, reduce_dtype, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tav_amd.ddp import BucketedAllReduce
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.GELU(), torch.nn.Linear(128, 128), torch.nn.GELU(), torch.nn.Linear(128, 7))
    unused = torch.nn.Linear(3, 3)                        # like PreFormer's unused encoder layers: never gets a gradient
    params = list(model.parameters()) + list(unused.parameters())
    red = BucketedAllReduce(params, bucket_mb=bucket_mb, reduce_dtype=reduce_dtype)
    g = torch.Generator().manual_seed(100 + rank)
    results = []
    for step in range(3):
        x = torch.randn(16, 64, generator=g)
        y = torch.randint(0, 7, (16,), generator=g)
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        local = [p.grad.clone() for p in model.parameters()]
        red.finish()
        # reference: plain all_reduce of the local grads
        ref = []
        for gl in local:
            t = gl.clone()
            dist.all_reduce(t)
            ref.append(t / world)
        err = max((p.grad - r).abs().max().item() for p, r in zip(model.parameters(), ref))
        results.append(err)
        assert all(p.grad is None for p in unused.parameters())
    nb = len(red.buckets)
    order_ok = red.buckets[0][0][0] is list(model.parameters())[-1] or red.buckets[0][0][0] is list(model.parameters())[-2]   # last layer first
    if rank == 0:
        out.put((results, nb, order_ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,reduce_dtype,tol", [(0.05, None, 1e-6), (48.0, None, 1e-6), (0.05, torch.bfloat16, 2e-2)])
def test_bucketed_allreduce_two_ranks(bucket_mb, reduce_dtype, tol):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, reduce_dtype, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results, nb, order_ok = q.get(timeout=10)
    assert max(results) < tol, results
    assert order_ok
    assert nb >= (2 if bucket_mb < 1 else 1)
