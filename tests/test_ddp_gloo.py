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


def _worker_manual(rank, world, port, out):
    """Graph-mode reducer (what bench.py times with N > 1): no hooks; pack_all() after the backward, collectives issued by the caller."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tav_amd.ddp import BucketedAllReduce
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.GELU(), torch.nn.Linear(128, 128), torch.nn.GELU(), torch.nn.Linear(128, 7))
    unused = torch.nn.Linear(3, 3)
    params = list(model.parameters()) + list(unused.parameters())
    red = BucketedAllReduce(params, bucket_mb=0.05)
    red.set_manual(True, bucket_mb=0.02)
    g = torch.Generator().manual_seed(100 + rank)
    errs = []
    for step in range(3):
        x = torch.randn(16, 64, generator=g)
        y = torch.randint(0, 7, (16,), generator=g)
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(model(x), y).backward()
        local = [p.grad.clone() for p in model.parameters()]
        red.finish()                                       # a no-op in manual mode
        red.pack_all()
        assert all(p.grad.data_ptr() != l.data_ptr() for p, l in zip(model.parameters(), local))     # re-pointed at bucket views
        red.reduce_packed()
        ref = []
        for gl in local:
            t = gl.clone()
            dist.all_reduce(t)
            ref.append(t / world)
        errs.append(max((p.grad - r).abs().max().item() for p, r in zip(model.parameters(), ref)))
        assert all(p.grad is None for p in unused.parameters())
    if rank == 0:
        out.put((errs, len(red.buckets)))
    dist.destroy_process_group()


def _worker_missing_grad(rank, world, port, out):
    """Hook mode when a bucketed parameter receives NO gradient on one rank in one step: every rank still issues the same
    collectives, the bucket is reduced with zeros in the missing slot, and all ranks end with identical gradients."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tav_amd.ddp import BucketedAllReduce
    torch.manual_seed(0)
    a, b = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
    params = list(a.parameters()) + list(b.parameters())
    red = BucketedAllReduce(params, bucket_mb=48.0)        # one bucket holding all four tensors
    g = torch.Generator().manual_seed(7 + rank)
    sums = []
    for step in range(3):
        x = torch.randn(4, 8, generator=g)
        for p in params:
            p.grad = None
        skip_b = step == 2 and rank == 1                    # rank 1 does not use branch b in the last step
        y = a(x).sum() if skip_b else (a(x) + b(x)).sum()
        y.backward()
        local_b = None if skip_b else b.weight.grad.clone()
        red.finish()
        t = torch.zeros_like(b.weight) if local_b is None else local_b.clone()
        dist.all_reduce(t)
        assert b.weight.grad is not None and torch.allclose(b.weight.grad, t / world, atol=1e-6)
        sums.append(float(sum(p.grad.sum() for p in params)))
    gathered = [None] * world
    dist.all_gather_object(gathered, sums)
    if rank == 0:
        out.put(gathered)
    dist.destroy_process_group()


def _spawn(target, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, *extra, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return q.get(timeout=10)


def test_manual_mode_two_ranks():
    errs, nb = _spawn(_worker_manual)
    assert max(errs) < 1e-6, errs
    assert nb >= 2


def test_missing_gradient_keeps_ranks_in_step():
    gathered = _spawn(_worker_missing_grad)
    assert gathered[0] == pytest.approx(gathered[1], rel=1e-6)
