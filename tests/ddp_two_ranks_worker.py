"""Worker of tests/test_00_ddp_two_ranks_gpu.py: started by torch.distributed.run with 2 ranks, BOTH on cuda:0, process group "gloo"
(RCCL refuses two ranks on one device; the collectives of the data-parallel step are eager calls between hipGraphs, so the backend is
interchangeable).  Compares ddp.GraphedStep -- segmented backward captured as a chain of hipGraphs, bucket all-reduces issued from the
reducer stream between the replays -- with the eager hook-mode step (ddp.BucketedAllReduce through TrainStep) on identical replicas:
losses, gradients as reduced, and parameters after three optimizer steps; plus that both ranks derived the same bucket plan."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import tav_amd  # noqa: F401,E402
from tav_amd import config as C  # noqa: E402
from tav_amd import engine, runtime, synthetic  # noqa: E402
from tav_amd.ddp import GraphedStep  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.train_model.tav_train import TrainStep  # noqa: E402
from tav_amd.utils.global_functions import CrossEntropyLoss  # noqa: E402


LR = 1e-5          # small steps: rounding-level differences (e.g. in the clipping norm, whose summation order follows buffer alignment) must not be amplified into different trajectories


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    cfg = C.preset("B-tiny")
    for k in ("text", "audio", "video", "fusion"):
        cfg[k]["layers"] = 4
    runtime.set_precision("bf16")
    args = dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        inp_g, lab_g = synthetic.make_batch(cfg, 2 * world, s_text=16, t_audio=8000, n_visual_true=4)
        sl = slice(2 * rank, 2 * rank + 2)                          # rank r owns rows [r * B / N, (r + 1) * B / N)
        inp = [{k: v[sl].contiguous().cuda() for k, v in d.items()} for d in inp_g]
        lab = lab_g[sl].contiguous().cuda()

        def build():
            torch.manual_seed(0)
            pre, model = PreFormer(cfg), TAVForMAE(args, cfg)
            synthetic.seeded_init_(pre, 1)
            synthetic.seeded_init_(model, 2)
            pre.cuda()
            model.cuda()
            return pre, model, TrainStep(model, pre, CrossEntropyLoss(), lr=LR, weight_decay=1e-2, clip=1.0, bucket_mb=2.0)

        def warm(st):            # one eager hook-mode step incl. the update on BOTH replicas (as bench.py's warm-up: optimizer state and chunk tables exist
            st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4)       # before the capture); both then continue from the same weights
            st.update()
            torch.cuda.synchronize()

        # (a) eager hook mode: three steps
        pre_a, model_a, st_a = build()
        assert st_a.reducer is not None and st_a.reducer.world == world
        warm(st_a)
        params_0 = [p.detach().clone() for p in st_a.params]
        losses_a, grads_a = [], []
        for step in range(3):
            loss = st_a.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4)
            grads_a.append([None if p.grad is None else p.grad.detach().clone() for p in st_a.params])
            losses_a.append(loss.item())
            st_a.update()
        torch.cuda.synchronize()
        params_a = [p.detach().clone() for p in st_a.params]
        st_a.reducer.remove()

        # (c) reference: ONE process, the full global batch, no reducer (the mean over 2N rows = the mean of the two ranks' means)
        pre_c, model_c, st_c = build()
        st_c.reducer.remove()
        st_c.reducer = None
        inp_full = [{k: v.contiguous().cuda() for k, v in d.items()} for d in inp_g]
        lab_full = lab_g.contiguous().cuda()
        losses_c = []
        for step in range(4):
            loss = st_c.forward_backward(inp_full, lab_full, check="val", epoch=0, n_visual_true=4)
            if step > 0:
                losses_c.append(loss.item())
            st_c.update()
        torch.cuda.synchronize()
        params_c = [p.detach().clone() for p in st_c.params]

        # (b) the graphed chain from the same state
        pre_b, model_b, st_b = build()
        warm(st_b)
        engine.bump_weight_epoch()
        st_b.opt.zero_grad()
        g = GraphedStep(st_b, lambda: st_b.forward_loss(inp, lab, check="val", epoch=0, n_visual_true=4), s, segments=4)
        sig = g.bucket_signature()
        sigs = [None] * world
        dist.all_gather_object(sigs, sig)
        assert all(x == sigs[0] for x in sigs), f"bucket plans differ between ranks: {sigs}"
        losses_b, grads_b = [], []
        index = {id(p): i for i, p in enumerate(st_b.params)}
        for step in range(3):
            loss = g.run()
            torch.cuda.synchronize()
            gb = [None] * len(st_b.params)                           # (read after the step: the flats still hold the reduced gradients of this step)
            for plist, flat in g.flats:
                off = 0
                for p in plist:
                    gb[index[id(p)]] = flat[off:off + p.numel()].view_as(p).clone()
                    off += p.numel()
            grads_b.append(gb)
            losses_b.append(loss.item())
        params_b = [p.detach().clone() for p in st_b.params]

    def rel(a, b):
        return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12)).item()

    assert [ga is None for ga in grads_a[0]] == [gb is None for gb in grads_b[0]], "sets of trained parameters differ"
    e_g0 = max(rel(gb, ga) for ga, gb in zip(grads_a[0], grads_b[0]) if ga is not None)        # first step: same weights, so the REDUCED gradients must agree tensor by tensor

    def l2(ts):
        return float(sum((t.double() ** 2).sum() for t in ts)) ** 0.5
    # later steps: global relative L2 distance (a tensor whose true gradient is zero -- the key bias, to which the softmax is invariant -- holds
    # rounding noise only, so a per-tensor relative figure is meaningless there)
    e_gs = [l2([gb - ga for ga, gb in zip(grads_a[k], grads_b[k]) if ga is not None]) / l2([ga for ga in grads_a[k] if ga is not None]) for k in range(3)]
    moved = l2([pa - p0 for pa, p0 in zip(params_a, params_0)])
    e_p = l2([pb - pa for pa, pb in zip(params_a, params_b)]) / moved                              # distance between the two end points / distance travelled
    e_l = max(abs(a - b) / abs(a) for a, b in zip(losses_a, losses_b))
    print(f"[rank {rank}] losses eager {losses_a} graphed {losses_b}; gradient distance per step {e_gs}", flush=True)
    d_ac = l2([pc - pa for pa, pc in zip(params_a, params_c)]) / moved
    d_bc = l2([pc - pb for pb, pc in zip(params_b, params_c)]) / moved
    print(f"[rank {rank}] single-process full-batch reference: global-batch losses {losses_c}; end-point distance eager-vs-reference {d_ac:.2e}, graphed-vs-reference {d_bc:.2e}", flush=True)
    # both paths must also agree ACROSS ranks (same reduced gradients everywhere)
    chk = torch.tensor([float(sum(p.double().sum() for p in params_b))], dtype=torch.float64)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    print(f"[rank {rank}] two-rank graphed step vs eager hook mode: reduced gradients of the first step {e_g0:.2e} (per tensor), later steps {max(e_gs):.2e} (global L2), end points {e_p:.2e} of the distance travelled, losses {e_l:.2e}; "
          f"buckets {sig}; {g.describe()}", flush=True)
    # Same weights -> the reduced gradients agree tensor by tensor.  After an update the two paths differ at rounding level (the clipping norm is
    # summed in buffer order, the buckets are laid out differently) and AdamW's normalisation amplifies that on parameters whose gradient is
    # noise; the yardstick is therefore the single-process reference: the graphed chain must end as close to it as the eager path does.
    assert e_g0 < 1e-6, e_g0
    assert d_bc < 2 * d_ac + 2e-3 and e_p < 2 * d_ac + 2e-3, (d_ac, d_bc, e_p)
    assert max(e_gs) < 5e-2 and e_l < 2e-2, (e_gs, e_l)
    assert all(abs(c.item() - allc[0].item()) <= 1e-9 * abs(allc[0].item()) for c in allc), "replicas diverged"
    dist.destroy_process_group()
    print(f"[rank {rank}] OK", flush=True)


if __name__ == "__main__":
    main()
