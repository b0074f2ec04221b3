"""Benchmark of the TAV hot path on MI355X: utterances/s of the full training step
(PreFormer.forward -> TAVForMAE.forward(check="val") -> CE -> backward [-> gradient all-reduce] -> clip_grad_norm_ -> AdamW)
on synthetic MELD-shaped batches (text 128 tokens, audio 16 kHz x 5 s, video 16x3x224x224, 104 fusion / 1464 encoder video tokens),
preset B (bert-base + wav2vec2-base + videomae-base), bf16 operands / f32 accumulate, batch 8 per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (gemm_nt, bf16 MFMA): algorithmic FLOPs / launch time measured live with HIP events
                  on the launch stream in a separate instrumented pass (the timed region itself is not instrumented);
  cpu_baseline -- the CPU oracle (oracle/tav_oracle.py, fp32 PyTorch restatement of the reference path, kind "port") timed
                  on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import tav_amd  # noqa: E402,F401
from tav_amd import config as C  # noqa: E402
from tav_amd import ops, runtime, synthetic  # noqa: E402
from tav_amd.models.tav import PreFormer, TAVForMAE  # noqa: E402
from tav_amd.train_model.tav_train import TrainStep  # noqa: E402
from tav_amd.utils.global_functions import CrossEntropyLoss  # noqa: E402

MFMA_PEAK_BF16_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic forward FLOPs per utterance, BASELINE.md §2 (fwd+bwd = 3x)
FWD_GFLOP_PER_UTT = {"B": 547.0, "A": 634.8}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(cfg, pre, model, sample_b, iters):
    """Oracle (CPU port of the reference path) fwd + loss + bwd, utterances/s on the host cores of this box."""
    from oracle import tav_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    (tx, au, vi), lab = synthetic.make_batch(cfg, sample_b, seed=4321)
    batch = dict(input_ids=tx["input_ids"], text_mask=tx["attention_mask"], audio_features=au["audio_features"], audio_mask=au["attention_mask"],
                 video_embeds=vi["visual_embeds"], visual_mask=vi["attention_mask"])
    sd_pre = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
    sd_model = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        _, loss = O.tav_step(sd_model, sd_pre, cfg, batch, lab.long())
        loss.backward()
        times.append(time.perf_counter() - t0)
        log(f"cpu baseline iter {it}: {times[-1]:.2f} s (batch {sample_b}, {cores} threads)")
        for v in list(sd_pre.values()) + list(sd_model.values()):
            v.grad = None
        if it >= 1 and sum(times) > 40.0:          # bounded sample
            break
    iters = len(times) - 1
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(sample_b / t, 4), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 (PyTorch CPU) fwd+loss+bwd, same preset and input shapes, batch {sample_b}, median of {iters} after 1 warm-up"}


def host_cores():
    """CPU threads this process may really use: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    # a one-GPU box of this pool shares its host: 16 CPU threads per GPU (task environment notes)
    return int(os.environ.get("TAV_CPU_THREADS", min(n, 16)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--preset", default="B")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+loss+bwd(+all-reduce) only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--bucket-mb", type=float, default=48.0)
    ap.add_argument("--reduce-bf16", action="store_true", help="all-reduce gradients in bf16 (halves xGMI bytes)")
    ap.add_argument("--graph", type=int, default=1, help="1: capture the whole step into a hipGraph and replay it (single-GPU runs); 0: eager launches")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtavhip has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    alone_ddp = world == 1 and os.environ.get("TAV_DDP_SINGLE_RANK", "0") == "1"      # exercise the data-parallel step with one rank
    if alone_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)
    if args.gpus != world:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")

    # every step (warm-up, capture, replay) runs on ONE non-default stream: autograd ties each parameter's AccumulateGrad node to
    # the stream of its first backward, and a hipGraph cannot be captured across the legacy default stream
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    cfg = C.preset(args.preset)
    runtime.set_precision(args.dtype)
    b = args.batch_per_gpu
    torch.manual_seed(0)
    pre = PreFormer(cfg)
    model = TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)       # identical replicas on every rank
    torch.set_num_threads(min(host_cores(), 16))
    log(f"[rank {rank}] models built ({sum(p.numel() for p in model.parameters()) / 1e6:.0f} M + {sum(p.numel() for p in pre.parameters()) / 1e6:.0f} M params)")
    pre.to(dev)
    model.to(dev)
    inp, labels = synthetic.make_batch(cfg, b, seed=1234 + rank, device=dev)      # resident in HBM before timing
    n_true = 104 if cfg["video"]["image"] == 224 else 4
    stepper = TrainStep(model, pre, CrossEntropyLoss(), lr=1e-6, weight_decay=1e-4, clip=1.0, bucket_mb=args.bucket_mb,
                        reduce_dtype=torch.bfloat16 if args.reduce_bf16 else None)

    def one_step():
        loss = stepper.forward_backward(inp, labels, check="val", epoch=0, n_visual_true=n_true)
        if args.no_optimizer:
            stepper.opt.zero_grad()
        else:
            stepper.update()
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = one_step()
        torch.cuda.synchronize()
        log(f"[rank {rank}] warm-up step {i} done")

    # hipGraph: capture one whole step (fwd, loss, bwd, clip, AdamW, weight re-casts) and replay it -- ~3000 kernel launches per
    # step would otherwise cost the Python host about as long as the GPU needs to run them.
    graph, eager_step = None, one_step
    if args.graph and stepper.reducer is not None and not args.no_optimizer:
        # Data parallel: two graphs per step with the gradient all-reduce issued eagerly between them, so that no RCCL call is ever
        # captured: G1 = forward + loss + backward + pack gradients into their buckets, then one all-reduce (mean) per bucket on the
        # work stream, then G2 = clip_grad_norm_ + AdamW reading the reduced buckets.  The host only enqueues ~35 collectives per
        # step (the eager hook-based path -- all-reduce overlapped with the backward on a side stream -- is host bound at ~44 ms/step).
        g1 = g2 = static_loss = None
        try:
            from tav_amd import engine
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            engine.bump_weight_epoch()
            stepper.opt.zero_grad()
            stepper.reducer.set_manual(True, bucket_mb=256.0)
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, stream=work_stream):
                static_loss = stepper.forward_backward(inp, labels, check="val", epoch=0, n_visual_true=n_true)
                stepper.reducer.pack_all()
            with torch.cuda.graph(g2, stream=work_stream, pool=g1.pool()):
                stepper.update()
            captured = True
        except Exception as e:
            import traceback
            log(f"[rank {rank}] graph capture failed ({type(e).__name__})\n" + "".join(traceback.format_exc().splitlines(True)[-14:]))
            captured = False
            torch.cuda.synchronize()
        # every rank must issue the same sequence of collectives: agree BEFORE the first replay (no collective was issued since the barrier)
        ok = torch.tensor([1.0 if captured else 0.0], device=dev)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if ok.item() > 0.5:
            graph = (g1, g2)

            def one_step():                                  # noqa: F811
                g1.replay()
                stepper.reducer.reduce_packed()
                g2.replay()
                return static_loss
            for _ in range(2):
                one_step()
            torch.cuda.synchronize()
            log(f"[rank {rank}] step captured into two hipGraphs around {len(stepper.reducer.buckets)} eager all-reduces")
        else:
            log(f"[rank {rank}] falling back to eager launches (hook-mode all-reduce) on all ranks")
            stepper.opt.zero_grad()
            stepper.reducer.set_manual(False)
            graph, one_step = None, eager_step
    elif args.graph and world == 1:
        try:
            from tav_amd import engine
            torch.cuda.synchronize()
            engine.bump_weight_epoch()                       # the operand casts must be part of the captured step
            stepper.opt.zero_grad()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=work_stream):
                static_loss = one_step()

            def one_step():                                  # noqa: F811
                graph.replay()
                return static_loss
            for _ in range(2):
                one_step()
            torch.cuda.synchronize()
            log(f"[rank {rank}] step captured into a hipGraph")
        except Exception as e:                               # keep the benchmark alive: fall back to eager launches
            import traceback
            log(f"[rank {rank}] graph capture failed ({type(e).__name__}); falling back to eager\n" + "".join(traceback.format_exc().splitlines(True)[-14:]))
            graph, one_step = None, eager_step
            torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    host_enqueue = time.perf_counter() - t0          # host time to enqueue K steps (no sync inside the loop)
    fence()
    elapsed = time.perf_counter() - t0
    log(f"[rank {rank}] host enqueue {host_enqueue / args.steps * 1e3:.2f} ms/step vs wall {elapsed / args.steps * 1e3:.2f} ms/step")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()
    log(f"[rank {rank}] timed region: {elapsed / args.steps * 1e3:.2f} ms/step, loss {final_loss:.5f}")

    roof = None
    if not args.no_roofline and args.dtype == "bf16":
        # (every rank runs it: in hook mode these steps contain the gradient all-reduces, which must be matched by all ranks)
        # Instrumented pass: eager launches (a HIP event pair per GEMM), all branches on ONE stream so that every launch has the
        # device to itself -- the same condition rocprofv3's kernel trace measures (it serialises dispatches), which is what
        # profiles/*kernel_stats* must agree with.  In the timed region above the four branches overlap.
        runtime.multistream[0] = False
        ops.profile_start("gemm_nt")
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        flops, secs, launches = ops.profile_stop()
        runtime.multistream[0] = True
        ach = flops / max(secs, 1e-9) / 1e12
    if rank == 0 and not args.no_roofline and args.dtype == "bf16":
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure is the committed rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE summary of this same command (profiles/r01_pmc/traffic_per_launch.json: 2 x FETCH + WRITE, averaged
        # over every gemm_nt launch), next to the algorithmic minimum (operands + output once) counted live
        traffic = None
        try:
            if not (b == 8 and args.preset == "B"):
                raise LookupError("the committed PMC summary was taken on the default workload only")
            with open(os.path.join(ROOT, "profiles", "r01_pmc", "traffic_per_launch.json")) as f:
                traffic = json.load(f)["families"]["gemm_nt_kernel"]["hbm_bytes_per_launch_corrected"]
        except Exception:
            pass
        roof = {"kernel": "tav::gemm_nt_kernel<bf16,*>", "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(getattr(ops.profile_stop, "algorithmic_bytes", 0.0) / max(launches, 1)), "launches_per_step": launches // 2,
                "avg_launch_us": round(secs / launches * 1e6, 2), "serial_ms_per_step": round(secs / 2 * 1e3, 3)}

    cpu_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t0 = time.time()
        cpu_ref = cpu_baseline(cfg, pre, model, args.cpu_sample_batch, args.cpu_iters)
        log(f"cpu baseline: {cpu_ref}  ({time.time() - t0:.1f} s)")

    if rank == 0:
        utt = world * b * args.steps
        value = utt / elapsed
        base = args.preset.split("-")[0]
        step_flops = 3 * FWD_GFLOP_PER_UTT.get(base, 0.0) * 1e9 * b
        out = {
            "metric": "utterances/sec fwd+bwd, TAV (BERT+Wav2Vec2+VideoMAE) b=32, 1/2/4/8 MI355X",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"tav_nn.py TAV preset {args.preset} (bert-base + wav2vec2-base + videomae-base), batch {b} per GPU, text 128 tok, "
                                   f"audio 80000 samples, video 16x3x224x224 (104 fusion / 1464 encoder tokens)",
                       "global_batch": world * b, "per_gpu_batch": b, "parallelism": f"dp{world}",
                       "step": "PreFormer+TAVForMAE fwd, CE, bwd" + (", grad all-reduce (RCCL)" if world > 1 else "")
                               + ("" if args.no_optimizer else ", clip_grad_norm_, AdamW"),
                       "weights": "random init (seeded), no checkpoints offline", "final_loss": round(final_loss, 5),
                       "launch": ("two hipGraphs + eager RCCL all-reduce" if isinstance(graph, tuple) else "hipGraph replay") if graph is not None else "eager",
                       "mfma_util_whole_step": round(step_flops / (elapsed / args.steps) / (MFMA_PEAK_BF16_TFLOPS * 1e12), 4)},
        }
        if roof is not None:
            out["roofline"] = roof
        if cpu_ref is not None:
            out["cpu_baseline"] = cpu_ref
        print(json.dumps(out), flush=True)
    if world > 1 or alone_ddp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
