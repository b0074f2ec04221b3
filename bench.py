"""Benchmark of the TAV hot path on MI355X: utterances/s of the training step
(PreFormer.forward -> TAVForMAE.forward(check="val") -> CE -> backward [-> gradient all-reduce] -> clip_grad_norm_ -> AdamW)
on synthetic MELD-shaped batches (text 128 tokens, audio 16 kHz x 5 s, video 16x3x224x224, 104 fusion / 1464 encoder video tokens),
preset B (bert-base + wav2vec2-base + videomae-base), bf16 operands / f32 accumulate.

The workload is BASELINE.json's headline: GLOBAL batch 32.  One GPU runs all 32 utterances; N GPUs shard them as 32/N contiguous rows
per rank, replicated weights, one gradient mean per step over RCCL -- total work is fixed, so "scaling": "strong".
`--batch-per-gpu B` switches to the weak variant (B utterances on every GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 ...          # starts the 8 ranks itself (python -m torch.distributed.run, 127.0.0.1) before touching a GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (gemm_nt, bf16 MFMA): algorithmic FLOPs / launch time measured live with HIP events
                  on the launch stream in a separate instrumented pass (the timed region itself is not instrumented);
                  `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this same
                  command, used only while the kernel sources and the workload still match what was profiled (else null);
  cpu_baseline -- the CPU oracle (oracle/tav_oracle.py, fp32 PyTorch restatement of the reference path, kind "port") timed
                  on this box's host cores on a bounded sample (rank 0, N = 1 only).
A failed hipGraph capture is fatal (exit code 3): an eager fallback would silently time a different launch mode.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_FP8_TFLOPS = 5000.0         # dense fp8 (block-scaled MFMA), same guide


def fwd_gflop_per_utt(cfg, s_text=128, t_audio=80000):
    """Algorithmic forward FLOPs per utterance, SURVEY.md §8(d) / BASELINE.md §2: per transformer layer 8 S H^2 + 4 S H F + 4 S^2 H; the wav2vec2
    conv stack, feature projection and positional conv; the tubelet patch embedding over all tokens (as the reference computes it); fwd+bwd = 3x.
    Reproduces the table there (preset B 547.0, preset A with 70 text tokens 634.8) and extends it to config 5."""
    def stack(L, S, H, F):
        return L * (8 * S * H * H + 4 * S * H * F + 4 * S * S * H)
    a, v, t, f = cfg["audio"], cfg["video"], cfg["text"], cfg["fusion"]
    lens, T, cin, conv = [], t_audio, 1, 0
    for cd, k, st in zip(a["conv_dim"], a["conv_kernel"], a["conv_stride"]):
        T = (T - k) // st + 1
        conv += 2 * T * cin * k * cd
        cin = cd
    sa, ha = T, a["hidden"]
    front = conv + 2 * sa * a["conv_dim"][-1] * ha + 2 * sa * ha * (ha // a["pos_groups"]) * a["pos_k"]
    ntok = (v["image"] // v["patch"]) ** 2 * (v["frames"] // v["tubelet"])
    n_fus = ntok // 15                                     # 1568 -> 104, 3136 -> 209
    patch = 2 * ntok * v["hidden"] * 3 * v["tubelet"] * v["patch"] ** 2
    bridge = lambda n: (2 * n * v["hidden"] * 768 if v["hidden"] != 768 else 0)      # noqa: E731
    pre = front + 2 * sa * ha * 768 + patch + bridge(n_fus)
    s_f = s_text + sa + n_fus
    model = (stack(t["layers"], s_text, t["hidden"], t["inter"]) + front + stack(a["layers"], sa, ha, a["inter"]) + 2 * sa * ha * 768
             + patch + stack(v["layers"], ntok - n_fus, v["hidden"], v["inter"]) + bridge(ntok - n_fus) + stack(f["layers"], s_f, f["hidden"], f["inter"]))
    return (pre + model) / 1e9, n_fus

GLOBAL_BATCH = 32                      # BASELINE.json metric: "... b=32, 1/2/4/8 MI355X"
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r04_pmc", "traffic_per_launch.json")
HBM_PEAK_GBS = 8000.0                 # HBM3E, same guide (about 6300 GB/s is what a streaming copy reaches)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_hash(files=("gemm.hip", "common.h")):
    """Identity of the code a PMC traffic figure was taken on: the kernel sources of that family (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "multi-modal-emotion_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


FAMILY_SOURCES = {"attention": ("attention.hip", "common.h"), "layernorm": ("norm.hip", "common.h"), "gemm_tn": ("gemm.hip", "common.h")}
FAMILY_KERNELS = {"attention": ("attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkdv_kernel"), "layernorm": ("ln_fwd_kernel", "ln_bwd_kernel"),
                  "gemm_tn": ("gemm_tn_grouped_big_kernel", "gemm_tn_grouped_kernel", "gemm_tn_kernel")}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH, help="utterances per step over ALL GPUs (strong scaling: 32/N per rank)")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="> 0: weak-scaling variant, this many utterances on every GPU")
    ap.add_argument("--preset", default="B")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8", "fp8-all", "fp8-wgrad8"],
                    help="fp8: e4m3 operands (per-tensor scales) in the linear layers of every transformer block, the rest as bf16 (BASELINE config 5)")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+loss+bwd(+all-reduce) only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (fwd+bwd only, batch 8, preset A)")
    ap.add_argument("--cpu-protocol", default="bounded", choices=["bounded", "full"],
                    help="bounded: b=1, 2 warm-ups, median of 5 (~25 s); full: BASELINE.md §3, b in {1, 8} (several minutes)")
    ap.add_argument("--bucket-mb", type=float, default=48.0)
    ap.add_argument("--reduce-bf16", action="store_true", help="all-reduce gradients in bf16 (halves xGMI bytes)")
    ap.add_argument("--reduce-bf16-tail", action="store_true", help="N > 1: only the LAST bucket (lowest segment: front-ends, embedding tables -- the one "
                    "all-reduce nothing is left to hide behind) crosses xGMI in bf16; every other gradient stays f32 on the wire")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1 (or TAV_DDP_SINGLE_RANK=1), chain mode: bucket ranges are reduced to their OWNER rank only, each rank clips and runs AdamW on the "
                         "1/N it owns and the updated parameters come back through the buckets (optim.ShardedAdamW; parameters bit-equal to the replicated "
                         "optimizer's -- tests).  Off by default: it has run with two ranks over gloo only, never on RCCL with N > 1")
    ap.add_argument("--graph", type=int, default=1, help="1: capture the step into hipGraphs and replay them; 0: eager launches (debugging)")
    ap.add_argument("--profile-serial", action="store_true",
                    help="profiling mode: eager launches, every branch on ONE stream, so that a rocprofv3 --kernel-trace --stats of this command sees each "
                         "kernel alone on the device (the condition of the roofline pass); the utterances/s of such a run is not the headline")
    ap.add_argument("--branch-streams", type=int, default=-1,
                    help="1: text / audio / video encoders on their own HIP streams beside the fusion stack, 0: everything on one stream, -1: library default")
    ap.add_argument("--ddp-segments", type=int, default=4, help="N > 1: backward segments per step; bucket i is reduced on a side stream while segment i+1 runs")
    ap.add_argument("--ddp-mode", default=os.environ.get("TAV_DDP_MODE", "auto"), choices=["auto", "single", "chain"],
                    help="N > 1: 'single' = the whole step as ONE hipGraph with the bucket all-reduces captured as raw RCCL calls on the reducer branch "
                         "(C ABI tav_allreduce_bucket); 'chain' = one graph per backward segment, torch.distributed all-reduces issued eagerly between "
                         "them (rounds 2-3); 'auto' = chain: the single graph is 2 % faster with one rank on RCCL (18.5 vs 19.0 ms at 4 utterances per "
                         "GPU, profiles/r04_ab_ddp.txt) but a captured collective has never met a second rank on this project, and a hang at N = 8 "
                         "would cost the whole scaling measurement -- so the path that has run with two ranks stays the default")
    args = ap.parse_args()
    if args.shard_optimizer and args.ddp_mode == "single":
        ap.error("--shard-optimizer runs as the graph chain (its exchanges are torch.distributed calls between the graphs): not with --ddp-mode single")
    return args


def self_launch(args):
    """`python bench.py --gpus N` run plainly: start the N ranks as a child torchrun BEFORE this process touches a GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"bench.py: launching {args.gpus} ranks: {' '.join(cmd)}")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def describe(cfg):
    t, a, v = cfg["text"], cfg["audio"], cfg["video"]
    return (f"{t['kind']} {t['layers']} L + wav2vec2 {a['layers']} L / {a['hidden']} + videomae {v['layers']} L / {v['hidden']} on {v['frames']} frames "
            f"+ fusion {cfg['fusion']['layers']} L")


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(torch, synthetic, cfg, pre, model, protocol, gpu_loss=None):
    """Oracle (CPU port of the reference path) fwd + loss + bwd on the host cores of this box, BASELINE.md §3: 2 warm-ups, median of 5.
    `bounded` times batch 1 only (about 25 s of CPU work: the default bench run must finish in minutes); `full` adds batch 8.
    The oracle's loss on the batch-1 sample is also compared with the product's on the same weights and inputs (`gpu_loss`, a callable):
    the timed CPU work doubles as a parity check of the benchmarked build."""
    from oracle import tav_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd_pre = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
    sd_model = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    res = {}
    for b in ((1,) if protocol == "bounded" else (1, 8)):
        (tx, au, vi), lab = synthetic.make_batch(cfg, b, seed=4321)
        batch = dict(input_ids=tx["input_ids"], text_mask=tx["attention_mask"], audio_features=au["audio_features"], audio_mask=au["attention_mask"],
                     video_embeds=vi["visual_embeds"], visual_mask=vi["attention_mask"])
        times = []
        if b == 1 and gpu_loss is not None:
            check = (gpu_loss(batch, lab), None)
        for it in range(7):
            t0 = time.perf_counter()
            _, loss = O.tav_step(sd_model, sd_pre, cfg, batch, lab.long())
            loss.backward()
            times.append(time.perf_counter() - t0)
            if b == 1 and gpu_loss is not None and check[1] is None:
                check = (check[0], float(loss.detach()))
            log(f"cpu baseline b={b} iter {it}: {times[-1]:.2f} s ({cores} threads)")
            for v in list(sd_pre.values()) + list(sd_model.values()):
                v.grad = None
        res[b] = b / sorted(times[2:])[len(times[2:]) // 2]
    out = {"value": round(res[max(res)], 4), "unit": "utterances/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
           "sample": f"oracle fp32 (PyTorch CPU, {cores} threads) fwd+loss+bwd on the same preset and input shapes, batch {max(res)}, "
                     "median of 5 after 2 warm-ups (BASELINE.md §3)"}
    if len(res) > 1:
        out["by_batch"] = {str(b): round(v, 4) for b, v in res.items()}
    if gpu_loss is not None:
        out["loss_check"] = {"oracle": round(check[1], 6), "libtavhip": round(check[0], 6), "rel_diff": round(abs(check[0] - check[1]) / abs(check[1]), 6),
                             "note": "same weights, same batch-1 sample; budget 1e-2 (bf16 policy)"}
    return out


def main():
    args = parse_args()
    if args.profile_serial:
        args.graph, args.no_secondary, args.no_cpu_baseline = 0, True, True
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}: refusing to report a {world}-GPU figure as {args.gpus} GPUs")
        sys.exit(2)

    import torch
    import tav_amd  # noqa: F401
    from tav_amd import config as C
    from tav_amd import engine, ops, runtime, synthetic
    from tav_amd.models.tav import PreFormer, TAVForMAE
    from tav_amd.train_model.tav_train import TrainStep
    from tav_amd.utils.global_functions import CrossEntropyLoss

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtavhip has no CPU path")
    # TAV_BENCH_REHEARSE=1: rehearsal of the N > 1 path on a ONE-GPU box -- every rank on cuda:0, process group "gloo" (RCCL refuses two ranks per
    # device; the step's collectives are eager calls between hipGraphs, so the backend is interchangeable).  Its numbers mean nothing.
    rehearse = os.environ.get("TAV_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    alone_ddp = world == 1 and os.environ.get("TAV_DDP_SINGLE_RANK", "0") == "1"      # exercise the data-parallel step with one rank
    if alone_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        if rehearse:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)

    weak = args.batch_per_gpu > 0
    if weak:
        b, gb = args.batch_per_gpu, args.batch_per_gpu * world
    else:
        gb = args.global_batch
        if gb % world:
            raise SystemExit(f"global batch {gb} does not divide over {world} ranks")
        b = gb // world

    # every step (warm-up, capture, replay) runs on ONE non-default stream: autograd ties each parameter's AccumulateGrad node to
    # the stream of its first backward, and a hipGraph cannot be captured across the legacy default stream
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    cfg = C.preset(args.preset)
    runtime.set_precision(args.dtype)
    if args.profile_serial:
        runtime.multistream[0] = False
    elif args.branch_streams >= 0:
        runtime.multistream[0] = bool(args.branch_streams)
    torch.manual_seed(0)
    pre = PreFormer(cfg)
    model = TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)       # identical replicas on every rank
    torch.set_num_threads(min(host_cores(), 16))
    log(f"[rank {rank}] models built ({sum(p.numel() for p in model.parameters()) / 1e6:.0f} M + {sum(p.numel() for p in pre.parameters()) / 1e6:.0f} M params)")
    pre.to(dev)
    model.to(dev)
    if weak:
        inp, labels = synthetic.make_batch(cfg, b, seed=1234 + rank, device=dev)      # resident in HBM before timing
    else:
        # strong scaling: ONE global batch (seed 1234), rank r owns its rows [r*b, (r+1)*b) -- SURVEY.md §8(e)
        inp_g, labels_g = synthetic.make_batch(cfg, gb, seed=1234)
        sl = slice(rank * b, (rank + 1) * b)
        inp = [{k: v[sl].contiguous().to(dev) for k, v in d.items()} for d in inp_g]
        labels = labels_g[sl].contiguous().to(dev)
        del inp_g, labels_g
    fwd_gf, n_true = fwd_gflop_per_utt(cfg)
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

    def capture_failed(e):
        import traceback
        log(f"[rank {rank}] hipGraph capture FAILED ({type(e).__name__}); refusing to time an eager fallback\n" + "".join(traceback.format_exc().splitlines(True)[-14:]))
        torch.cuda.synchronize()
        os._exit(3)              # every rank takes this path or hangs in the next collective: leave at once, non-zero

    for i in range(args.warmup):
        loss = one_step()
        torch.cuda.synchronize()
        log(f"[rank {rank}] warm-up step {i} done")

    # hipGraph: capture one whole step (fwd, loss, bwd, clip, AdamW, weight re-casts) and replay it -- ~3000 kernel launches per
    # step would otherwise cost the Python host about as long as the GPU needs to run them.
    graph, eager_step, launch = None, one_step, "eager launches, one stream (--profile-serial)" if args.profile_serial else "eager"
    abi_comm = None
    if args.graph and stepper.reducer is not None and not args.no_optimizer:
        # Data parallel: the step is a chain of hipGraphs with the gradient all-reduces issued eagerly between them, so that no RCCL
        # call is ever captured (ddp.GraphedStep): the backward is cut into --ddp-segments graphs at bucket boundaries and the
        # all-reduce of the gradients graph i produced runs on a side stream while graph i+1 executes; the last graph is
        # clip_grad_norm_ + AdamW on the reduced buckets.
        from tav_amd.ddp import GraphedStep
        try:
            fence()
            engine.bump_weight_epoch()
            stepper.opt.zero_grad()
            fwd = lambda: stepper.forward_loss(inp, labels, check="val", epoch=0, n_visual_true=n_true)      # noqa: E731
            gstep = None
            if args.ddp_mode == "single":
                gstep = GraphedStep(stepper, fwd, work_stream, segments=args.ddp_segments, mode="single", tail_bf16=args.reduce_bf16_tail)
            if gstep is None:
                gstep = GraphedStep(stepper, fwd, work_stream, segments=args.ddp_segments, mode="chain", tail_bf16=args.reduce_bf16_tail,
                                    shard_optimizer=args.shard_optimizer)
        except Exception as e:
            capture_failed(e)
        graph, one_step = gstep, gstep.run
        abi_comm = gstep.comm                        # (--ddp-mode single: the C ABI's own RCCL communicator, destroyed at exit)
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        launch = gstep.describe()
        log(f"[rank {rank}] {launch}")
    elif args.graph:
        try:
            torch.cuda.synchronize()
            engine.bump_weight_epoch()                       # the operand casts must be part of the captured step
            stepper.opt.zero_grad()
            graph = torch.cuda.CUDAGraph()
            with runtime.capture(graph, work_stream):
                static_loss = one_step()
        except Exception as e:
            capture_failed(e)

        def one_step():                                  # noqa: F811
            graph.replay()
            return static_loss
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        launch = "hipGraph replay"
        log(f"[rank {rank}] step captured into a hipGraph")

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        host = time.perf_counter() - t0          # host time to enqueue K steps (no sync inside the loop)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = t.item()
        return el, host, out

    elapsed, host_enqueue, loss = timed(one_step, args.steps)
    final_loss = loss.item()
    log(f"[rank {rank}] host enqueue {host_enqueue / args.steps * 1e3:.2f} ms/step vs wall {elapsed / args.steps * 1e3:.2f} ms/step")
    log(f"[rank {rank}] timed region: {elapsed / args.steps * 1e3:.2f} ms/step, loss {final_loss:.5f}")
    # A non-finite loss means the timed steps computed garbage (round 4: an fp8 overflow turned the activations into NaN, and kernels fed NaN ran
    # FASTER -- comparisons false, branches skipped, atomics skipped -- which produced a flattering, invalid number).  Never report such a run.
    bad = torch.tensor([0.0 if math.isfinite(final_loss) else 1.0], device=dev)
    if world > 1:
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
    if bad.item() > 0:
        log(f"[rank {rank}] loss {final_loss} is not finite after the timed steps: the measurement is INVALID, refusing to print a benchmark line")
        torch.cuda.synchronize()
        os._exit(4)

    # ---- secondary measurements (single GPU): the same step without clip+AdamW (the metric's "fwd+bwd"), and batch 8 (BASELINE configs[1])
    secondary = {}
    if world == 1 and not alone_ddp and args.graph and not args.no_secondary and not args.no_optimizer:
        def graphed(fn):
            stepper.opt.zero_grad()
            g = torch.cuda.CUDAGraph()
            with runtime.capture(g, work_stream):
                out = fn()
            return lambda: (g.replay(), out)[1]

        def fwd_bwd_only():
            ls = stepper.forward_backward(inp, labels, check="val", epoch=0, n_visual_true=n_true)
            stepper.opt.zero_grad()
            return ls
        try:
            torch.cuda.synchronize()
            fb = graphed(fwd_bwd_only)
            fb()
            el, _, _ = timed(fb, args.steps)
            secondary["fwd_bwd_only"] = {"utterances_per_s": round(b * args.steps / el, 2), "ms_per_step": round(el / args.steps * 1e3, 3)}
            if b > 8:
                inp8 = [{k: v[:8].contiguous() for k, v in d.items()} for d in inp]
                lab8 = labels[:8].contiguous()

                def step8():
                    ls = stepper.forward_backward(inp8, lab8, check="val", epoch=0, n_visual_true=n_true)
                    stepper.update()
                    return ls
                for _ in range(2):
                    step8()                                  # shapes change: warm the workspaces before capturing
                torch.cuda.synchronize()
                engine.bump_weight_epoch()
                s8 = graphed(step8)
                s8()
                el, _, _ = timed(s8, args.steps)
                secondary["batch_8"] = {"utterances_per_s": round(8 * args.steps / el, 2), "ms_per_step": round(el / args.steps * 1e3, 3),
                                        "note": "BASELINE.json configs[1]: same step at 8 utterances per GPU"}
            if args.preset == "B" and args.dtype == "bf16" and os.environ.get("TAV_BENCH_PRESET_A", "1") == "1":
                # the reference-faithful geometry (preset A: distilroberta 6 L + wav2vec2-large-xlsr 24 L / 1024 + videomae-base, what the
                # reference's hard-coded from_pretrained names resolve to, models/tav.py:438,455-457) beside the headline, same global batch
                cfg_a = C.preset("A")
                torch.manual_seed(0)
                pre_a, model_a = PreFormer(cfg_a), TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg_a)
                synthetic.seeded_init_(pre_a, 1)
                synthetic.seeded_init_(model_a, 2)
                pre_a.to(dev)
                model_a.to(dev)
                inp_a, lab_a = synthetic.make_batch(cfg_a, b, seed=1234, device=dev)
                gf_a, n_true_a = fwd_gflop_per_utt(cfg_a)
                step_a = TrainStep(model_a, pre_a, CrossEntropyLoss(), lr=1e-6, weight_decay=1e-4, clip=1.0)

                def stepA():
                    ls = step_a.forward_backward(inp_a, lab_a, check="val", epoch=0, n_visual_true=n_true_a)
                    step_a.update()
                    return ls
                for _ in range(2):
                    stepA()
                torch.cuda.synchronize()
                engine.bump_weight_epoch()
                step_a.opt.zero_grad()
                ga = torch.cuda.CUDAGraph()
                with runtime.capture(ga, work_stream):
                    loss_a = stepA()
                ga.replay()
                el, _, _ = timed(lambda: (ga.replay(), loss_a)[1], args.steps)
                secondary["preset_A"] = {"utterances_per_s": round(b * args.steps / el, 2), "ms_per_step": round(el / args.steps * 1e3, 3),
                                         "workload": f"preset A ({describe(cfg_a)}), global batch {b}, same input shapes, {gf_a:.1f} GFLOP forward per utterance",
                                         "mfma_util_whole_step": round(3 * gf_a * 1e9 * b / (el / args.steps) / (MFMA_PEAK_BF16_TFLOPS * 1e12), 4),
                                         "final_loss": round(float(loss_a.detach()), 5)}
                del ga, step_a, pre_a, model_a, inp_a, lab_a
                torch.cuda.empty_cache()
        except Exception as e:
            capture_failed(e)
        log(f"[rank {rank}] secondary: {secondary}")

    roof = None
    if not args.no_roofline and args.dtype != "fp32":
        # (every rank runs it: in hook mode these steps contain the gradient all-reduces, which must be matched by all ranks)
        # Instrumented pass: eager launches (a HIP event pair per GEMM), all branches on ONE stream so that every launch has the
        # device to itself -- the same condition rocprofv3's kernel trace measures (it serialises dispatches), which is what
        # profiles/*kernel_stats* must agree with.  In the timed region above the four branches overlap.
        if stepper.reducer is not None:
            stepper.reducer.set_manual(False)
        stepper.opt.zero_grad()
        ms_default = runtime.multistream[0]
        runtime.multistream[0] = False
        # The captured step's autograd graph (kept alive by its static loss) holds the parameters' AccumulateGrad nodes, each tied to the
        # BRANCH stream of its first backward; this pass runs every branch on one stream, and autograd would synchronise -- and warn -- on
        # every such gradient.  Release the captured graph first: the nodes are re-created on this pass's stream.
        graph = static_loss = loss = None                    # noqa: F841
        import gc
        gc.collect()
        # (nodes that survive -- kept alive elsewhere -- still trigger the warning; the mismatch is this pass's purpose: each launch alone on
        # the device, one stream.  The extra synchronisation it warns of cannot change a per-launch event time.)
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        eager_step()                                         # shapes may have changed (secondary batch): re-warm
        ops.profile_start(("gemm_nt", "gemm_tn", "attn", "ln"))
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        flops, secs, launches = ops.profile_stop("fp8" if args.dtype.startswith("fp8") else "bf16")
        runtime.multistream[0] = ms_default
        ach = flops / max(secs, 1e-9) / 1e12
        peak = MFMA_PEAK_FP8_TFLOPS if args.dtype.startswith("fp8") else MFMA_PEAK_BF16_TFLOPS
    if rank == 0 and not args.no_roofline and args.dtype != "fp32":
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure is the committed rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE summary of this same command (tools/pmc_traffic.py: 2 x FETCH + WRITE averaged over every gemm_nt
        # launch), valid only while the kernel sources and the workload are the ones that were profiled
        traffic, traffic_src = None, "no PMC summary committed for this code/workload"
        try:
            with open(PMC_SUMMARY) as f:
                rec = json.load(f)
            same = rec.get("kernel_source_hash") == kernel_source_hash() and rec.get("per_gpu_batch") == b and rec.get("preset") == args.preset
            if same:
                traffic = rec["families"]["gemm_nt_kernel"]["hbm_bytes_per_launch_corrected"]
                traffic_src = f"recorded: {os.path.relpath(PMC_SUMMARY, ROOT)} (kernel sources {rec['kernel_source_hash']}, commit {rec.get('head', '?')})"
            else:
                traffic_src = (f"stale: {os.path.relpath(PMC_SUMMARY, ROOT)} was taken on kernel sources {rec.get('kernel_source_hash')} / batch "
                               f"{rec.get('per_gpu_batch')}, now {kernel_source_hash()} / {b}")
        except Exception:
            pass
        # the other kernel families, timed in the same instrumented pass (HIP event pair per launch, each launch alone on the device):
        # attention (forward 4 B h S^2 d, backward 8 B h S^2 d algorithmic FLOPs -- the contract's fwd + bwd = 3 x fwd; the backward EXECUTES 14, it
        # recomputes S and dP in both of its kernels to stay free of atomics: `frac_executed`), the weight-gradient GEMMs (2 tokens N1 N2) and the
        # LayerNorm kernels (HBM-bound: bytes read + written once)
        fams = {}
        for name, d in getattr(ops.profile_stop, "families", {}).items():
            if name == "gemm_nt" or not d["launches"]:
                continue
            if name == "ln":
                a_ = d["bytes"] / max(d["secs"], 1e-9) / 1e9
                fams["layernorm"] = {"kernel": "tav::ln_fwd_kernel / ln_bwd_kernel", "bound": "hbm", "achieved": round(a_, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(a_ / HBM_PEAK_GBS, 4)}
            else:
                a_ = d["work"] / max(d["secs"], 1e-9) / 1e12
                fams[{"attn": "attention", "gemm_tn": "gemm_tn"}[name]] = {
                    "kernel": {"attn": "tav::attn_fwd_kernel / attn_bwd_dq_kernel / attn_bwd_dkdv_kernel", "gemm_tn": "tav::gemm_tn_grouped_big_kernel / gemm_tn_grouped_kernel / gemm_tn_kernel"}[name],
                    "bound": "mfma", "achieved": round(a_, 2), "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(a_ / MFMA_PEAK_BF16_TFLOPS, 4)}
                if name == "attn":
                    fams["attention"]["frac_executed"] = round(d["work_exec"] / max(d["secs"], 1e-9) / 1e12 / MFMA_PEAK_BF16_TFLOPS, 4)
                    fams["attention"]["flops_convention"] = "algorithmic 4 (fwd) + 8 (bwd) B h S^2 d = 12 per layer; executed 4 + 14"
            f_ = fams["layernorm" if name == "ln" else {"attn": "attention", "gemm_tn": "gemm_tn"}[name]]
            f_.update({"launches_per_step": d["launches"] // 2, "avg_launch_us": round(d["secs"] / d["launches"] * 1e6, 2), "serial_ms_per_step": round(d["secs"] / 2 * 1e3, 3),
                       "algorithmic_bytes_per_launch": int(d["bytes"] / d["launches"])})
        roof = {"kernel": f"tav::gemm_nt_kernel<{args.dtype},*>", "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(getattr(ops.profile_stop, "algorithmic_bytes", 0.0) / max(launches, 1)), "launches_per_step": launches // 2,
                "avg_launch_us": round(secs / launches * 1e6, 2), "serial_ms_per_step": round(secs / 2 * 1e3, 3)}
        # HBM traffic of the other families' kernels from the same committed PMC passes (bytes per launch of each kernel, averaged over every launch of
        # the replayed steps: all four stacks' shapes), valid while THAT family's sources are the ones that were profiled
        try:
            for fam, f_ in fams.items():
                want = kernel_source_hash(FAMILY_SOURCES[fam])
                ok = rec.get("family_source_hashes", {}).get(fam) == want and rec.get("per_gpu_batch") == b and rec.get("preset") == args.preset
                f_["traffic"] = ({k: rec["families"][k]["hbm_bytes_per_launch_corrected"] for k in FAMILY_KERNELS[fam] if k in rec["families"]} if ok else None)
                f_["traffic_source"] = (f"recorded: {os.path.relpath(PMC_SUMMARY, ROOT)} (bytes per kernel launch; sources {want})" if ok else
                                        f"stale or absent: {os.path.relpath(PMC_SUMMARY, ROOT)} has {rec.get('family_source_hashes', {}).get(fam)} / batch "
                                        f"{rec.get('per_gpu_batch')}, now {want} / {b}")
        except Exception:
            pass
        if fams:
            roof["families"] = fams

    cpu_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t0 = time.time()
        def gpu_loss(batch, lab):
            from tav_amd.train_model.tav_train import get_statistics
            dev_in = [{"input_ids": batch["input_ids"].to(dev), "attention_mask": batch["text_mask"].to(dev)},
                      {"audio_features": batch["audio_features"].to(dev), "attention_mask": batch["audio_mask"].to(dev)},
                      {"visual_embeds": batch["video_embeds"].to(dev), "attention_mask": batch["visual_mask"].to(dev)}]
            with torch.no_grad():
                ls = get_statistics(dev_in, lab.to(dev), model, pre, CrossEntropyLoss(), None, check="val", epoch=0, n_visual_true=n_true)
            torch.cuda.synchronize()
            return float(ls)
        cpu_ref = cpu_baseline(torch, synthetic, cfg, pre, model, args.cpu_protocol, gpu_loss)
        log(f"cpu baseline: {cpu_ref}  ({time.time() - t0:.1f} s)")

    if rank == 0:
        utt = world * b * args.steps
        value = utt / elapsed
        step_flops = 3 * fwd_gf * 1e9 * b
        what = "fwd+bwd" if args.no_optimizer else "fwd+bwd (+clip_grad_norm_+AdamW inside the timed step)"
        out = {
            "metric": f"utterances/sec {what}, TAV (BERT+Wav2Vec2+VideoMAE) b={gb}, 1/2/4/8 MI355X",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearse else ""),
            "config": {"workload": f"tav_nn.py TAV preset {args.preset} ({describe(cfg)}), global batch {gb} = {b} per GPU x {world}, text 128 tok, audio 80000 "
                                   f"samples, video {cfg['video']['frames']}x3x{cfg['video']['image']}x{cfg['video']['image']} ({n_true} fusion / "
                                   f"{(cfg['video']['image'] // 16) ** 2 * (cfg['video']['frames'] // 2) - n_true} encoder tokens), {fwd_gf:.1f} GFLOP forward per utterance",
                       "global_batch": gb, "per_gpu_batch": b, "parallelism": f"dp{world}",
                       "step": "PreFormer+TAVForMAE fwd (check=\"val\": head dropout off, the parity configuration), CE, bwd"
                               + (", grad all-reduce (RCCL)" if world > 1 else "") + ("" if args.no_optimizer else ", clip_grad_norm_, AdamW"),
                       "weights": "random init (seeded), no checkpoints offline", "final_loss": round(final_loss, 5), "launch": launch,
                       "mfma_util_whole_step": round(step_flops / (elapsed / args.steps) / (MFMA_PEAK_BF16_TFLOPS * 1e12), 4)},
        }
        if secondary:
            out["config"]["secondary"] = secondary
        if roof is not None:
            out["roofline"] = roof
        if cpu_ref is not None:
            out["cpu_baseline"] = cpu_ref
        print(json.dumps(out), flush=True)
    if abi_comm is not None:
        torch.cuda.synchronize()
        abi_comm.destroy()
    if world > 1 or alone_ddp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
