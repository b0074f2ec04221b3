"""Data-parallel gradient exchange for the TAV step: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all; the unit of
sharding is the utterance, the only exchange is the gradient mean.

Buckets are filled in the order gradients become ready (reverse execution order: tail -> fusion -> text/video/audio ->
front-ends -> PreFormer), ~48 MiB each so a reduce-scatter/all-gather ring step moves multi-MB chunks per xGMI link.  When the
last gradient of a bucket has been accumulated, a side HIP stream waits on an event, packs the bucket, launches the
all-reduce (average) and re-points the parameters' .grad at views of the reduced bucket (no copy back).  The main stream
keeps running the backward of earlier layers meanwhile; `finish()` joins the side stream before clipping / the update.
"""
import os

import torch
import torch.distributed as dist



def _rt():
    from . import runtime          # (lazy: runtime imports engine, which imports this module's users)
    return runtime

class AbiComm:
    """An RCCL communicator owned through the C ABI (include/tavhip.h: tav_comm_*, tav_allreduce_bucket) next to the process group: a raw
    ncclAllReduce on the CALLER's stream, which -- unlike a torch.distributed collective, whose process group keeps a watchdog thread
    polling events -- can be captured into the step's hipGraph.  The 128-byte id travels from rank 0 over the existing process group."""

    def __init__(self, pg, world, rank):
        import ctypes
        from ._lib import check, lib
        h = lib()
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            check(h.tav_comm_unique_id(uid), "comm_unique_id")
        if world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(pg, 0) if pg is not None else 0, group=pg)
            uid = ctypes.create_string_buffer(box[0], 128)
        self._h, self.comm = h, ctypes.c_void_p()
        check(h.tav_comm_init_rank(ctypes.byref(self.comm), world, uid, rank), "comm_init_rank")       # collective: every rank is here
        ver = ctypes.c_int32()
        h.tav_comm_rccl_version(ctypes.byref(ver))
        self.version = ver.value

    def allreduce_mean(self, t, stream):
        """In place, mean over the ranks, enqueued on `stream` (a torch.cuda.Stream); f32 or bf16."""
        from ._lib import check, dt
        check(self._h.tav_allreduce_bucket(t.data_ptr(), t.numel() * t.element_size(), dt(t), self.comm, stream.cuda_stream), "allreduce_bucket")

    def destroy(self):
        if self.comm is not None and self.comm.value:
            self._h.tav_comm_destroy(self.comm)
        self.comm = None


class BucketedAllReduce:
    def __init__(self, params, bucket_mb=48.0, process_group=None, reduce_dtype=None, single_rank_ok=False):
        self.params = [p for p in params if p.requires_grad]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._active = self.world > 1 or (single_rank_ok and dist.is_initialized())     # single_rank_ok: run the collectives even alone (tests)
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self.reduce_dtype = reduce_dtype            # e.g. torch.bfloat16 halves the bytes on the wire
        self.order = None                            # ready order learned during the first backward
        self.buckets = None
        self._ready = []
        self._pending = {}
        self._streams = {}                           # bucket -> streams its gradients were written on (the branches run on their own)
        self._works = []
        self._complete = set()                       # buckets whose gradients have all arrived (hook mode)
        self._next = 0                               # hook mode launches buckets STRICTLY in index order: next one to go
        self._manual = False
        self._cuda = self.params[0].is_cuda
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"    # RCCL has AVG; gloo sums then scales
        self.side = torch.cuda.Stream() if self._cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_ready) for p in self.params]
        self._sync_ln_defer()

    # ---- bucket planning -------------------------------------------------------------------------------------------
    def _plan(self, order):
        buckets, cur, size = [], [], 0
        for p in order:
            nb = p.numel() * 4
            if cur and size + nb > self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nb
        if cur:
            buckets.append(cur)
        self.buckets = []
        for plist in buckets:
            n = sum(p.numel() for p in plist) + len(plist)         # + one "used" flag per parameter (see _launch)
            flat = torch.empty(n, dtype=torch.float32, device=plist[0].device)
            flat[n - len(plist):].fill_(1.0)                        # (manual / graph mode never rewrites the flags: every gradient is required there)
            self.buckets.append((plist, flat))
        self._bucket_of = {p: i for i, (plist, _) in enumerate(self.buckets) for p in plist}

    # ---- hooks -----------------------------------------------------------------------------------------------------
    def _on_ready(self, p):
        if not self._active or self._manual:
            return
        if self.buckets is None:
            self._ready.append(p)          # first step: learn the order, reduce everything in finish()
            return
        i = self._bucket_of.get(p)
        if i is None:
            return
        left = self._pending.get(i)
        if left is None:
            left = self._pending[i] = set(self.buckets[i][0])
        left.discard(p)
        if self._cuda:
            # this hook runs under the stream guard of the parameter's AccumulateGrad node: remember every stream that wrote a
            # gradient of this bucket (text / audio / video branches accumulate on their own streams, runtime.branch_streams)
            self._streams.setdefault(i, set()).add(torch.cuda.current_stream())
        if not left:
            # Collectives must be issued in the SAME order on every rank.  A rank that misses a gradient of bucket 0 would otherwise
            # launch 1, 2, ... from its hooks and 0 only in finish(), against 0, 1, 2 on the other ranks (differently sized
            # all-reduces paired up: a hang or mixed buckets).  So bucket i goes only once 0 .. i-1 have gone; whatever is still
            # held back when the backward ends is launched by finish(), again in index order.
            self._complete.add(i)
            while self._next in self._complete:
                self._launch(self._next)
                self._next += 1

    def _launch(self, i):
        plist, flat = self.buckets[i]
        if self._cuda:
            # the bucket's gradients were produced on several streams (buckets follow the ready order and span branches): the
            # reducer stream waits for the tail of EVERY one of them, not only of the stream that completed the bucket
            for st in self._streams.pop(i, set()) | {torch.cuda.current_stream()}:
                _rt().stream_wait(self.side, st)
            ctx = torch.cuda.stream(self.side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx, torch.no_grad():
            off = 0
            views = []
            missing = [p.grad is None for p in plist]
            for p in plist:
                n = p.numel()
                v = flat[off:off + n].view_as(p)
                if p.grad is None:
                    v.zero_()                        # no gradient on this rank this step: contribute zeros, keep the collective sequence
                else:
                    v.copy_(p.grad)
                views.append(v)
                off += n
            # tail of the bucket: one "this rank produced a gradient" flag per parameter, reduced with the data.  A parameter whose flag
            # is zero on EVERY rank was used nowhere this step and keeps .grad = None (as torch DDP and the single-GPU step leave it:
            # the optimizer must not decay it); the flags are read back only by a rank that misses a gradient itself.
            flat[off:].fill_(1.0)
            for j, m in enumerate(missing):
                if m:
                    flat[off + j].zero_()
            buf = flat if self.reduce_dtype is None else flat.to(self.reduce_dtype)
            if self._cuda:
                buf.record_stream(self.side)
            work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((work, i, buf, views, any(missing)))

    def finish(self):
        """Join all outstanding reductions; afterwards every p.grad holds the mean over ranks."""
        if not self._active or self._manual:
            return
        if self.buckets is None:                     # first step: plan from the observed order, then reduce all
            seen = set()
            order = [p for p in self._ready if not (p in seen or seen.add(p))]
            self._plan(order)
            self._ready = []
            for i in range(len(self.buckets)):
                self._launch(i)
        else:
            # Buckets not launched by the hooks: some parameter received no gradient on this rank this step.  Every rank must still
            # issue the same collectives in the same order, so ALL remaining buckets are reduced, in index order, with zeros in the
            # slots of the missing gradients (their .grad then becomes the mean over the ranks that did produce one).
            for i in range(self._next, len(self.buckets)):
                self._launch(i)
        import contextlib
        for work, i, buf, views, any_missing in self._works:
            plist, flat = self.buckets[i]
            with (torch.cuda.stream(self.side) if self._cuda else contextlib.nullcontext()), torch.no_grad():
                work.wait()                          # side stream (not the main one) waits for the collective
                if buf is not flat:
                    flat.copy_(buf)
                if not self._avg:
                    flat.mul_(1.0 / self.world)
            used = None
            if any_missing:                          # (a host read, only on a rank and step that missed a gradient)
                if self._cuda:
                    self.side.synchronize()
                used = (flat[flat.numel() - len(plist):] > 0).tolist()
            for j, (p, v) in enumerate(zip(plist, views)):
                p.grad = v if used is None or used[j] or p.grad is not None else None
        self._works = []
        self._pending = {}
        self._streams = {}
        self._complete = set()
        self._next = 0
        if self._cuda:
            _rt().stream_wait(torch.cuda.current_stream(), self.side)

    # ---- graph mode: no hooks, no side stream.  The backward (captured in one hipGraph) ends with pack_all(); the collectives are
    # issued eagerly between that graph and the optimizer graph (no RCCL call is ever captured); see bench.py.
    def set_manual(self, on=True, bucket_mb=None):
        """Graph mode on/off.  Turning it on drops the learned bucket plan: pack_all() re-plans from the parameter list order, which
        is identical on every rank by construction (and may use larger buckets: nothing overlaps in this mode)."""
        self._manual = bool(on)
        self._sync_ln_defer()
        if on:
            self.buckets = None
            self._pending, self._works, self._ready = {}, [], []
            self._complete, self._next = set(), 0
            if bucket_mb is not None:
                self.bucket_bytes = int(bucket_mb * 2 ** 20)

    def pack_all(self):
        """Copy every gradient into its bucket and re-point .grad at the bucket view (call right after backward; capturable)."""
        if self.buckets is None:
            self._plan([p for p in reversed(self.params) if p.grad is not None])
        with torch.no_grad():
            for plist, flat in self.buckets:
                live = [p for p in plist if p.grad is not None]
                if len(live) != len(plist):
                    raise RuntimeError("graph-mode reducer needs every bucketed parameter to receive a gradient each step")
                views, off = [], 0
                for p in plist:
                    n = p.numel()
                    views.append(flat[off:off + n].view_as(p))
                    off += n
                torch._foreach_copy_(views, [p.grad for p in plist])
                for p, v in zip(plist, views):
                    p.grad = v

    def reduce_packed(self):
        """All-reduce (mean) the packed buckets on the current stream (eager)."""
        if not self._active:
            return
        for _, flat in self.buckets:
            buf = flat if self.reduce_dtype is None else flat.to(self.reduce_dtype)
            dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg)
            if buf is not flat:
                flat.copy_(buf)
            if not self._avg:
                flat.mul_(1.0 / self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._sync_ln_defer()

    def _sync_ln_defer(self):
        """Hook mode packs a gradient the moment autograd has accumulated it -- before the backward pass ends -- so the LayerNorm parameter
        gradients must be complete then: the deferred second stage (ops.ln_flush, run when the pass ends) is switched off while hooks are live."""
        from . import ops
        ops.ln_defer[0] = not (self._active and not self._manual and bool(self._hooks))


# ====================================================================================================================
# Graph mode with overlap: the backward in segments.
#
# A captured hipGraph cannot contain the RCCL calls (kept eager on purpose) and an eager hook-driven backward is host bound, so the
# data-parallel step is a CHAIN of graphs:   G0 = forward + loss + backward of the top segment + pack bucket 0
#                                            Gs = backward of segment s + pack bucket s            (s = 1 .. S-1)
#                                            Gu = clip_grad_norm_ + AdamW on the reduced buckets
# and between two graphs the host records an event on the work stream, makes the reducer stream wait for it and issues the
# all-reduce of the bucket that graph just packed: bucket s crosses xGMI while graph s+1 differentiates the next lower layers.
# Segments are cut where the encoder stacks reported cut tensors (runtime.cut_point): every segment spans ALL four branches (text /
# audio / video / fusion keep running side by side on their streams inside each graph), the lowest one holds the first layers, the
# front-ends, the embedding tables and PreFormer and is the only one whose all-reduce is exposed.
def _walk_leaves(root_fns, stop_fns):
    """Leaf tensors (AccumulateGrad variables) reachable from the autograd nodes `root_fns` without entering `stop_fns`."""
    seen, out, stack = set(), [], [f for f in root_fns if f is not None]
    while stack:
        fn = stack.pop()
        if fn in seen or fn in stop_fns:
            continue
        seen.add(fn)
        if hasattr(fn, "variable"):
            out.append(fn.variable)
            continue
        stack.extend(nf for nf, _ in fn.next_functions if nf is not None)
    return out


class SegmentedBackward:
    """`loss.backward()` executed as S calls.  `cuts` = {branch: [(x, x_cut), ...] in forward order} as recorded by runtime.cut_point:
    the forward continued from the detached leaf x_cut, so the autograd graph is in pieces.  Segment s differentiates the pieces whose
    roots are the x of the cuts segment s-1 ended at (the loss for s = 0), feeding them the gradients that arrived at the matching
    x_cut; a branch with fewer cuts than the others reaches its leaves in an earlier segment.  Pure autograd bookkeeping (device
    agnostic); a parameter used by several pieces accumulates and is final after the last of them."""

    def __init__(self, loss, cuts):
        per = [list(reversed(v)) for v in cuts.values() if v]
        self.nseg = 1 + max((len(v) for v in per), default=0)
        self.roots = [[loss]] + [[v[s][0] for v in per if len(v) > s] for s in range(self.nseg - 1)]
        self.root_src = [[None]] + [[v[s][1] for v in per if len(v) > s] for s in range(self.nseg - 1)]     # whose gradient feeds each root
        self.stops = [[v[s][1] for v in per if len(v) > s] for s in range(self.nseg - 1)] + [[]]
        cut_leaves = {id(xc) for v in per for _, xc in v}
        last = {}
        self.leaves = []
        for s in range(self.nseg):
            uniq, ids = [], set()
            for p in _walk_leaves([t.grad_fn for t in self.roots[s]], set()):
                if id(p) not in ids and id(p) not in cut_leaves and p.requires_grad:
                    ids.add(id(p))
                    uniq.append(p)
            self.leaves.append(uniq)
            for p in uniq:
                last[id(p)] = s
        # parameters whose gradient is complete once segment s has run (what bucket s may ship)
        self.final = [[p for p in self.leaves[s] if last[id(p)] == s] for s in range(self.nseg)]
        self._pending = {}

    def run(self, s):
        """Differentiate segment s; gradients land in p.grad (added to an existing one).  Returns the parameters that are final now.
        torch.autograd.grad rather than .backward(): captured gradients register their producer stream with the engine's final
        stream join, so the branch streams are handled exactly as in a whole backward, and no AccumulateGrad node (with its own
        stream affinity) is involved."""
        roots = self.roots[s]
        grads = None if s == 0 else [self._pending.pop(id(xc)) for xc in self.root_src[s]]
        stops, leaves = list(self.stops[s]), self.leaves[s]
        out = torch.autograd.grad(roots, stops + leaves, grad_outputs=grads, allow_unused=True)
        for xc, g in zip(stops, out[:len(stops)]):
            self._pending[id(xc)] = g
        for p, g in zip(leaves, out[len(stops):]):
            if g is not None:
                p.grad = g if p.grad is None else p.grad + g
        return self.final[s]


_SKIP_REDUCE = os.environ.get("TAV_DDP_SKIP_REDUCE", "0") == "1"
ARENA = os.environ.get("TAV_DDP_ARENA", "1") == "1"      # gradient arena: the layers' weight-gradient kernels write into the bucket (0: pack everything by copy)     # measurement knob: time the graph chain without the collective (results unreduced)


class GraphedStep:
    """The data-parallel training step of bench.py (N > 1): S+1 hipGraphs with the bucket all-reduces issued eagerly between them on the
    reducer stream (see the block comment above).  `forward_loss()` must run PreFormer + model + criterion on static input tensors.
    `use_graphs=False` runs the same chain eagerly (forward, segments, packs, collectives in the same order) on any device: that is what
    the 2-rank gloo test on the CPU exercises, and a debugging aid on the GPU."""

    def __init__(self, stepper, forward_loss, stream=None, segments=4, use_graphs=True, fractions=None, mode="chain", tail_bf16=False, shard_optimizer=False):
        from . import runtime
        red = stepper.reducer
        self.red, self.stepper, self.stream = red, stepper, stream
        red.set_manual(True)
        self.side = red.side
        self.use_graphs = bool(use_graphs)
        self.mode = mode if self.use_graphs else "chain"
        self.comm = None
        self.tail_bf16 = bool(tail_bf16)
        # shard_optimizer: bucket s is reduced to the OWNERS of its ranges instead of all-reduced, each rank clips and updates the ranges it owns
        # and the updated parameters come back through the buckets (optim.ShardedAdamW: same bytes on the wire, 1/N of the optimizer's work and
        # state per GPU).  The optimizer then consists of three device phases with two exchanges between them: three graphs in the chain.
        self.shard = None
        if shard_optimizer:
            if self.mode != "chain":
                raise ValueError("GraphedStep(shard_optimizer=True) runs as the graph chain (mode='chain'): its exchanges are torch.distributed calls")
            from .optim import ShardedAdamW
            if not isinstance(stepper.opt, ShardedAdamW):
                stepper.opt = ShardedAdamW.from_replicated(stepper.opt, red.world, dist.get_rank(red.pg) if dist.is_initialized() else 0, red.pg)
            self.shard = stepper.opt
            self.shard.exchange_alone = bool(red._active and red.world == 1)
        self._forward_loss = forward_loss
        k = max(1, int(segments))
        if fractions is None:
            fractions = tuple(runtime.CUT_FRACTIONS) if k == 4 else tuple((j + 0.25) / k for j in range(k - 1)) if k > 1 else ()
        self.fractions = tuple(fractions)
        self.graphs, self.flats, self.groups = [], [], []
        self._plan = None                           # [(parameter names / sizes per bucket)] fixed by the first pass, checked on later eager ones
        if not self.use_graphs:
            self.loss = None
            return
        # capture_error_mode "thread_local": the process group's watchdog thread polls its work events (hipEventQuery) at any time;
        # under the default global mode such a call from ANOTHER thread invalidates the capture (hipErrorStreamCaptureUnsupported)
        mode = dict(capture_error_mode="thread_local")
        if self.mode == "single":
            self._capture_single(stepper, forward_loss, stream, mode)
            return
        g0 = torch.cuda.CUDAGraph()
        runtime.begin_cuts(self.fractions)
        runtime.begin_layer_groups()
        with runtime.capture(g0, stream, **mode):
            self.loss = forward_loss()
            self.seg = SegmentedBackward(self.loss, runtime.end_cuts())
            self.groups = runtime.end_layer_groups()
            self._run_and_pack(0)
        self.graphs.append(g0)
        for s in range(1, self.seg.nseg):
            g = torch.cuda.CUDAGraph()
            with runtime.capture(g, stream, pool=g0.pool(), **mode):
                self._run_and_pack(s)
            self.graphs.append(g)
        self._adopt_buckets()
        self.check_plan_across_ranks()
        if self.shard is not None:
            self._attach_shards()
            self.shard.finalize()                                  # (allocates the state slices and tables: outside the captures)
            self.gu = []
            for k, phase in enumerate(self._shard_phases()):
                if k == 0 and getattr(stepper, "clip", None) is None:
                    self.gu.append(None)                   # no clipping: no norm phase, no exchange of partials (an empty capture is not a graph)
                    continue
                g = torch.cuda.CUDAGraph()
                with runtime.capture(g, stream, pool=g0.pool(), **mode):
                    phase()
                self.gu.append(g)
            return
        self.gu = torch.cuda.CUDAGraph()
        with runtime.capture(self.gu, stream, pool=g0.pool(), **mode):
            stepper.update()

    def _attach_shards(self):
        for s, (plist, flat) in enumerate(self.flats):
            if plist:
                self.shard.attach_bucket(s, plist, flat)

    def _shard_phases(self):
        """The three device phases of the sharded optimizer step (what TrainStep.update() is for the replicated one); exchange_norm() runs
        between the first two, exchange_params() between the last two."""
        st, sh = self.stepper, self.shard
        clip = getattr(st, "clip", None)

        def norm():
            if clip is not None:
                sh.phase_norm()

        def update():
            sh.phase_update(clip)

        def adopt():
            sh.phase_adopt()
            sh.zero_grad(set_to_none=getattr(st, "zero_to_none", True))
        return norm, update, adopt

    def _capture_single(self, stepper, forward_loss, stream, mode):
        """Round 4: the WHOLE data-parallel step as ONE hipGraph.  The collective of bucket s is a raw RCCL all-reduce (AbiComm) captured on the
        reducer stream, a branch forked from the capture's origin right after segment s has packed its bucket and joined back just before the
        optimizer: it crosses xGMI while the origin and the three encoder branches differentiate segment s+1, exactly as in the chain, but
        without the S graph boundaries (each drained all four branches and cost a graph launch).  Legal capture topology only: every edge
        starts or ends at the origin (runtime.stream_wait enforces it)."""
        from . import runtime
        red = self.red
        if not red._cuda or red.side is None:
            raise RuntimeError("GraphedStep(mode='single') needs the GPU reducer stream")
        self.comm = AbiComm(red.pg, red.world, dist.get_rank(red.pg) if dist.is_initialized() else 0)
        # RCCL sets up its channels on first use: that must not happen inside a capture
        warm = torch.zeros(1 << 16, dtype=torch.float32, device=red.params[0].device)
        with torch.cuda.stream(red.side):
            self.comm.allreduce_mean(warm, red.side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        runtime.begin_cuts(self.fractions)
        runtime.begin_layer_groups()
        with runtime.capture(g, stream, branches=[red.side], **mode):
            self.loss = forward_loss()
            self.seg = SegmentedBackward(self.loss, runtime.end_cuts())
            self.groups = runtime.end_layer_groups()
            for s in range(self.seg.nseg):
                self._run_and_pack(s)
                flat = self.flats[s][1]
                if flat is None or not red._active or _SKIP_REDUCE:
                    continue
                runtime.stream_wait(red.side, stream)               # fork: bucket s is packed
                with torch.cuda.stream(red.side):
                    wd = red.reduce_dtype if red.reduce_dtype is not None else (torch.bfloat16 if (self.tail_bf16 and s == self.seg.nseg - 1) else None)
                    buf = flat if wd is None else flat.to(wd)
                    self.comm.allreduce_mean(buf, red.side)
                    if buf is not flat:
                        flat.copy_(buf)
            runtime.stream_wait(stream, red.side)                   # join: every bucket reduced
            stepper.update()
        self.graphs = [g]
        self.gu = None
        self._adopt_buckets()
        self.check_plan_across_ranks()

    def _adopt_buckets(self):
        red = self.red
        if self.shard is None:
            red.buckets = [(plist, flat) for plist, flat in self.flats if plist]  # hook mode, if switched back on, reuses these buckets
            red._bucket_of = {p: i for i, (plist, _) in enumerate(red.buckets) for p in plist}
        else:
            red.buckets = None                       # (slice padding sits between gradients and flags: hook mode plans its own buckets)
        red._pending, red._streams, red._works = {}, {}, []
        self._bucket_norm()

    def _bucket_norm(self):
        """The replicated optimizer takes the clip norm over the buckets (FusedAdamW.norm_buffers): every gradient of this mode lives in one, four
        tensors instead of ~400, and the chunk grid a sharded optimizer's ranks reproduce bit for bit."""
        opt = getattr(self.stepper, "opt", None)
        if self.shard is None and self.mode == "chain" and opt is not None and hasattr(opt, "prepare_norm_buffers"):
            opt.norm_buffers = [flat[:sum(p.numel() for p in plist)] for plist, flat in self.flats if plist]
            if opt.norm_buffers and opt.norm_buffers[0].is_cuda:
                opt.prepare_norm_buffers()               # (tables uploaded here, between the captures; the single-graph mode keeps the per-parameter norm)

    def _bucket_order(self, params):
        """Order of a segment's parameters inside its bucket: as the autograd walk found them, except that the members of one transformer
        layer (runtime.note_layer_group) stand together in the order wq wk wv | bq bk bv | wo bo w1 b1 w2 b2 -- the grouped weight-gradient
        launch writes Wq | Wk | Wv as ONE fused tensor, so their slots must be adjacent.  Deterministic, hence identical on every rank."""
        member = {}
        for gi, grp in enumerate(self.groups):
            for p in grp:
                if p is not None:
                    member[id(p)] = gi
        have = {id(p) for p in params}
        out, done = [], set()
        for p in params:
            gi = member.get(id(p))
            if gi is None or not all(q is None or id(q) in have for q in self.groups[gi]):
                if id(p) not in done:
                    out.append(p)
                    done.add(id(p))
                continue
            if gi in done:
                continue
            done.add(gi)
            for q in self.groups[gi]:
                if q is not None and id(q) not in done:
                    out.append(q)
                    done.add(id(q))
        return out

    def _flat_elems(self, n, k):
        """Elements of a bucket buffer for n gradient elements of k parameters: the gradients, the k "used" flags -- and, with a sharded optimizer,
        room for `world` equal slices in front of the flags, so that reduce-scatter / all-gather can run in place on the buffer's head."""
        if self.shard is None:
            return n + k
        return max(n, self.shard.world * self.shard.slice_elems(n)) + k

    def _run_and_pack(self, s):
        from . import runtime
        expected = self._bucket_order([p for p in self.seg.final[s] if p.requires_grad])
        flat, views = None, {}
        if expected and ARENA:
            n = sum(p.numel() for p in expected)
            reuse = self.flats[s][1] if len(self.flats) > s and self.flats[s][1] is not None else None     # (eager mode: keep the buffers)
            # allocated inside the capture, i.e. from the graphs' private pool: the reference kept in self.flats pins it for good, and
            # the reducer / RCCL streams only ever touch it between two replays.  (+ the reducer's per-parameter "used" flags, all ones:
            # in this mode every bucketed parameter must produce a gradient every step)
            if reuse is not None and reuse.numel() == self._flat_elems(n, len(expected)):
                flat = reuse
            else:
                flat = torch.empty(self._flat_elems(n, len(expected)), dtype=torch.float32, device=expected[0].device)
                flat[n:].fill_(1.0)
            off = 0
            for p in expected:
                views[id(p)] = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            # the gradient ARENA: the layers' grouped weight-gradient launches of this segment write straight into these views
            runtime.grad_slots.clear()
            runtime.grad_slots.update({p.data_ptr(): views[id(p)] for p in expected})
        try:
            plist = [p for p in self.seg.run(s) if p.grad is not None]
        finally:
            runtime.grad_slots.clear()
        if not plist:                                # a segment that finalises no parameter (shallow stack, few cuts): nothing to ship
            if len(self.flats) > s:
                self.flats[s] = ([], None)
            else:
                self.flats.append(([], None))
            return
        if flat is not None and {id(p) for p in plist} == {id(p) for p in expected}:
            plist = expected                         # (bucket order = the planned one)
        else:
            # (no arena, or a parameter of the plan produced no gradient: lay the bucket out from what is there)
            plist = self._bucket_order(plist)
            n = sum(p.numel() for p in plist)
            flat = torch.empty(self._flat_elems(n, len(plist)), dtype=torch.float32, device=plist[0].device)
            flat[n:].fill_(1.0)
            views, off = {}, 0
            for p in plist:
                views[id(p)] = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        src, dst = [], []
        for p in plist:
            v = views[id(p)]
            if p.grad.data_ptr() != v.data_ptr():    # not written in place by its kernel: pack it
                src.append(p.grad)
                dst.append(v)
        self.in_place = getattr(self, "in_place", 0) + sum(p.numel() for p in plist) - sum(t.numel() for t in src)
        if dst:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)
        for p in plist:
            p.grad = views[id(p)]
        if len(self.flats) > s:
            self.flats[s] = (plist, flat)
        else:
            self.flats.append((plist, flat))

    def bucket_signature(self):
        """[(number of parameters, total elements)] per bucket: must be identical on every rank (each rank derives its buckets from
        its own autograd graph; check_plan_across_ranks() compares it on every rank before the chain's first replay -- bench.py gets that
        through GraphedStep.__init__ -- and the two-rank tests compare it explicitly)."""
        return [(len(pl), int(sum(p.numel() for p in pl))) for pl, _ in self.flats]

    def check_plan_across_ranks(self):
        """Every rank derives its buckets from its OWN autograd walk and arena reordering; two ranks that disagree would pair differently
        sized all-reduces (a hang, or buckets mixed up, with no diagnosis).  Gather the plans once, before the first collective of the
        chain, and raise on every rank if they differ.  Called by __init__ (graph mode) / after the first eager pass; one small object
        collective, outside any capture and outside the timed region."""
        red = self.red
        if not (dist.is_available() and dist.is_initialized()) or not red._active or red.world <= 1:
            return
        mine = self.bucket_signature()
        plans = [None] * red.world
        dist.all_gather_object(plans, mine, group=red.pg)
        if any(p != plans[0] for p in plans):
            raise RuntimeError("data-parallel bucket plans differ across ranks (rank: [(parameters, elements) per bucket]): "
                               + "; ".join(f"{r}: {p}" for r, p in enumerate(plans)))

    def _wire_dtype(self, s):
        """Wire format of bucket s: the reducer's reduce_dtype, or -- `tail_bf16` -- bf16 for the LAST bucket only: the lowest segment (first
        layers, front-ends, the two 94 MB embedding tables, PreFormer) is the one all-reduce with no backward left to hide behind, so
        halving ITS bytes halves the exposed tail while every other gradient still travels in f32."""
        red = self.red
        if red.reduce_dtype is not None:
            return red.reduce_dtype
        return torch.bfloat16 if (self.tail_bf16 and s == self.seg.nseg - 1) else None

    def _reduce(self, flat, s=None):
        red = self.red
        if flat is None:
            return
        if self.shard is not None and s is not None:
            self.shard.attach_bucket(s, self.flats[s][0], flat)     # (eager chain: the plan is met bucket by bucket in the first pass)
            if red._active and not _SKIP_REDUCE:
                self.shard.reduce_to_owners(s, self._wire_dtype(s), average_on_wire=red._avg)
            return
        if not red._active or _SKIP_REDUCE:
            return
        wd = self._wire_dtype(s) if s is not None else red.reduce_dtype
        buf = flat if wd is None else flat.to(wd)
        dist.all_reduce(buf, op=dist.ReduceOp.AVG if red._avg else dist.ReduceOp.SUM, group=red.pg)
        if buf is not flat:
            flat.copy_(buf)
        if not red._avg:
            flat.mul_(1.0 / red.world)

    def run(self):
        if not self.use_graphs:
            return self._run_eager()
        if self.mode == "single":
            self.graphs[0].replay()
            return self.loss
        main = torch.cuda.current_stream()
        for s, (g, (_, flat)) in enumerate(zip(self.graphs, self.flats)):
            g.replay()
            _rt().stream_wait(self.side, main)                  # bucket s is packed: ship it while the next graph runs
            with torch.cuda.stream(self.side):
                self._reduce(flat, s)
        _rt().stream_wait(main, self.side)
        if self.shard is not None:
            g_norm, g_update, g_adopt = self.gu
            if g_norm is not None:
                g_norm.replay()
                self.shard.exchange_norm()
            g_update.replay()
            self.shard.exchange_params()
            g_adopt.replay()
            return self.loss
        self.gu.replay()
        return self.loss

    def _run_eager(self):
        """The same chain without capture: forward + loss, then per segment {differentiate, pack bucket s, all-reduce bucket s}, then
        clip + AdamW.  On the GPU the collective of bucket s goes to the reducer stream exactly as in run()."""
        from . import runtime
        cuda = self.side is not None
        for p in self.red.params:
            p.grad = None
        runtime.begin_cuts(self.fractions)
        runtime.begin_layer_groups()
        self.loss = self._forward_loss()
        self.seg = SegmentedBackward(self.loss, runtime.end_cuts())
        self.groups = runtime.end_layer_groups()
        for s in range(self.seg.nseg):
            self._run_and_pack(s)
            flat = self.flats[s][1]
            if cuda:
                _rt().stream_wait(self.side, torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    self._reduce(flat, s)
            else:
                self._reduce(flat, s)
        if cuda:
            _rt().stream_wait(torch.cuda.current_stream(), self.side)
        sig = self.bucket_signature()
        if self._plan is None:
            self._plan = sig
            self._adopt_buckets()
            self.check_plan_across_ranks()
        elif sig != self._plan:
            raise RuntimeError(f"data-parallel bucket plan changed between steps: {self._plan} -> {sig}")
        self._bucket_norm()
        self.stepper.update()
        return self.loss

    def describe(self):
        sizes = ", ".join(f"{(flat.numel() * 4 / 2 ** 20) if flat is not None else 0:.0f}" for _, flat in self.flats)
        how = f"{len(self.graphs)} hipGraphs" if self.use_graphs else "eager chain"
        tot = sum(sum(p.numel() for p in pl) for pl, flat in self.flats if flat is not None)
        arena = f"; gradient arena: {100.0 * getattr(self, 'in_place', 0) / max(tot, 1):.0f} % of the bucket elements written in place by their kernels" if ARENA else ""
        if self.mode == "single":
            return (f"ONE hipGraph (forward, backward in {self.seg.nseg} segments, optimizer); bucket s all-reduced by a CAPTURED raw RCCL call (C ABI "
                    f"tav_allreduce_bucket, RCCL {self.comm.version}) on the reducer branch while segment s+1 runs; buckets [{sizes}] MiB{arena}")
        if self.shard is not None:
            own, tot = self.shard.owned_elements() if self.shard._ready else (0, 0)
            return (f"{how} (forward + backward cut into {self.seg.nseg} segments) + SHARDED optimizer ({'3 graphs' if self.use_graphs else 'calls'}: partial norms | "
                    f"exchange | clip + AdamW on the owned ranges | parameter broadcast | adopt); bucket s reduced to its range owners "
                    f"({'RCCL' if self.red._avg else 'gloo'}, eager, side stream) while segment s+1 runs; this rank owns {100.0 * own / max(tot, 1):.1f} % of "
                    f"{tot} elements; buckets [{sizes}] MiB{arena}")
        return (f"{how} (forward + backward cut into {self.seg.nseg} segments) + optimizer {'graph' if self.use_graphs else 'call'}; bucket s all-reduced "
                f"({'RCCL' if self.red._avg else 'gloo'}, eager, side stream) while segment s+1 runs; buckets [{sizes}] MiB{arena}")
