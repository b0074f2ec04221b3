"""Data-parallel gradient exchange for the TAV step: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all; the unit of
sharding is the utterance, the only exchange is the gradient mean.

Buckets are filled in the order gradients become ready (reverse execution order: tail -> fusion -> text/video/audio ->
front-ends -> PreFormer), ~48 MiB each so a reduce-scatter/all-gather ring step moves multi-MB chunks per xGMI link.  When the
last gradient of a bucket has been accumulated, a side HIP stream waits on an event, packs the bucket, launches the
all-reduce (average) and re-points the parameters' .grad at views of the reduced bucket (no copy back).  The main stream
keeps running the backward of earlier layers meanwhile; `finish()` joins the side stream before clipping / the update.
"""
import torch
import torch.distributed as dist


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
        self._manual = False
        self._cuda = self.params[0].is_cuda
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"    # RCCL has AVG; gloo sums then scales
        self.side = torch.cuda.Stream() if self._cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_ready) for p in self.params]

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
            n = sum(p.numel() for p in plist)
            flat = torch.empty(n, dtype=torch.float32, device=plist[0].device)
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
            self._launch(i)

    def _launch(self, i):
        plist, flat = self.buckets[i]
        if self._cuda:
            # the bucket's gradients were produced on several streams (buckets follow the ready order and span branches): the
            # reducer stream waits for the tail of EVERY one of them, not only of the stream that completed the bucket
            for st in self._streams.pop(i, set()) | {torch.cuda.current_stream()}:
                self.side.wait_stream(st)
            ctx = torch.cuda.stream(self.side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx, torch.no_grad():
            off = 0
            views = []
            for p in plist:
                n = p.numel()
                v = flat[off:off + n].view_as(p)
                if p.grad is None:
                    v.zero_()                        # no gradient on this rank this step: contribute zeros, keep the collective sequence
                else:
                    v.copy_(p.grad)
                views.append(v)
                off += n
            buf = flat if self.reduce_dtype is None else flat.to(self.reduce_dtype)
            if self._cuda:
                buf.record_stream(self.side)
            work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((work, i, buf, views))

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
            launched = {i for _, i, _, _ in self._works}
            for i in range(len(self.buckets)):
                if i not in launched:
                    self._launch(i)
        import contextlib
        for work, i, buf, views in self._works:
            plist, flat = self.buckets[i]
            with (torch.cuda.stream(self.side) if self._cuda else contextlib.nullcontext()), torch.no_grad():
                work.wait()                          # side stream (not the main one) waits for the collective
                if buf is not flat:
                    flat.copy_(buf)
                if not self._avg:
                    flat.mul_(1.0 / self.world)
            for p, v in zip(plist, views):
                p.grad = v
        self._works = []
        self._pending = {}
        self._streams = {}
        if self._cuda:
            torch.cuda.current_stream().wait_stream(self.side)

    # ---- graph mode: no hooks, no side stream.  The backward (captured in one hipGraph) ends with pack_all(); the collectives are
    # issued eagerly between that graph and the optimizer graph (no RCCL call is ever captured); see bench.py.
    def set_manual(self, on=True, bucket_mb=None):
        """Graph mode on/off.  Turning it on drops the learned bucket plan: pack_all() re-plans from the parameter list order, which
        is identical on every rank by construction (and may use larger buckets: nothing overlaps in this mode)."""
        self._manual = bool(on)
        if on:
            self.buckets = None
            self._pending, self._works, self._ready = {}, [], []
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
