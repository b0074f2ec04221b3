"""Optimiser side of the step (reference train_model/tav_train.py:61-62,148): global-norm clipping + AdamW as
multi-tensor HIP kernels driven by device pointer tables -- two launches for the norm, one for the update, no host
synchronisation (the clip coefficient stays on the device)."""
import ctypes as C

import torch

from . import engine, ops
from ._lib import check, lib, ptr, stream


class _PtrTable:
    """Pinned host staging + device array of int64 (pointers / sizes), refreshed with an async copy."""
    CAPTURE_SLOTS = 8

    def __init__(self, n, device):
        # two pinned staging buffers used in turn: the asynchronous H2D copy of step k may still be pending when the host prepares
        # step k+1 (eager mode at large batch: the GPU lags the host), so a buffer is rewritten only after ITS last copy finished
        self.host = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.done = [None, None]
        self.turn = 0
        self.last = None
        # pinned sources of CAPTURED uploads, one slot per capture, allocated here (a pinned allocation inside a capture can itself
        # invalidate it) and never rewritten once a capture has taken it
        self.cap_slots = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(self.CAPTURE_SLOTS)]
        self.cap_used = 0
        self.dev = torch.empty(n, dtype=torch.int64, device=device)

    def set(self, values):
        capturing = self.dev.is_cuda and torch.cuda.is_current_stream_capturing()
        n = len(values)
        if capturing:
            # A captured H2D copy is re-executed by every replay and reads its pinned source THEN: the source must belong to this capture
            # alone (never one of the two eager staging buffers, which later eager calls overwrite), and the upload is always recorded,
            # even when the device table already holds these values -- another graph or an eager step may rewrite it between replays.
            if self.cap_used >= len(self.cap_slots):
                raise RuntimeError(f"FusedAdamW: more than {len(self.cap_slots)} captures of the optimizer step (raise _PtrTable.CAPTURE_SLOTS)")
            buf = self.cap_slots[self.cap_used]
            self.cap_used += 1
            buf[:n].copy_(torch.tensor(values, dtype=torch.int64))
            self.dev[:n].copy_(buf[:n], non_blocking=True)
            self.last = None                         # a replay rewrites self.dev behind the host's back: the next eager call uploads again
            return self.dev
        if self.last == values:                      # unchanged (gradient arena, same parameter list): nothing to upload
            return self.dev
        k = self.turn
        self.turn ^= 1
        if self.done[k] is not None:
            self.done[k].synchronize()
        self.host[k][:n].copy_(torch.tensor(values, dtype=torch.int64))
        self.dev[:n].copy_(self.host[k][:n], non_blocking=True)
        self.done[k] = None
        if self.dev.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.done[k] = ev
        self.last = list(values)
        return self.dev


class FusedAdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt) for every parameter that
    has a gradient at step() time; parameters whose .grad is None are skipped like torch does."""

    def __init__(self, params, lr=1e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state = {}
        self._tables = None
        self._chunks = None
        self.last_norm = None
        self._lr_dev = None          # device copies of lr / step counter: graph replay must not bake them into kernel args
        self._lr_host = None
        # Data-parallel graph mode (ddp.GraphedStep): every gradient lives in one of a few flat buckets.  With norm_buffers = [those buckets (their
        # gradient part)] the clip norm is taken over the buckets -- chunks of tav_optim_chunk_elems() elements counted from each bucket's start --
        # instead of parameter by parameter: the same number up to summation order, and the grid on which a ShardedAdamW's ranks can each compute
        # the partials of their own range and still reproduce this optimizer bit for bit.
        self.norm_buffers = None
        self._norm_tables = None

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (torch >= 2.0 default): drop the gradients, the next step() skips those parameters.  set_to_none=False (the default of
        the torch 1.10 the reference pins, README_and_Requirements/requirements.txt:100): keep the tensors and fill them with zeros -- a later
        step() then still decays the weights and applies the momentum (see train_model.tav_train.grad_accum)."""
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _active(self):
        return [p for p in self.params if p.grad is not None]

    def _ensure(self, act):
        dev = act[0].device
        n = len(act)
        if self._tables is None or self._tables[0].dev.numel() < n:
            self._tables = tuple(_PtrTable(max(n, len(self.params)), dev) for _ in range(5))
            self._scal = torch.zeros(8, dtype=torch.float32, device=dev)
            self._step_dev = torch.full((1,), int(getattr(self, "_loaded_step", 0)), dtype=torch.int32, device=dev)
            self._lr_pin = torch.empty(1, dtype=torch.float32).pin_memory()
        for p in act:
            if p not in self.state:
                if p.is_cuda and torch.cuda.is_current_stream_capturing():
                    # zeros allocated inside a capture would be re-zeroed by every replay: the moments must exist before
                    raise RuntimeError("FusedAdamW: run one eager optimizer step before capturing the step into a hipGraph")
                self.state[p] = (torch.zeros_like(p, memory_format=torch.contiguous_format), torch.zeros_like(p, memory_format=torch.contiguous_format))

    def _grads_in_norm_buffers(self, act):
        """norm_buffers is a promise about where the gradients live; a step whose gradients are somewhere else (a plain eager backward after the
        graph chain set the buffers) must not take its norm from stale buckets: every active gradient has to lie inside one of the buffers and the
        buffers must hold nothing else -- otherwise the per-parameter norm is used.  Checked once per set of gradient addresses."""
        key = tuple(p.grad.data_ptr() for p in act)
        if getattr(self, "_norm_ok_key", None) != key:
            spans = [(b.data_ptr(), b.data_ptr() + b.numel() * b.element_size()) for b in self.norm_buffers]
            inside = all(any(lo <= p.grad.data_ptr() and p.grad.data_ptr() + p.grad.numel() * p.grad.element_size() <= hi for lo, hi in spans) for p in act)
            self._norm_ok = inside and sum(p.grad.numel() for p in act) == sum(b.numel() for b in self.norm_buffers)
            self._norm_ok_key = key
        return self._norm_ok

    def prepare_norm_buffers(self):
        """Device tables of norm_buffers (built on first use or when the buffers changed; allocates and uploads: not inside a capture)."""
        if self._norm_tables is None or self._norm_tables[0] != [(b.data_ptr(), b.numel()) for b in self.norm_buffers]:
            if self.norm_buffers[0].is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdamW: norm_buffers set or changed inside a capture (call prepare_norm_buffers() before it)")
            self._norm_tables = bucket_norm_tables(self.norm_buffers, int(lib().tav_optim_chunk_elems()))
        return self._norm_tables

    def clip_and_step(self, max_norm=None):
        """clip_grad_norm_(params, max_norm) (if max_norm) followed by one AdamW step.  Returns the device scalar holding
        the total gradient norm (read it with .item() only when you need it on the host)."""
        act = self._active()
        if not act:
            return None
        self._ensure(act)
        for p in act:
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        tp, tg, tm, tv, ts = self._tables
        n = len(act)
        d_p = tp.set([p.data_ptr() for p in act])
        d_g = tg.set([p.grad.data_ptr() for p in act])
        d_m = tm.set([self.state[p][0].data_ptr() for p in act])
        d_v = tv.set([self.state[p][1].data_ptr() for p in act])
        d_s = ts.set([p.numel() for p in act])
        sizes = tuple(p.numel() for p in act)
        if self._chunks is None or self._chunks[0] != sizes:        # chunk table of the chunked kernels: static as long as the sizes are
            ce = lib().tav_optim_chunk_elems()
            counts = torch.tensor([(s + ce - 1) // ce for s in sizes], dtype=torch.int64)
            prefix = (torch.cumsum(counts, 0) - counts).to(torch.int32)
            self._chunks = (sizes, prefix.to(act[0].device), int(counts.sum()))
        d_c, nchunks = self._chunks[1], self._chunks[2]
        coef = None
        if max_norm is not None and self.norm_buffers and self._grads_in_norm_buffers(act):
            _, t_g, t_s, t_c, nb, nch, part = self.prepare_norm_buffers()
            check(lib().tav_sumsq_chunked(ptr(t_g), ptr(t_s), ptr(t_c), nb, nch, ptr(part), ptr(self._scal[0:1]), stream()), "sumsq_chunked")
        elif max_norm is not None:
            part = torch.empty(nchunks, dtype=torch.float32, device=act[0].device)
            check(lib().tav_sumsq_chunked(ptr(d_g), ptr(d_s), ptr(d_c), n, nchunks, ptr(part), ptr(self._scal[0:1]), stream()), "sumsq_chunked")
        if max_norm is not None:
            check(lib().tav_clip_coef(ptr(self._scal[0:1]), float(max_norm), ptr(self._scal[1:2]), ptr(self._scal[2:3]), stream()), "clip_coef")
            coef = self._scal[1:2]
            self.last_norm = self._scal[2:3]
        if self._lr_host != self.lr:                # refresh the device-side learning rate only when the schedule moved it
            self._lr_pin[0] = self.lr
            self._scal[4:5].copy_(self._lr_pin, non_blocking=True)
            self._lr_host = self.lr
        check(lib().tav_adamw_chunked(ptr(d_p), ptr(d_g), ptr(d_m), ptr(d_v), ptr(d_s), ptr(d_c), n, nchunks, ptr(coef), ptr(self._scal[4:5]),
                                      self.betas[0], self.betas[1], self.eps, self.weight_decay, ptr(self._step_dev), ptr(self._scal[5:7]), stream()),
              "adamw_chunked")
        engine.bump_weight_epoch()          # parameters changed through raw pointers: refresh cached operand copies
        ops.fp8_roll_all()                  # fp8 policy: this step's gathered maxima become the next step's quantisation scales (no-op otherwise)
        return self.last_norm

    def step(self):
        return self.clip_and_step(None)

    # ---- checkpoint format of torch.optim.AdamW (reference utils/global_functions.py:199-258 saves optimizer.state_dict() into best.pt and
    # reloads it into a fresh AdamW over the same parameter list): {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}
    def state_dict(self):
        step = self.step_count
        state = {}
        for i, p in enumerate(self.params):
            if p in self.state:
                m, v = self.state[p]
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        idx = [i for g in groups for i in g["params"]]
        if len(idx) != len(self.params):
            raise ValueError(f"optimizer state has {len(idx)} parameters, this optimizer {len(self.params)}")
        g0 = groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]
        self._lr_host = None
        steps = set()
        for pos, i in enumerate(idx):
            st = sd["state"].get(i)
            if st is None:
                continue
            p = self.params[pos]
            if st["exp_avg"].shape != p.shape:
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
            self.state[p] = (st["exp_avg"].to(p.device, torch.float32).contiguous().clone(), st["exp_avg_sq"].to(p.device, torch.float32).contiguous().clone())
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused update keeps one counter (torch semantics when all parameters train together)")
        self._loaded_step = steps.pop() if steps else 0
        if self._tables is not None:
            self._step_dev.fill_(self._loaded_step)

    @property
    def step_count(self):
        return int(self._step_dev.item()) if self._tables is not None else int(getattr(self, "_loaded_step", 0))


def bucket_norm_tables(buffers, chunk):
    """Device tables for tav_sumsq_chunked over flat buffers taken as single tensors: (key, pointers, sizes, chunk prefix, n, chunks, partials)."""
    dev = buffers[0].device
    counts = [(b.numel() + chunk - 1) // chunk for b in buffers]
    pre, c = [], 0
    for k in counts:
        pre.append(c)
        c += k
    return ([(b.data_ptr(), b.numel()) for b in buffers], torch.tensor([b.data_ptr() for b in buffers], dtype=torch.int64).to(dev),
            torch.tensor([b.numel() for b in buffers], dtype=torch.int64).to(dev), torch.tensor(pre, dtype=torch.int32).to(dev), len(buffers), c,
            torch.zeros(max(c, 1), dtype=torch.float32, device=dev))


def shard_cuts(n, world, chunk):
    """Owner ranges of a gradient bucket of n elements: `world` EQUAL slices of L = ceil(n / world) rounded up to whole chunks -- rank r owns
    [r L, (r + 1) L) clipped to n (the last ranks of a small bucket own nothing).  Equal slices are what reduce-scatter / all-gather move in
    one call per bucket; whole chunks keep every range on the grid of the bucket-wise norm (FusedAdamW.norm_buffers).  Returns world + 1 positions."""
    L = shard_slice(n, world, chunk)
    return [min(r * L, n) for r in range(world)] + [n]


def shard_slice(n, world, chunk):
    return (-(-n // world) + chunk - 1) // chunk * chunk


def shard_pieces(plist, cuts, r):
    """What rank r owns of a bucket: [(parameter, first element inside the parameter, elements, position inside the bucket)]."""
    lo, hi, out, st = cuts[r], cuts[r + 1], [], 0
    for p in plist:
        n = p.numel()
        a, b = max(lo, st), min(hi, st + n)
        if a < b:
            out.append((p, a - st, b - a, a))
        st += n
    return out


class ShardedAdamW(FusedAdamW):
    """The optimizer of the data-parallel step with its work and its state divided over the ranks (what DeepSpeed calls ZeRO stage 1; the
    reference has no distributed code, SURVEY.md §2.1 -- this serves BASELINE.json configs[2], where at 4 utterances per GPU the replicated
    clip + AdamW is ~2.2 ms of a ~15 ms step, identical on all eight ranks).  Layout: the reducer's gradient buckets (ddp.GraphedStep: one flat
    f32 buffer per backward segment, the parameters' gradients back to back).  Rank r owns one slice of every bucket (shard_cuts) and
      1. receives the MEAN of that slice only (reduce-scatter: the first half of the all-reduce it replaces),
      2. computes the sum-of-squares partials of the chunks of its slice; the ranks exchange the partials (one small all-reduce of an array in
         which every entry is non-zero on exactly one rank) and each adds the complete array in the replicated optimizer's order: same norm, same bits,
      3. clips and updates the parameter pieces under its slice -- both moments exist only there (1/N of the optimizer state per GPU) --
      4. publishes the updated values through the bucket, which the gradients no longer need (all-gather: the second half of the all-reduce),
      5. and every rank copies the slices it does not own from the bucket into its parameters.
    Same bytes on the wire as the all-reduce, two collectives per bucket; (N-1)/N of the clip + AdamW traffic gone.  Parameters, norm and (gathered)
    moments bit-equal to FusedAdamW with norm_buffers = the buckets (tests/test_ddp_gloo.py: the host logic with two ranks on the CPU;
    tests/test_model_gpu.py: the HIP kernels, two ranks on one GPU).  RCCL ("nccl") moves the slices with reduce_scatter_tensor /
    all_gather_into_tensor in place; gloo, which has no reduce-scatter, with one reduce / broadcast per slice.
    The phases are separate methods so that ddp.GraphedStep can capture the device work into hipGraphs and issue the exchanges between them."""

    def __init__(self, params, world, rank, group=None, **kw):
        super().__init__(params, **kw)
        self.world, self.rank, self.group = int(world), int(rank), group
        self.buckets = {}                 # bucket index -> (parameters, flat buffer, cuts)
        self._ready = False
        self._full = None                 # {parameter: (exp_avg, exp_avg_sq)} of steps taken before the state was sharded (warm-up, resume)
        self.exchange_alone = False       # issue the collectives even with one rank (TAV_DDP_SINGLE_RANK=1: the one-GPU proxy of the data-parallel step)

    @classmethod
    def from_replicated(cls, opt, world, rank, group=None):
        """Continue a FusedAdamW (its warm-up steps, or a resumed best.pt) as a sharded optimizer: each rank keeps the slices it will own."""
        new = cls(opt.params, world, rank, group, lr=opt.lr, betas=opt.betas, eps=opt.eps, weight_decay=opt.weight_decay)
        new._full = dict(opt.state)
        new._loaded_step = opt.step_count
        return new

    # ---- plan ------------------------------------------------------------------------------------------------------
    def chunk_elems(self):
        return int(lib().tav_optim_chunk_elems())

    def slice_elems(self, n):
        return shard_slice(n, self.world, self.chunk_elems())

    def attach_bucket(self, b, plist, flat):
        have = self.buckets.get(b)
        if have is not None and [id(p) for p in have[0]] == [id(p) for p in plist] and have[1] is flat:
            return
        if self._ready:
            raise RuntimeError("ShardedAdamW: the bucket plan changed after the state was sharded")
        n = sum(p.numel() for p in plist)
        if flat.numel() < n:
            raise RuntimeError("ShardedAdamW: bucket buffer shorter than its parameters")
        self.buckets[b] = (list(plist), flat, shard_cuts(n, self.world, self.chunk_elems()))

    def _global_rank(self, r):
        import torch.distributed as dist
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _exchange(self):
        import torch.distributed as dist
        return (self.world > 1 or self.exchange_alone) and dist.is_available() and dist.is_initialized()

    def _whole_slices(self, b):
        """True when bucket b can travel as `world` equal slices in place (RCCL reduce-scatter / all-gather): the buffer holds world * L elements."""
        import torch.distributed as dist
        plist, flat, cuts = self.buckets[b]
        return dist.get_backend(self.group) == "nccl" and flat.numel() >= self.world * self.slice_elems(cuts[-1])

    def finalize(self):
        """Fix the plan: pieces, state slices, pointer tables.  Allocates: call it outside any capture, once every bucket is attached."""
        if self._ready:
            return
        if not self.buckets:
            raise RuntimeError("ShardedAdamW: no gradient bucket attached (ddp.GraphedStep(shard_optimizer=True) attaches them)")
        ce = self.chunk_elems()
        order = sorted(self.buckets)
        where = {}
        for b in order:
            st = 0
            for p in self.buckets[b][0]:
                if p in where:
                    raise RuntimeError("ShardedAdamW: a parameter sits in two buckets")
                where[p] = (b, st)
                st += p.numel()
        self._where = where
        known = set(self.params)
        if any(p not in known for p in where):
            raise RuntimeError("ShardedAdamW: a bucketed tensor is not one of the optimizer's parameters")
        # the norm's chunk numbering: bucket after bucket, chunks counted from each bucket's start (FusedAdamW.norm_buffers order)
        bprefix, c = {}, 0
        for b in order:
            bprefix[b] = c
            c += (self.buckets[b][2][-1] + ce - 1) // ce
        self._nchunks_all = c
        mine, theirs, my_slices = [], [], []
        for b in order:
            plist, flat, cuts = self.buckets[b]
            for r in range(self.world):
                for (p, off, n, pos) in shard_pieces(plist, cuts, r):
                    (mine if r == self.rank else theirs).append((p, off, n, flat, pos))
            lo, hi = cuts[self.rank], cuts[self.rank + 1]
            if hi > lo:
                my_slices.append((flat[lo:hi], bprefix[b] + lo // ce))
        dev = self.buckets[order[0]][1].device
        pad = lambda n: (n + 63) // 64 * 64                        # (each piece's moments on a 256-byte boundary: the 16-byte form of the update kernel)
        total = sum(pad(n) for _, _, n, _, _ in mine)
        self._m = torch.zeros(max(total, 1), dtype=torch.float32, device=dev)
        self._v = torch.zeros(max(total, 1), dtype=torch.float32, device=dev)
        self._mine, a = [], 0
        for (p, off, n, flat, pos) in mine:
            self._mine.append((p, off, n, flat, pos, a))
            if self._full is not None and p in self._full:
                fm, fv = self._full[p]
                self._m[a:a + n].copy_(fm.reshape(-1)[off:off + n])
                self._v[a:a + n].copy_(fv.reshape(-1)[off:off + n])
            a += pad(n)
        self._full = None
        self.state = {}                                           # (the replicated moments, if any, are released here)
        flatp = lambda p: p.detach().view(-1)
        self._param_mine = [flatp(p)[off:off + n] for (p, off, n, _, _, _) in self._mine]
        self._wire_mine = [flat[pos:pos + n] for (_, _, n, flat, pos, _) in self._mine]
        self._param_theirs = [flatp(p)[off:off + n] for (p, off, n, _, _) in theirs]
        self._wire_theirs = [flat[pos:pos + n] for (_, _, n, flat, pos) in theirs]
        self._slices = [t for t, _ in my_slices]
        gidx = []
        for t, g0 in my_slices:
            gidx.extend(range(g0, g0 + (t.numel() + ce - 1) // ce))
        self._gidx = torch.tensor(gidx, dtype=torch.int64, device=dev)
        self._part_all = torch.zeros(max(self._nchunks_all, 1), dtype=torch.float32, device=dev)
        self._n_local = len(self._mine)
        self._build_tables(dev, ce)
        self._ready = True

    def _build_tables(self, dev, ce):
        """Device pointer / size / chunk tables for the chunked HIP kernels (static: parameters, buckets and moments never move): the norm runs
        over this rank's slices as single tensors, the update over the parameter pieces under them."""
        i64 = lambda v: torch.tensor(v if v else [0], dtype=torch.int64).to(dev)
        self._t_p = i64([p.data_ptr() + 4 * off for (p, off, _, _, _, _) in self._mine])
        self._t_g = i64([flat.data_ptr() + 4 * pos for (_, _, _, flat, pos, _) in self._mine])
        self._t_m = i64([self._m.data_ptr() + 4 * a for (_, _, _, _, _, a) in self._mine])
        self._t_v = i64([self._v.data_ptr() + 4 * a for (_, _, _, _, _, a) in self._mine])
        sizes = [n for (_, _, n, _, _, _) in self._mine]
        self._t_s = i64(sizes)
        pre, c = [], 0
        for n in sizes:
            pre.append(c)
            c += (n + ce - 1) // ce
        self._nchunks_local = c
        self._t_c = torch.tensor(pre if pre else [0], dtype=torch.int32).to(dev)
        self._norm_local = bucket_norm_tables(self._slices, ce) if self._slices else None
        self._scal = torch.zeros(8, dtype=torch.float32, device=dev)
        self._step_dev = torch.full((1,), int(getattr(self, "_loaded_step", 0)), dtype=torch.int32, device=dev)
        self._lr_pin = torch.empty(1, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.empty(1, dtype=torch.float32)
        self._tables = ()                                          # (step_count / load_state_dict of the base class look at this)

    # ---- the device math (HIP; tests of the host logic on the CPU substitute these three) ----------------------------
    def _math_partials(self):
        """Partial sums of squares of this rank's slices, one per chunk, in slice order (returns the tensor holding them)."""
        _, t_g, t_s, t_c, nb, nch, part = self._norm_local
        check(lib().tav_sumsq_chunked(ptr(t_g), ptr(t_s), ptr(t_c), nb, nch, ptr(part), ptr(self._scal[3:4]), stream()), "sumsq_chunked")
        return part[:nch]

    def _math_coef(self, max_norm):
        check(lib().tav_sum_partials(ptr(self._part_all), self._nchunks_all, ptr(self._scal[0:1]), stream()), "sum_partials")
        check(lib().tav_clip_coef(ptr(self._scal[0:1]), float(max_norm), ptr(self._scal[1:2]), ptr(self._scal[2:3]), stream()), "clip_coef")

    def _math_update(self, clipped):
        check(lib().tav_adamw_chunked(ptr(self._t_p), ptr(self._t_g), ptr(self._t_m), ptr(self._t_v), ptr(self._t_s), ptr(self._t_c), self._n_local,
                                      self._nchunks_local, ptr(self._scal[1:2]) if clipped else None, ptr(self._scal[4:5]), self.betas[0], self.betas[1],
                                      self.eps, self.weight_decay, ptr(self._step_dev), ptr(self._scal[5:7]), stream()), "adamw_chunked")

    # ---- phases ----------------------------------------------------------------------------------------------------
    def phase_norm(self):
        """Device: partial sums of squares of the owned chunks, scattered into the (zeroed) complete array."""
        self._part_all.zero_()
        if self._slices:
            self._part_all.index_copy_(0, self._gidx, self._math_partials())

    def exchange_norm(self):
        import torch.distributed as dist
        if self._exchange():
            dist.all_reduce(self._part_all, group=self.group)       # x + 0 + ... + 0: exact

    def phase_update(self, max_norm=None):
        """Device: norm -> clip coefficient -> AdamW on the owned pieces; the updated values are copied into the bucket (the wire buffer)."""
        if max_norm is not None:
            self._math_coef(max_norm)
            self.last_norm = self._scal[2:3]
        if self._lr_host != self.lr:
            self._lr_pin[0] = self.lr
            self._scal[4:5].copy_(self._lr_pin, non_blocking=True)
            self._lr_host = self.lr
        if self._n_local:
            self._math_update(max_norm is not None)
            torch._foreach_copy_(self._wire_mine, self._param_mine)
        else:
            self._step_dev += 1                                   # (a rank that owns nothing still counts the step)

    def exchange_params(self):
        import torch.distributed as dist
        if not self._exchange():
            return
        for b in sorted(self.buckets):
            _, flat, cuts = self.buckets[b]
            if self._whole_slices(b):
                L = self.slice_elems(cuts[-1])
                dist.all_gather_into_tensor(flat[:self.world * L], flat[self.rank * L:(self.rank + 1) * L], group=self.group)
                continue
            for r in range(self.world):
                if cuts[r + 1] > cuts[r]:
                    dist.broadcast(flat[cuts[r]:cuts[r + 1]], src=self._global_rank(r), group=self.group)

    def phase_adopt(self):
        """Device: the slices other ranks own, from the bucket into the parameters; operand caches are stale from here."""
        if self._param_theirs:
            torch._foreach_copy_(self._param_theirs, self._wire_theirs)
        engine.bump_weight_epoch()
        ops.fp8_roll_all()

    def reduce_to_owners(self, b, wire_dtype=None, average_on_wire=False):
        """The gradient exchange of bucket b in sharded form: each slice is reduced to its owner only (ddp.GraphedStep calls this where the
        replicated step all-reduces the bucket).  Afterwards the bucket holds the mean in THIS rank's slice and undefined values elsewhere."""
        import torch.distributed as dist
        _, flat, cuts = self.buckets[b]
        lo, hi = cuts[self.rank], cuts[self.rank + 1]
        if not self._exchange():
            return
        op = dist.ReduceOp.AVG if average_on_wire else dist.ReduceOp.SUM
        if self._whole_slices(b):
            L = self.slice_elems(cuts[-1])
            buf = flat[:self.world * L] if wire_dtype is None else flat[:self.world * L].to(wire_dtype)
            mine = buf[self.rank * L:(self.rank + 1) * L]
            dist.reduce_scatter_tensor(mine, buf, op=op, group=self.group)
            if wire_dtype is not None and hi > lo:
                flat[lo:hi].copy_(mine[:hi - lo])
        else:
            n = cuts[-1]
            buf = flat[:n] if wire_dtype is None else flat[:n].to(wire_dtype)
            for r in range(self.world):
                if cuts[r + 1] > cuts[r]:
                    dist.reduce(buf[cuts[r]:cuts[r + 1]], dst=self._global_rank(r), op=op, group=self.group)
            if wire_dtype is not None and hi > lo:
                flat[lo:hi].copy_(buf[lo:hi])
        if hi > lo and not average_on_wire:
            flat[lo:hi].mul_(1.0 / self.world)

    def clip_and_step(self, max_norm=None):
        """The five phases back to back (eager).  The gradients must have gone through reduce_to_owners."""
        if not self._ready:
            if self.buckets and next(iter(self.buckets.values()))[1].is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ShardedAdamW: finalize() before capturing the optimizer phases")
            self.finalize()
        if max_norm is not None:
            self.phase_norm()
            self.exchange_norm()
        self.phase_update(max_norm)
        self.exchange_params()
        self.phase_adopt()
        return self.last_norm if max_norm is not None else None

    def owned_elements(self):
        return sum(n for (_, _, n, _, _, _) in self._mine), sum(c[-1] for _, _, c in self.buckets.values())

    # ---- checkpoint: the complete state in FusedAdamW's / torch.optim.AdamW's format, on every rank ------------------
    def _gather(self, owned):
        """{parameter: full tensor} of a per-piece quantity: through the buckets, like the parameters (overwrites the buckets: between steps only)."""
        if self._mine:
            torch._foreach_copy_(self._wire_mine, owned)
        self.exchange_params()
        out = {}
        for p, (b, st) in self._where.items():
            out[p] = self.buckets[b][1][st:st + p.numel()].view_as(p).clone()
        return out

    def state_dict(self):
        if not self._ready:
            self.finalize()
        m = self._gather([self._m[a:a + n] for (_, _, n, _, _, a) in self._mine])
        v = self._gather([self._v[a:a + n] for (_, _, n, _, _, a) in self._mine])
        step, state = self.step_count, {}
        for i, p in enumerate(self.params):
            if p in m:
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": m[p], "exp_avg_sq": v[p]}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        ready, self._tables = self._ready, None
        super().load_state_dict(sd)                               # validates, fills self.state with the complete moments
        full, self.state = dict(self.state), {}
        if not ready:
            self._full = full
            return
        self._tables = ()
        for (p, off, n, _, _, a) in self._mine:
            if p in full:
                self._m[a:a + n].copy_(full[p][0].reshape(-1)[off:off + n])
                self._v[a:a + n].copy_(full[p][1].reshape(-1)[off:off + n])
        self._step_dev.fill_(int(getattr(self, "_loaded_step", 0)))


def grad_norm(params):
    """Global L2 norm of the gradients as a device scalar (the value clip_grad_norm_ returns)."""
    act = [p for p in params if p.grad is not None]
    dev = act[0].device
    n = len(act)
    gp = torch.tensor([p.grad.data_ptr() for p in act], dtype=torch.int64).to(dev)
    sz = torch.tensor([p.grad.numel() for p in act], dtype=torch.int64).to(dev)
    part = torch.empty(lib().tav_sumsq_partials(n), dtype=torch.float32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    check(lib().tav_sumsq_multi(ptr(gp), ptr(sz), n, ptr(part), ptr(out), stream()), "sumsq_multi")
    return out.sqrt()
