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
        if max_norm is not None:
            part = torch.empty(nchunks, dtype=torch.float32, device=act[0].device)
            check(lib().tav_sumsq_chunked(ptr(d_g), ptr(d_s), ptr(d_c), n, nchunks, ptr(part), ptr(self._scal[0:1]), stream()), "sumsq_chunked")
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
