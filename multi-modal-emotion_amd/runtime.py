"""Process-wide execution context: precision policy + weight-operand cache (one per process = one per GPU)."""
from . import engine

_ctx = None


def set_precision(name="bf16"):
    """'bf16' (bf16 operands / f32 accumulate, the benchmarked mode) or 'fp32' (exact-f32 MFMA, parity mode)."""
    global _ctx
    _ctx = engine.Ctx(name)
    return _ctx


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = engine.Ctx("bf16")
    return _ctx


def precision():
    return ctx().pol.name


# ---- branch streams: text / audio / video encoders run beside the fusion encoder (reference models/tav.py:476-487 are
# four independent sub-graphs that meet only at the concat, :495).  Autograd replays each branch's backward on the stream its
# forward used, so the backward overlaps the same way; under hipGraph capture the fork/join becomes parallel graph branches.
_streams = {}
_inputs_event = [None]
multistream = [True]


import os as _os

front_side = [_os.environ.get("TAV_FRONT_STREAMS", "0") == "1"]     # PreFormer front-ends on side streams (measured: slower, off)
_prio = _os.environ.get("TAV_STREAM_PRIO", "0") == "1"     # measured: priorities break the overlap under graph replay (55 vs 37.7 ms)


def branch_streams(n=3):
    """(audio, video, text) streams; the video branch is the critical path, so it gets the high priority."""
    import torch
    dev = torch.cuda.current_device()
    if dev not in _streams:
        pr = [0, -1, 0] if _prio else [0, 0, 0]
        _streams[dev] = [torch.cuda.Stream(device=dev, priority=pr[i]) for i in range(n)]
    _register_branches(_streams[dev])
    return _streams[dev]


_front = {}


def front_streams(n=2):
    import torch
    dev = torch.cuda.current_device()
    if dev not in _front:
        _front[dev] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    _register_branches(_front[dev])
    return _front[dev]


# ---- capture guard.  Round 3 lost a run to a core dump (`hipStreamEndCapture` inside torch/cuda/graphs.py capture_end, ROCm 7.2): while a
# stream capture is in progress, a dependency edge between a BRANCH stream and any stream other than the capture's ORIGIN -- branch to
# branch, or to a stream the capture never forked -- builds a graph the runtime crashes on when the capture ends.  The legal topology is a
# star: a branch forks from the origin (waits for an event recorded there) and is joined by the origin (origin waits for the branch).
# Every cross-stream dependency of this package goes through stream_wait(); captures go through capture(); while one is active an illegal
# edge raises RuntimeError naming the two streams instead of producing that graph.  Pure host logic (stream identity only): tested on the CPU.
_capture = None          # {"origin": stream, "branches": [streams handed out or forked while the capture is active]}


def _same(a, b):
    return a is b or a == b


def _register_branches(streams):
    if _capture is not None:
        for st in streams:
            if not _same(st, _capture["origin"]) and not any(_same(st, b) for b in _capture["branches"]):
                _capture["branches"].append(st)


def capture_active():
    return _capture is not None


def check_edge(waiter, producer):
    """Raise if `waiter` waiting for `producer` is an edge the capture in progress cannot hold (no-op outside a capture)."""
    if _capture is None or _same(waiter, producer):
        return
    origin, branches = _capture["origin"], _capture["branches"]

    def known(st):
        return _same(st, origin) or any(_same(st, b) for b in branches)
    for st in (waiter, producer):
        if not known(st):
            raise RuntimeError(f"hipGraph capture on {origin}: stream {st} is neither the capture's origin nor one of its registered branches "
                               f"(runtime.branch_streams / front_streams / fork); a dependency on it ({waiter} waits for {producer}) would crash hipStreamEndCapture")
    if not _same(waiter, origin) and not _same(producer, origin):
        raise RuntimeError(f"hipGraph capture on {origin}: dependency between two branch streams ({waiter} waits for {producer}); under capture every "
                           f"edge must start or end at the origin stream (ROCm 7.2 crashes in hipStreamEndCapture otherwise) -- join through the origin")


def stream_wait(waiter, producer, event=None):
    """`waiter` waits for everything `producer` has been given so far (or for `event`, which was recorded on `producer`).  The one place
    cross-stream dependencies are made, so the capture guard sees them all."""
    check_edge(waiter, producer)
    if event is not None:
        waiter.wait_event(event)
    else:
        waiter.wait_stream(producer)


class capture:
    """`with runtime.capture(graph, stream, **kw):` = torch.cuda.graph(graph, stream=stream, **kw) with the guard above armed: `stream` is the
    origin; streams handed out by branch_streams() / front_streams() inside the block (or passed as `branches=`) are its branches."""

    def __init__(self, graph, stream, branches=(), **kw):
        import torch
        self._cm = torch.cuda.graph(graph, stream=stream, **kw)
        self._origin, self._branches = stream, list(branches)

    def __enter__(self):
        global _capture
        if _capture is not None:
            raise RuntimeError("runtime.capture: a capture is already in progress in this process")
        _capture = {"origin": self._origin, "branches": list(self._branches)}
        try:
            return self._cm.__enter__()
        except BaseException:
            _capture = None
            raise

    def __exit__(self, *exc):
        global _capture
        try:
            return self._cm.__exit__(*exc)
        finally:
            _capture = None


class guard_only:
    """The guard without a hipGraph (CPU tests of the logic; `origin` / branches are any objects with identity)."""

    def __init__(self, origin, branches=()):
        self._origin, self._branches = origin, list(branches)

    def __enter__(self):
        global _capture
        _capture = {"origin": self._origin, "branches": list(self._branches)}

    def __exit__(self, *exc):
        global _capture
        _capture = None


def share_with(stream, *tensors):
    """Tensors allocated under the caller's stream (e.g. a batch just shipped to the device) that a branch stream will read, also in its
    backward: tell the caching allocator, or the block is handed out again on the caller's stream the moment the last reference dies on
    the HOST -- while the branch's kernels that read it may not have run yet (found as a wrong conv0 weight gradient at full size)."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)


def mark_inputs_ready():
    """Record 'the batch is resident' on the current stream (called at the start of PreFormer.forward): the encoder branches of
    TAVForMAE wait for this event only, not for PreFormer's own kernels."""
    import torch
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _inputs_event[0] = ev
    return ev


def take_inputs_event():
    ev, _inputs_event[0] = _inputs_event[0], None
    return ev


# ---- backward segments (ddp.GraphedStep): while a recorder is active every encoder stack reports the residual-stream tensor that
# enters chosen layers; the data-parallel step cuts its backward there so that the gradients of the upper layers can travel over xGMI
# while the lower layers are still being differentiated.
_cut_rec = None
CUT_FRACTIONS = (1.0 / 12.0, 1.0 / 3.0, 2.0 / 3.0)      # of a stack's depth: the lowest segment (front-ends, embeddings, first layer) stays light


def begin_cuts(fractions=CUT_FRACTIONS):
    global _cut_rec
    _cut_rec = {"fractions": tuple(fractions), "pts": {}}


def end_cuts():
    """-> {branch: [(x, x_cut) in forward order]}: x belongs to the graph below the cut, x_cut (a detached leaf) starts the graph above."""
    global _cut_rec
    rec, _cut_rec = _cut_rec, None
    return rec["pts"] if rec is not None else {}


def cut_point(branch, i, L, x):
    """Called by an encoder stack before its layer i (of L) with the f32 residual stream entering that layer; returns the tensor the
    stack continues with.  Without a recorder that is x itself.  With one, at the chosen depths, it is a DETACHED leaf: the autograd
    graph is physically cut there (the engine cannot be told to stop at an interior node -- it walks through it whenever a requested
    leaf is also reachable underneath), and ddp.SegmentedBackward chains the pieces by handing x_cut's gradient to x."""
    if _cut_rec is None or i == 0 or not x.requires_grad:
        return x
    marks = sorted({min(L - 1, max(1, int(round(f * L)))) for f in _cut_rec["fractions"]})
    if i not in marks:
        return x
    xc = x.detach().requires_grad_(True)
    _cut_rec["pts"].setdefault(branch, []).append((x, xc))
    return xc


# ---- gradient arena (ddp.GraphedStep): while a data-parallel step is being captured, the weight-gradient kernels of a transformer layer
# write straight into the bucket that will be all-reduced, so that bucket needs no pack copy.  `grad_slots` maps parameter.data_ptr() -> its f32
# view inside the bucket (the backward sees the parameters as unpacked saved tensors: other Python objects, same storage); `layer_groups` collects, during the forward, the parameter tuples whose gradients one grouped launch produces
# (wq, wk, wv share ONE fused [3H, K] output, so their views must be adjacent and in that order).
grad_slots = {}
layer_groups = None


def begin_layer_groups():
    global layer_groups
    layer_groups = []


def end_layer_groups():
    global layer_groups
    g, layer_groups = layer_groups, None
    return g or []


def note_layer_group(wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2):
    if layer_groups is not None:
        layer_groups.append((wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2))


def fused_slot(params):
    """The arena views of `params` as ONE contiguous tensor [sum of rows, ...] if every one has a slot and they are adjacent in this order."""
    views = [grad_slots.get(p.data_ptr()) if p is not None else None for p in params]
    if any(v is None for v in views):
        return None
    ptr = views[0].data_ptr()
    for v in views:
        if v.data_ptr() != ptr:
            return None
        ptr += v.numel() * 4
    base = views[0]
    rows = sum(v.shape[0] for v in views)
    return base.as_strided((rows,) + tuple(base.shape[1:]), base.stride())
