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
    return _streams[dev]


_front = {}


def front_streams(n=2):
    import torch
    dev = torch.cuda.current_device()
    if dev not in _front:
        _front[dev] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _front[dev]


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
