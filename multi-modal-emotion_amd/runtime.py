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
