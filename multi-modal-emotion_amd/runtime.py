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
