"""Tensor-level wrappers over the C ABI: allocate outputs with torch, pass raw pointers + the current HIP stream.

No arithmetic happens here; torch is used for device memory and streams only.  Scratch buffers are cached per
(name, stream) so that concurrent branches on different HIP streams never share a workspace.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import check, dt, lib, ptr, stream

_ws = {}


def workspace(name, nfloats, device):
    key = (name, torch.cuda.current_stream().cuda_stream, str(device))
    t = _ws.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(int(nfloats), 1), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


def clear_workspaces():
    _ws.clear()


# Optional live timing of one kernel family with HIP events on the launch stream (bench.py's roofline pass only).
_prof = None


def profile_start(kinds):
    """kinds: one family name or several of "gemm_nt", "gemm_tn", "attn", "ln" (every launch of those families gets a HIP event pair)."""
    global _prof
    _prof = {"kinds": {kinds} if isinstance(kinds, str) else set(kinds), "ev": []}


def _prof_launch(family, select, work, nbytes, launch, work_exec=None):
    """Run `launch()`; while a profile of `family` is active, bracket it with a HIP event pair on the launch stream and record its
    algorithmic work (FLOPs for the MFMA-bound families, 0 for the HBM-bound one) and algorithmic bytes.  work_exec: the FLOPs the kernels
    really execute where that differs from the algorithmic count (the atomic-free attention backward recomputes two products)."""
    if _prof is not None and family in _prof["kinds"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        _prof["ev"].append((e0, e1, float(work), select, float(nbytes), family, float(work if work_exec is None else work_exec)))
    else:
        launch()


def profile_stop(select="bf16"):
    """-> (algorithmic FLOPs, seconds inside the kernels, launches) of the gemm_nt launches with `select` operands ("bf16" / "fp8") since
    profile_start(); profile_stop.families = {family: {"work", "bytes", "secs", "launches"}} for every profiled family (all operand types)."""
    global _prof
    torch.cuda.synchronize()
    allev, _prof = _prof["ev"], None
    fam = {}
    for e0, e1, work, sel, nbytes, family, work_exec in allev:
        d = fam.setdefault(family, {"work": 0.0, "work_exec": 0.0, "bytes": 0.0, "secs": 0.0, "launches": 0})
        d["work"] += work
        d["work_exec"] += work_exec
        d["bytes"] += nbytes
        d["secs"] += e0.elapsed_time(e1) * 1e-3
        d["launches"] += 1
    profile_stop.families = fam
    ev = [e for e in allev if e[5] == "gemm_nt" and e[3] == select]
    secs = sum(e[0].elapsed_time(e[1]) for e in ev) * 1e-3
    profile_stop.algorithmic_bytes = sum(e[4] for e in ev)          # operands + output once each (the minimum the launches could move)
    return sum(e[2] for e in ev), secs, len(ev)


# ---------------------------------------------------------------------------------------------- GEMM
def gemm_nt(a, b, *, bias=None, act=0, resid=None, gelu_in=None, out_dtype=None, want_pre=False, out=None,
            accumulate=False, alpha=1.0, M=None, N=None, K=None, lda=None, ldb=None, ldc=None,
            nzb=1, nzg=1, a_zb=0, a_zg=0, b_zb=0, b_zg=0, c_zb=0, c_zg=0, bias_zg=0, out_shape=None, tile_m=0, a_dequant=None, b_dequant=None,
            ld_gelu=None, gelu_zb=0):
    """C = epi(alpha * A @ B^T).  Plain use: a [M,K], b [N,K] contiguous.  Strided/batched use: pass sizes/strides.
    fp8 operands (torch.float8_e4m3fn): pass their dequantisation scalars (Fp8.dequant) and an explicit out_dtype."""
    if M is None:
        M, K = a.shape[-2], a.shape[-1]
        N = b.shape[-2]
        lda, ldb = a.stride(-2), b.stride(-2)
    out_dtype = out_dtype or (torch.bfloat16 if a.dtype == torch.float8_e4m3fn else a.dtype)
    if out is None:
        out = torch.empty(out_shape or (M, N), dtype=out_dtype, device=a.device)
    if ldc is None:
        ldc = out.stride(-2)
    pre = torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=out.device) if want_pre else None   # (same row stride as out: ld_pre = ldc)
    g = L.GemmNTArgs()
    g.A, g.B, g.C, g.C_pre = ptr(a), ptr(b), ptr(out), ptr(pre)
    g.bias, g.gelu_in, g.resid = ptr(bias), ptr(gelu_in), ptr(resid)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc, g.ld_pre = lda, ldb, ldc, ldc
    g.ld_gelu_in = (ld_gelu if ld_gelu is not None else gelu_in.stride(-2)) if gelu_in is not None else 0
    g.gelu_zb = gelu_zb
    g.ld_resid = resid.stride(-2) if resid is not None else 0
    g.nzb, g.nzg = nzb, nzg
    g.a_zb, g.a_zg, g.b_zb, g.b_zg, g.c_zb, g.c_zg, g.bias_zg = a_zb, a_zg, b_zb, b_zg, c_zb, c_zg, bias_zg
    g.in_dtype, g.out_dtype, g.act, g.accumulate, g.alpha = dt(a), dt(out), act, int(accumulate), alpha
    g.tile_m_hint = tile_m
    g.a_dequant, g.b_dequant = ptr(a_dequant), ptr(b_dequant)
    if bias is not None:
        assert bias.dtype == torch.float32
    if resid is not None:
        assert resid.dtype == torch.float32
    if gelu_in is not None:
        assert gelu_in.dtype == (torch.bfloat16 if a.dtype == torch.float8_e4m3fn else a.dtype)
    nz = max(nzb, 1) * max(nzg, 1)
    osz = 4 if out_dtype == torch.float32 else 2
    # operands, output and every epilogue side tensor once each: the minimum the launch could move
    nbytes = nz * ((M * K + N * K) * a.element_size() + M * N * (osz * (2 if want_pre else 1) + (4 if resid is not None else 0)
                                                                 + (2 if gelu_in is not None else 0) + (osz if accumulate else 0)))
    sel = "bf16" if a.dtype == torch.bfloat16 else ("fp8" if a.dtype == torch.float8_e4m3fn else "f32")
    _prof_launch("gemm_nt", sel, 2.0 * M * N * K * nz, nbytes, lambda: check(lib().tav_gemm_nt(C.byref(g), stream()), "gemm_nt"))
    return (out, pre) if want_pre else out


def gemm_tn(a, b, *, out=None, accumulate=False, N1=None, N2=None, lda=None, ldb=None, rows_per_batch=None, nbatch=1,
            a_zb=0, b_zb=0, perm_inner=0, perm_outer=0, scale=1.0, out_shape=None, want_bias=False):
    """out[n1][perm(n2)] (+)= sum_tokens A[t][n1] * B[t][n2]  (weight gradient).  want_bias: also return
    dbias[n1] = sum_tokens A[t][n1] computed inside the same kernel."""
    if N1 is None:
        N1, N2 = a.shape[-1], b.shape[-1]
        rows_per_batch = a.shape[-2]
        lda, ldb = a.stride(-2), b.stride(-2)
    cr, ns = C.c_int32(), C.c_int32()
    check(lib().tav_gemm_tn_splits(N1, N2, rows_per_batch, nbatch, C.byref(cr), C.byref(ns)), "gemm_tn_splits")
    slabs = workspace("tn_slabs", ns.value * N1 * N2, a.device)
    if out is None:
        out = torch.empty(out_shape or (N1, N2), dtype=torch.float32, device=a.device)
    g = L.GemmTNArgs()
    g.A, g.B, g.slabs, g.out = ptr(a), ptr(b), ptr(slabs), ptr(out)
    g.N1, g.N2, g.lda, g.ldb = N1, N2, lda, ldb
    g.rows_per_batch, g.nbatch, g.a_zb, g.b_zb = rows_per_batch, nbatch, a_zb, b_zb
    g.chunk_rows, g.nsplit, g.perm_inner, g.perm_outer = cr.value, ns.value, perm_inner, perm_outer
    g.dtype, g.accumulate, g.scale = dt(a), int(accumulate), scale
    dbias = None
    if want_bias:
        dbias = torch.empty(N1, dtype=torch.float32, device=a.device)
        bpart = workspace("tn_bias", ns.value * N1, a.device)
        g.dbias, g.bias_partials = ptr(dbias), ptr(bpart)
    assert a.dtype == b.dtype
    tokens = rows_per_batch * nbatch
    _prof_launch("gemm_tn", "bf16" if a.dtype == torch.bfloat16 else "f32", 2.0 * tokens * N1 * N2, tokens * (N1 + N2) * a.element_size() + 4 * N1 * N2,
                 lambda: check(lib().tav_gemm_tn(C.byref(g), stream()), "gemm_tn"))
    return (out, dbias) if want_bias else out


def gemm_tn_grouped(pairs, want_bias=True, flags=0, into=None):
    """[(A_k [rows, N1_k], B_k [rows, N2_k]), ...] (<= 4, same rows and dtype) -> [(dW_k [N1_k, N2_k] f32, db_k [N1_k] f32 | None), ...]
    in ONE launch (the weight gradients of one transformer layer): 128-wide tiles that each sum over all rows, or -- bf16, large problems --
    256-wide tiles with the rows split over workgroups into f32 slabs plus a fixed-order reduce (the library decides; `flags` as in tavhip.h).
    into: optional [(dW_buffer | None, db_buffer | None), ...] -- contiguous f32 destinations (the data-parallel gradient arena: the kernels
    write the bucket directly, ddp.GraphedStep); missing entries are allocated."""
    n = len(pairs)
    rows, dtype = pairs[0][0].shape[0], pairs[0][0].dtype
    probs = (L.GemmTNProblem * n)()
    outs = []
    for k, (a, b) in enumerate(pairs):
        assert a.shape[0] == rows and b.shape[0] == rows and a.dtype == dtype and b.dtype == dtype
        N1, N2 = a.shape[1], b.shape[1]
        dW_to, db_to = into[k] if into is not None else (None, None)
        dW = dW_to if dW_to is not None else torch.empty(N1, N2, dtype=torch.float32, device=a.device)
        db = (db_to if db_to is not None else torch.empty(N1, dtype=torch.float32, device=a.device)) if want_bias else None
        assert dW.shape == (N1, N2) and dW.is_contiguous() and dW.dtype == torch.float32 and (db is None or (db.shape == (N1,) and db.is_contiguous()))
        pr = probs[k]
        pr.A, pr.B, pr.out, pr.dbias = ptr(a), ptr(b), ptr(dW), ptr(db)
        pr.N1, pr.N2, pr.lda, pr.ldb = N1, N2, a.stride(0), b.stride(0)
        outs.append((dW, db))
    nbytes = lib().tav_gemm_tn_grouped_ws_bytes(probs, n, rows, dt(pairs[0][0]), flags)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pairs[0][0].device) if nbytes > 0 else None
    es = pairs[0][0].element_size()
    _prof_launch("gemm_tn", "bf16" if dtype == torch.bfloat16 else "f32", sum(2.0 * rows * a.shape[1] * b.shape[1] for a, b in pairs),
                 sum(rows * (a.shape[1] + b.shape[1]) * es + 4 * a.shape[1] * b.shape[1] for a, b in pairs),
                 lambda: check(lib().tav_gemm_tn_grouped_ws(probs, n, rows, dt(pairs[0][0]), ptr(ws), nbytes, flags, stream()), "gemm_tn_grouped"))
    return outs


# ---------------------------------------------------------------------------------------------- fp8 operands (BASELINE config 5)
FP8 = torch.float8_e4m3fn
FP8_KPAD = 1024        # the token axis of a transposed copy is padded to a multiple of this (8 K-splits of whole 128-byte K-tiles)


class Fp8:
    """A per-tensor-scaled e4m3 copy of a 2-D activation / gradient / weight: q [rows, cols] (the NT GEMM's A or B operand), optionally
    qt [cols, rows_pad] (the same values transposed, token axis zero padded: the operand of the weight-gradient GEMM), and the device
    scalars (scales[0] = 448/amax, scales[1] = amax/448 = `dequant`)."""
    __slots__ = ("q", "qt", "scales", "rows", "cols")

    def __init__(self, q, qt, scales, rows, cols):
        self.q, self.qt, self.scales, self.rows, self.cols = q, qt, scales, rows, cols

    @property
    def dequant(self):
        return self.scales[1:2]


# Delayed scaling (round 4): a tensor that is quantised every step at the same place of the graph keeps a 4-float STATE on the device
# (tav_fp8_quantize_delayed).  Its first quantisation calibrates (current scaling: amax passes + quantiser, what rounds 2-3 did every time);
# from then on ONE launch quantises with the scale the previous step left and gathers the new maximum on the way; fp8_roll_all() -- called by the
# optimizer step, inside the captured graph -- turns the gathered maxima into the next step's scales.  TAV_FP8_DELAYED=0: always current scaling.
fp8_delayed = [os.environ.get("TAV_FP8_DELAYED", "1") == "1"]
_fp8_state_sets = []


class Fp8States:
    """`n` states of 4 floats in one device tensor; slot(key) hands out (and remembers) the state of one quantisation site."""

    def __init__(self, device, n=8192):
        import weakref
        self.dev = torch.zeros(n, 4, dtype=torch.float32, device=device)
        self.index, self.calibrated = {}, set()
        _fp8_state_sets.append(weakref.ref(self))

    def slot(self, key):
        i = self.index.get(key)
        if i is None:
            i = len(self.index)
            if i >= self.dev.shape[0]:
                raise RuntimeError("Fp8States: out of slots")
            self.index[key] = i
        return i

    def roll(self):
        if self.index:
            check(lib().tav_fp8_roll_states(ptr(self.dev), len(self.index), stream()), "fp8_roll_states")


def fp8_roll_all():
    """End of a training step: every quantisation site's gathered maximum becomes its next scale (one tiny launch per state set)."""
    live = []
    for r in _fp8_state_sets:
        st = r()
        if st is not None:
            st.roll()
            live.append(r)
    _fp8_state_sets[:] = live


def fp8_quantize(x, *, want_q=True, want_t=False, state=None):
    """x [rows, cols] f32 / bf16 (row stride arbitrary) -> Fp8.  Without `state`: current scaling (two launches for the absolute maximum, one for
    the copies).  state = (Fp8States, key): delayed scaling once the site is calibrated (see above).  The scale stays on the device either way."""
    rows, cols = x.shape
    q = torch.empty(rows, cols, dtype=FP8, device=x.device) if want_q else None
    rows_pad = (rows + FP8_KPAD - 1) // FP8_KPAD * FP8_KPAD
    qt = torch.empty(cols, rows_pad, dtype=FP8, device=x.device) if want_t else None
    if state is not None:
        sts, key = state
        i = sts.slot(key)
        scales = sts.dev[i]
        if fp8_delayed[0] and i in sts.calibrated:
            check(lib().tav_fp8_quantize_delayed(ptr(x), dt(x), rows, cols, x.stride(0), ptr(scales), ptr(q), cols, ptr(qt), rows_pad, rows_pad, stream()),
                  "fp8_quantize_delayed")
            return Fp8(q, qt, scales, rows, cols)
        sts.calibrated.add(i)
    else:
        scales = torch.empty(3, dtype=torch.float32, device=x.device)
    part = workspace("fp8_amax", lib().tav_fp8_amax_partials(rows, cols), x.device)
    check(lib().tav_fp8_amax(ptr(x), dt(x), rows, cols, x.stride(0), ptr(part), ptr(scales), stream()), "fp8_amax")
    check(lib().tav_fp8_quantize(ptr(x), dt(x), rows, cols, x.stride(0), ptr(scales), ptr(q), cols, ptr(qt), rows_pad, rows_pad, stream()), "fp8_quantize")
    return Fp8(q, qt, scales, rows, cols)


def gemm_nt_fp8(a8, b8, **kw):
    """epi(A @ B^T) on two Fp8 operands (their .q copies)."""
    return gemm_nt(a8.q, b8.q, a_dequant=a8.dequant, b_dequant=b8.dequant, **kw)


def wgrad_fp8(dy8, x8, nsplit=8):
    """dW [N1, N2] f32 = dY^T X for two Fp8 tensors over the same token axis, as the NT GEMM of their transposed copies: M = N1, N = N2,
    K = padded tokens split into `nsplit` z-slices that write f32 slabs, summed in a fixed order by tav_splitk_reduce."""
    at, bt = dy8.qt, x8.qt
    N1, Kp = at.shape
    N2 = bt.shape[0]
    assert bt.shape[1] == Kp and Kp % (128 * nsplit) == 0
    kc = Kp // nsplit
    slabs = workspace("fp8_wgrad_slabs", nsplit * N1 * N2, at.device)
    gemm_nt(at, bt, out=slabs.view(-1)[:nsplit * N1 * N2].view(nsplit * N1, N2), out_dtype=torch.float32, M=N1, N=N2, K=kc, lda=Kp, ldb=Kp, ldc=N2,
            nzb=nsplit, a_zb=kc, b_zb=kc, c_zb=N1 * N2, a_dequant=dy8.dequant, b_dequant=x8.dequant)
    out = torch.empty(N1, N2, dtype=torch.float32, device=at.device)
    check(lib().tav_splitk_reduce(ptr(slabs), ptr(out), nsplit, N1 * N2, 0, stream()), "splitk_reduce")
    return out


def colsum(x, *, out=None, accumulate=False, M=None, N=None, ld=None):
    if M is None:
        M, N, ld = x.shape[-2], x.shape[-1], x.stride(-2)
    nparts = max(1, min(256, (M + 15) // 16))        # a workgroup column-sums >= 16 rows (hundreds of rows per workgroup at small M was latency bound)
    part = workspace("colsum", nparts * N, x.device)
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    check(lib().tav_colsum(ptr(x), dt(x), M, N, ld, ptr(part), nparts, ptr(out), int(accumulate), stream()), "colsum")
    return out


# ---------------------------------------------------------------------------------------------- attention
def _attn_args(q, k, v, o, B, S, nheads, key_mask, lse, corr, mask_mode, scale, o_soft=None, q_prescaled=False):
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o = ptr(q), ptr(k), ptr(v), ptr(o)
    a.key_mask, a.lse, a.corr, a.o_soft = ptr(key_mask), ptr(lse), ptr(corr), ptr(o_soft)
    a.B, a.S, a.nheads = B, S, nheads
    a.ld_q, a.ld_k, a.ld_v, a.ld_o = q.stride(-2), k.stride(-2), v.stride(-2), o.stride(-2)
    a.dtype, a.mask_mode, a.scale, a.q_prescaled = dt(q), mask_mode, scale, int(bool(q_prescaled))
    return a


ATTN_Q_PRESCALE = 0.125 * 1.4426950408889634      # what a prescaled q carries: softmax scale (head dim 64) x log2(e)


def attn_fwd(q, k, v, B, S, nheads, *, key_mask=None, mask_mode=0, scale=0.125, q_prescaled=False):
    """q,k,v: [B*S, >=nheads*64] views (row stride arbitrary).  Returns o [B*S, nheads*64], lse, corr.
    q_prescaled: q already holds q * scale * log2(e) (tav_attn_args.q_prescaled)."""
    H = nheads * 64
    o = torch.empty(B * S, H, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, nheads, S, dtype=torch.float32, device=q.device)
    corr = torch.empty(B, nheads, 64, dtype=torch.float32, device=q.device) if mask_mode == 2 else None
    o_soft = torch.empty_like(o) if mask_mode == 2 else None
    a = _attn_args(q, k, v, o, B, S, nheads, key_mask, lse, corr, mask_mode, scale, o_soft, q_prescaled)
    es = q.element_size()
    _prof_launch("attn", "bf16" if q.dtype == torch.bfloat16 else "f32", 4.0 * B * nheads * S * S * 64, 4 * B * S * H * es,      # Q, K, V, O once
                 lambda: check(lib().tav_attn_fwd(C.byref(a), stream()), "attn_fwd"))
    return o, lse, (corr, o_soft)


def attn_bwd(q, k, v, o, dout, lse, corr, B, S, nheads, *, key_mask=None, mask_mode=0, scale=0.125, dqkv=None, q_prescaled=False):
    """Returns dqkv [B*S, 3H] (dq | dk | dv), the layout the fused QKV projection's backward consumes."""
    H = nheads * 64
    if dqkv is None:
        dqkv = torch.empty(B * S, 3 * H, dtype=q.dtype, device=q.device)
    dq, dk, dv = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]
    delta = workspace("attn_delta", B * nheads * S, q.device)
    corr, o_soft = corr if corr is not None else (None, None)
    a = _attn_args(q, k, v, o, B, S, nheads, key_mask, lse, corr, mask_mode, scale, o_soft, q_prescaled)
    a.dout, a.dq, a.dk, a.dv, a.delta = ptr(dout), ptr(dq), ptr(dk), ptr(dv), ptr(delta)
    a.ld_do, a.ld_dq, a.ld_dk, a.ld_dv = dout.stride(-2), dq.stride(-2), dk.stride(-2), dv.stride(-2)
    es = q.element_size()
    # algorithmic work by the contract (BASELINE.md section 2 / SURVEY.md section 8d: fwd + bwd = 3 x fwd, so the backward is 2 x 4 B h S^2 d); the two
    # atomic-free kernels EXECUTE 14 B h S^2 d (S and dP are recomputed in both), reported beside it as `frac_executed`
    _prof_launch("attn", "bf16" if q.dtype == torch.bfloat16 else "f32", 8.0 * B * nheads * S * S * 64, 8 * B * S * H * es,     # Q, K, V, O, dO in; dQ, dK, dV out
                 lambda: check(lib().tav_attn_bwd(C.byref(a), stream()), "attn_bwd"), work_exec=14.0 * B * nheads * S * S * 64)
    return dqkv


def attn_probs(q, k, lse, B, S, nheads, *, key_mask=None, mask_mode=0, scale=0.125, q_prescaled=False, head_scale=None):
    """Attention probabilities as a tensor, f32 [B, nheads, S, S] (slow path: VideoMAEEncoder(output_attentions=True), reference
    utils/TAVFormer.py:362-375, :389), from q, k and the lse of attn_fwd.  head_scale: f32 [nheads] or [B, nheads] (head_mask) or None."""
    probs = torch.empty(B, nheads, S, S, dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, q, q, B, S, nheads, key_mask, lse, None, mask_mode, scale, None, q_prescaled)
    bstride = 0 if head_scale is None or head_scale.numel() == nheads else nheads
    check(lib().tav_attn_probs(C.byref(a), ptr(probs), ptr(head_scale), bstride, stream()), "attn_probs")
    return probs


def head_scale(a, b, hs, c0, B, S, nheads, out=None):
    """out[r, h*64+d] = (a or 0) + (c0 + hs[b, h]) * b[r, h*64+d] for token-major [B*S, nheads*64] tensors (row stride arbitrary); hs f32
    [nheads] / [B, nheads] or None.  The head_mask arithmetic of the fusion encoder's slow path (reference utils/TAVFormer.py:368-370)."""
    if out is None:
        out = torch.empty(B * S, nheads * 64, dtype=b.dtype, device=b.device)
    bstride = 0 if hs is None or hs.numel() == nheads else nheads
    check(lib().tav_head_scale(ptr(a), ptr(b), ptr(out), dt(b), ptr(hs), bstride, float(c0), B, S, nheads,
                               a.stride(-2) if a is not None else 0, b.stride(-2), out.stride(-2), stream()), "head_scale")
    return out


# ---------------------------------------------------------------------------------------------- layer norm
def ln_fwd(x, gamma, beta, eps, *, want_f32=True, lp_dtype=None, act=0):
    """x [rows, W] f32 or bf16.  Returns (y_f32 | None, y_lp | None, mean, rstd)."""
    rows, W = x.shape
    y32 = torch.empty(rows, W, dtype=torch.float32, device=x.device) if want_f32 else None
    ylp = torch.empty(rows, W, dtype=lp_dtype, device=x.device) if lp_dtype is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    a = L.LnArgs()
    a.x, a.x_dtype, a.gamma, a.beta = ptr(x), dt(x), ptr(gamma), ptr(beta)
    a.y_f32, a.y_lp, a.lp_dtype = ptr(y32), ptr(ylp), dt(lp_dtype) if lp_dtype is not None else 0
    a.mean, a.rstd = ptr(mean), ptr(rstd)
    a.rows, a.W, a.ld_x, a.ld_y, a.eps, a.act = rows, W, x.stride(0), W, eps, act
    nb = rows * W * (x.element_size() + (4 if want_f32 else 0) + (ylp.element_size() if ylp is not None else 0)) + rows * 8
    _prof_launch("ln", "hbm", 0.0, nb, lambda: check(lib().tav_ln_fwd(C.byref(a), stream()), "ln_fwd"))
    return y32, ylp, mean, rstd


# Deferred second stage of the LayerNorm backward.  Inside an autograd backward pass nobody reads dgamma / dbeta before the pass ends (they
# flow straight into AccumulateGrad), so every LayerNorm keeps its own partial slab and ONE launch per 64 of them reduces the lot from a
# callback the engine runs when the pass is over -- on the caller's stream, after it has been synchronised with every stream a gradient was
# produced on (torch/csrc/autograd/engine.cpp, GraphTask::exec_post_processing), hence also under multi-stream capture.  ~210 launches per step
# become 4.  Off (immediate reduce) outside a backward pass and while a hook-mode reducer is active (ddp.BucketedAllReduce reads gradients
# from post-accumulate hooks, i.e. before the pass ends).
# What autograd does with a returned gradient decides the rest (torch/csrc/autograd/functions/accumulate_grad.h):
#   * a parameter WITHOUT a gradient takes it over -- `grad = new_grad.detach()`, same storage -- provided nobody else holds the tensor: the
#     pending list therefore keeps the STORAGES of dgamma / dbeta alive, never the tensors (a second reference would make AccumulateGrad
#     clone the still unwritten buffer);
#   * a parameter that already HAS a gradient is accumulated in place, at once: it reads the buffer -- so a LayerNorm whose gamma / beta
#     carry a gradient (accumulation over several backwards, zero-filled gradients) is reduced immediately, not deferred.
ln_defer = [os.environ.get("TAV_LN_DEFER", "1") == "1"]      # (TAV_LN_DEFER=0: the immediate two-launch form, for A/B runs)
_ln_pending = {"task": -1, "items": []}


def _graph_task_id():
    try:
        return torch._C._current_graph_task_id()
    except Exception:
        return -1


def ln_flush():
    """Reduce every pending LayerNorm's partial slab into its dgamma / dbeta (current stream)."""
    items, _ln_pending["items"] = _ln_pending["items"], []
    _ln_pending["task"] = -1
    cur = torch.cuda.current_stream() if items and items[0][0].is_cuda else None
    for i in range(0, len(items), L.LN_REDUCE_MAX):
        chunk = items[i:i + L.LN_REDUCE_MAX]
        arr = (L.LnReduceItem * len(chunk))()
        for j, (part, dg_st, db_st, dg_ptr, db_ptr, nblk, W) in enumerate(chunk):
            if cur is not None:              # allocated under a branch stream, used here: the allocator must not recycle them before this launch ran
                part.record_stream(cur)
                for st in (dg_st, db_st):
                    torch.empty(0, dtype=torch.float32, device=part.device).set_(st).record_stream(cur)
            arr[j].partials, arr[j].dgamma, arr[j].dbeta, arr[j].nblocks, arr[j].W, arr[j].accumulate = ptr(part), dg_ptr, db_ptr, nblk, W, 0
        check(lib().tav_ln_param_reduce_multi(arr, len(chunk), stream()), "ln_param_reduce_multi")


def _ln_defer_register(part, dgamma, dbeta, nblk, W):
    tid = _graph_task_id()
    item = (part, dgamma.untyped_storage(), dbeta.untyped_storage(), dgamma.data_ptr(), dbeta.data_ptr(), nblk, W)
    if _ln_pending["task"] != tid:
        # a new backward pass (whatever an aborted earlier one left behind is dropped: its gradients were never consumed)
        _ln_pending["task"], _ln_pending["items"] = tid, []
        try:
            torch.autograd.Variable._execution_engine.queue_callback(ln_flush)
        except Exception:                       # (a torch without this private hook: reduce at once and stop deferring)
            ln_defer[0] = False
            _ln_pending["items"] = [item]
            ln_flush()
            return
    _ln_pending["items"].append(item)


def _grad_free_leaf(p):
    return bool(getattr(p, "is_leaf", False)) and p.grad is None


def ln_bwd(dy, x, gamma, beta, mean, rstd, *, dx_add=None, want_f32=True, lp_dtype=None, act=0, param_grads=True):
    """Returns (dx_f32 | None, dx_lp | None, dgamma, dbeta)."""
    rows, W = x.shape
    dx32 = torch.empty(rows, W, dtype=torch.float32, device=x.device) if want_f32 else None
    dxlp = torch.empty(rows, W, dtype=lp_dtype, device=x.device) if lp_dtype is not None else None
    dgamma = dbeta = part = None
    defer = False
    if param_grads:
        dgamma = torch.empty(W, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(W, dtype=torch.float32, device=x.device)
        nblk = lib().tav_ln_bwd_partials(rows)
        defer = ln_defer[0] and _graph_task_id() >= 0 and _grad_free_leaf(gamma) and _grad_free_leaf(beta)
        if defer:
            part = torch.empty(nblk * 2 * W, dtype=torch.float32, device=x.device)      # its own slab: alive until ln_flush()
        else:
            part = workspace("ln_part", nblk * 2 * W, x.device)
    a = L.LnArgs()
    a.x, a.x_dtype, a.gamma, a.beta = ptr(x), dt(x), ptr(gamma), ptr(beta)
    a.mean, a.rstd = ptr(mean), ptr(rstd)
    a.dy, a.dy_dtype, a.dx_add = ptr(dy), dt(dy), ptr(dx_add)
    a.dx_f32, a.dx_lp, a.lp_dtype = ptr(dx32), ptr(dxlp), dt(lp_dtype) if lp_dtype is not None else 0
    a.dgamma, a.dbeta, a.partials, a.accumulate_params = ptr(dgamma), ptr(dbeta), ptr(part), 0
    a.defer_param_reduce = int(defer)
    a.rows, a.W, a.ld_x, a.ld_dy, a.ld_dx, a.act = rows, W, x.stride(0), dy.stride(0), W, act
    nb = rows * W * (x.element_size() + dy.element_size() + (4 if dx_add is not None else 0) + (4 if want_f32 else 0)
                     + (dxlp.element_size() if dxlp is not None else 0)) + rows * 8
    _prof_launch("ln", "hbm", 0.0, nb, lambda: check(lib().tav_ln_bwd(C.byref(a), stream()), "ln_bwd"))
    if defer:
        _ln_defer_register(part, dgamma, dbeta, nblk, W)
    return dx32, dxlp, dgamma, dbeta


# ---------------------------------------------------------------------------------------------- casts & small ops
def cast_weight(w, dtype, *, want_t=True, want_n=True, out_n=None, out_t=None):
    """f32 [R,C] parameter -> (copy in dtype [R,C], transposed copy [C,R]).  out_n / out_t may be (strided) views of a
    larger fused operand (row stride taken from the view)."""
    R, Cc = w.shape
    n = out_n if out_n is not None else (torch.empty(R, Cc, dtype=dtype, device=w.device) if want_n else None)
    t = out_t if out_t is not None else (torch.empty(Cc, R, dtype=dtype, device=w.device) if want_t else None)
    check(lib().tav_cast_weight(ptr(w), R, Cc, ptr(n), n.stride(0) if n is not None else 0, ptr(t), t.stride(0) if t is not None else 0,
                                dt(dtype), stream()), "cast_weight")
    return n, t


def make_cast_descs(entries, device):
    """entries: list of (src f32 [R,C], dst|None, dst_t|None[, scale_n]).  scale_n multiplies the values written to dst only (the transposed
    copy stays unscaled): the attention pre-scale of the q rows.  Returns a device uint8 tensor holding the C descriptor table."""
    import struct
    buf = bytearray()
    for ent in entries:
        src, dst, dst_t = ent[:3]
        scale_n = float(ent[3]) if len(ent) > 3 and ent[3] is not None else 1.0
        R, Cc = src.shape
        ref = dst if dst is not None else dst_t
        buf += struct.pack("<QQQqqiiif", src.data_ptr(), dst.data_ptr() if dst is not None else 0, dst_t.data_ptr() if dst_t is not None else 0,
                           dst.stride(0) if dst is not None else Cc, dst_t.stride(0) if dst_t is not None else R, R, Cc, dt(ref), scale_n)
    tiles = max(((e[0].shape[0] + 31) // 32) * ((e[0].shape[1] + 31) // 32) for e in entries)
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone().to(device), tiles


def cast_weights_multi(descs, n, blocks_per_tensor=96):
    check(lib().tav_cast_weights_multi(ptr(descs), n, blocks_per_tensor, stream()), "cast_weights_multi")


def cast_conv_weight(w, dtype, conv_stride=0):
    """-> (forward operand [co, k*ci], column-buffer dgrad operand [k*ci, co], per-phase dgrad operands flat [co*ci*k] or None).
    conv_stride > 0 asks for the phase operands (tav_cast_conv_weight, include/tavhip.h)."""
    co, ci, k = w.shape
    n = torch.empty(co, k * ci, dtype=dtype, device=w.device)
    t = torch.empty(k * ci, co, dtype=dtype, device=w.device)
    ph = torch.empty(co * ci * k, dtype=dtype, device=w.device) if conv_stride and conv_stride <= k else None
    check(lib().tav_cast_conv_weight(ptr(w), co, ci, k, ptr(n), ptr(t), ptr(ph), conv_stride if ph is not None else 0, dt(dtype), stream()), "cast_conv_weight")
    return n, t, ph


def zero_pad_rows(buf, B, T, Cc, pad):
    """Zero the `pad` leading and trailing rows of each of the B entries of a [B, T + 2*pad, Cc] buffer."""
    check(lib().tav_zero_pad_rows(ptr(buf), dt(buf), B, T, Cc, pad, stream()), "zero_pad_rows")
    return buf


def cast2d(x, dtype, out=None):
    R, Cc = x.shape
    if out is None:
        out = torch.empty(R, Cc, dtype=dtype, device=x.device)
    check(lib().tav_cast2d(ptr(x), dt(x), x.stride(0), ptr(out), dt(out), out.stride(0), R, Cc, stream()), "cast2d")
    return out


def transpose2d(x, R, Cc, nbatch):
    """x: [nbatch*R, Cc] -> per batch transposed [nbatch*Cc, R] (returned with the same 2-D shape as x: the reference re-views it)."""
    out = torch.empty_like(x)
    check(lib().tav_transpose2d(ptr(x), ptr(out), dt(x), R, Cc, nbatch, stream()), "transpose2d")
    return out


def add_f32(a, b, *, want_f32=True, lp_dtype=None):
    y = torch.empty_like(a) if want_f32 else None
    ylp = torch.empty(a.shape, dtype=lp_dtype, device=a.device) if lp_dtype is not None else None
    check(lib().tav_add_f32(ptr(a), ptr(b), ptr(y), ptr(ylp), dt(lp_dtype) if lp_dtype is not None else 0, a.numel(), stream()), "add_f32")
    return y, ylp


def zeros_f32(shape, device):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib().tav_fill_f32(ptr(t), 0.0, t.numel(), stream()), "fill")
    return t


def embed_add_fwd(x, ids, table):
    rows, W = x.shape
    out = torch.empty_like(x)
    check(lib().tav_embed_add_fwd(ptr(x), ptr(ids), ptr(table), ptr(out), rows, W, table.shape[0], stream()), "embed_add_fwd")
    return out


def embed_add_bwd(dy, ids, ntable):
    rows, W = dy.shape
    parts = lib().tav_embed_add_bwd_parts(rows)
    part = workspace("embed_part", parts * ntable * W, dy.device)
    d = torch.empty(ntable, W, dtype=torch.float32, device=dy.device)
    check(lib().tav_embed_add_bwd(ptr(dy), ptr(ids), ptr(d), ptr(part), rows, W, ntable, 0, stream()), "embed_add_bwd")
    return d


def text_embed_fwd(ids, word, pos, type_, gamma, beta, eps, pad_id, *, want_f32=True, lp_dtype=None):
    B, S = ids.shape
    W = word.shape[1]
    dev = ids.device
    pre = torch.empty(B * S, W, dtype=torch.float32, device=dev)
    pos_ids = torch.empty(B, S, dtype=torch.int64, device=dev)
    y32 = torch.empty(B * S, W, dtype=torch.float32, device=dev) if want_f32 else None
    ylp = torch.empty(B * S, W, dtype=lp_dtype, device=dev) if lp_dtype is not None else None
    mean = torch.empty(B * S, dtype=torch.float32, device=dev)
    rstd = torch.empty(B * S, dtype=torch.float32, device=dev)
    a = L.TextEmbedArgs()
    a.ids, a.word, a.pos, a.type, a.gamma, a.beta = ptr(ids), ptr(word), ptr(pos), ptr(type_), ptr(gamma), ptr(beta)
    a.pre, a.pos_ids, a.y_f32, a.y_lp = ptr(pre), ptr(pos_ids), ptr(y32), ptr(ylp)
    a.lp_dtype = dt(lp_dtype) if lp_dtype is not None else 0
    a.mean, a.rstd = ptr(mean), ptr(rstd)
    a.B, a.S, a.W, a.vocab, a.max_pos, a.pad_id, a.eps = B, S, W, word.shape[0], pos.shape[0], pad_id, eps
    check(lib().tav_text_embed_fwd(C.byref(a), stream()), "text_embed_fwd")
    return y32, ylp, pre, pos_ids, mean, rstd


def scatter_add_rows(d, idx, ntable):
    rows, W = d.shape
    out = zeros_f32((ntable, W), d.device)
    check(lib().tav_scatter_add_rows(ptr(d), ptr(idx), ptr(out), rows, W, ntable, stream()), "scatter_add_rows")
    return out


def gather_rows(table, idx):
    rows, W = idx.numel(), table.shape[1]
    out = torch.empty(rows, W, dtype=torch.float32, device=table.device)
    check(lib().tav_gather_rows(ptr(table), ptr(idx), ptr(out), rows, W, table.shape[0], stream()), "gather_rows")
    return out


def mask_to_index(mask_bool, keep_value, nkeep):
    B, n = mask_bool.shape
    m8 = mask_bool.view(torch.uint8) if mask_bool.dtype == torch.bool else mask_bool
    idx = torch.empty(B, nkeep, dtype=torch.int32, device=mask_bool.device)
    counts = torch.empty(B, dtype=torch.int32, device=mask_bool.device)
    check(lib().tav_mask_to_index(ptr(m8), int(keep_value), ptr(idx), ptr(counts), B, n, nkeep, stream()), "mask_to_index")
    return idx, counts


def patchify(video, keep_idx, dtype):
    B, F, Cc, H, W = video.shape
    assert Cc == 3 and video.dtype == torch.float32 and video.is_contiguous()
    nkeep = keep_idx.shape[1]
    out = torch.empty(B * nkeep, 1536, dtype=dtype, device=video.device)
    check(lib().tav_patchify(ptr(video), ptr(keep_idx), ptr(out), dt(dtype), B, F, H, W, nkeep, stream()), "patchify")
    return out


def mean_pool_fwd(x, B, S):
    W = x.shape[-1]
    y = torch.empty(B, W, dtype=torch.float32, device=x.device)
    check(lib().tav_mean_pool_fwd(ptr(x), ptr(y), B, S, W, stream()), "mean_pool_fwd")
    return y


def mean_pool_bwd(dy, B, S, *, want_f32=True, lp_dtype=None):
    W = dy.shape[-1]
    dx = torch.empty(B * S, W, dtype=torch.float32, device=dy.device) if want_f32 else None
    dxlp = torch.empty(B * S, W, dtype=lp_dtype, device=dy.device) if lp_dtype is not None else None
    check(lib().tav_mean_pool_bwd(ptr(dy), ptr(dx), ptr(dxlp), dt(lp_dtype) if lp_dtype is not None else 0, B, S, W, stream()), "mean_pool_bwd")
    return dx, dxlp


def head_fwd(x, W, b):
    B, K = x.shape
    N = W.shape[0]
    y = torch.empty(B, N, dtype=torch.float32, device=x.device)
    check(lib().tav_head_fwd(ptr(x), ptr(W), ptr(b), ptr(y), B, K, N, stream()), "head_fwd")
    return y


def head_bwd(x, W, dy, need_dx=True, need_db=True):
    B, K = x.shape
    N = W.shape[0]
    dx = torch.empty(B, K, dtype=torch.float32, device=x.device) if need_dx else None
    dW = torch.empty(N, K, dtype=torch.float32, device=x.device)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if need_db else None
    check(lib().tav_head_bwd(ptr(x), ptr(W), ptr(dy), ptr(dx), ptr(dW), ptr(db), B, K, N, 0, stream()), "head_bwd")
    return dx, dW, db


def tanh_fwd(x):
    y = torch.empty_like(x)
    check(lib().tav_tanh_fwd(ptr(x), ptr(y), x.numel(), stream()), "tanh_fwd")
    return y


def tanh_bwd(y, dy):
    dx = torch.empty_like(y)
    check(lib().tav_tanh_bwd(ptr(y), ptr(dy), ptr(dx), y.numel(), stream()), "tanh_bwd")
    return dx


def cross_entropy(logits, target, class_weight=None, grad_scale=1.0, want_grad=True):
    B, Cn = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dlog = torch.empty_like(logits) if want_grad else None
    check(lib().tav_cross_entropy(ptr(logits), ptr(target), ptr(class_weight), ptr(loss), ptr(dlog), B, Cn, grad_scale, stream()), "cross_entropy")
    return loss, dlog


def dropout_fwd(x, p, seed, offset):
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib().tav_dropout_fwd(ptr(x), ptr(y), ptr(mask), x.numel(), p, seed, offset, stream()), "dropout_fwd")
    return y, mask


def dropout_bwd(dy, mask, p):
    dx = torch.empty_like(dy)
    check(lib().tav_dropout_bwd(ptr(dy), ptr(mask), ptr(dx), dy.numel(), p, stream()), "dropout_bwd")
    return dx


# ---------------------------------------------------------------------------------------------- audio front-end
def conv0_fwd(wave, w, bias, T_out, stride, dtype):
    B, T_in = wave.shape
    Cc, K = w.shape[0], w.shape[-1]
    y = torch.empty(B, T_out, Cc, dtype=dtype, device=wave.device)
    check(lib().tav_conv0_fwd(ptr(wave), ptr(w), ptr(bias), ptr(y), dt(dtype), B, T_in, T_out, Cc, K, stride, stream()), "conv0_fwd")
    return y


def conv0_bwd_w(wave, dy, K, stride, want_bias):
    B, T_in = wave.shape
    _, T_out, Cc = dy.shape
    part = workspace("conv0_part", lib().tav_conv0_bwd_partials(B, T_out, Cc, K), wave.device)
    dw = torch.empty(Cc, 1, K, dtype=torch.float32, device=wave.device)
    db = torch.empty(Cc, dtype=torch.float32, device=wave.device) if want_bias else None
    check(lib().tav_conv0_bwd_w(ptr(wave), ptr(dy), dt(dy), ptr(dw), ptr(db), ptr(part), B, T_in, T_out, Cc, K, stride, 0, stream()), "conv0_bwd_w")
    return dw, db


def gn_gelu_fwd(x, gamma, beta, eps):
    B, T, Cc = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(B, Cc, 2, dtype=torch.float32, device=x.device)
    wsb = workspace("gn_ws", lib().tav_gn_workspace_floats(B, Cc), x.device)
    check(lib().tav_gn_gelu_fwd(ptr(x), ptr(y), dt(x), ptr(gamma), ptr(beta), ptr(stats), ptr(wsb), B, T, Cc, eps, stream()), "gn_gelu_fwd")
    return y, stats


def gn_gelu_bwd(x, dy, gamma, beta, stats):
    B, T, Cc = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(Cc, dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    wsb = workspace("gn_ws", lib().tav_gn_workspace_floats(B, Cc), x.device)
    check(lib().tav_gn_gelu_bwd(ptr(x), ptr(dy), ptr(dx), dt(x), ptr(gamma), ptr(beta), ptr(stats), ptr(wsb), ptr(dg), ptr(db), B, T, Cc, 0, stream()), "gn_gelu_bwd")
    return dx, dg, db


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(lib().tav_gelu_bwd(ptr(x), ptr(dy), ptr(dx), dt(x), x.numel(), stream()), "gelu_bwd")
    return dx


def col2im_1d(dcol, B, T_in, T_out, Cc, K, stride, pre_act=None):
    dx = torch.empty(B, T_in, Cc, dtype=dcol.dtype, device=dcol.device)
    check(lib().tav_col2im_1d(ptr(dcol), ptr(dx), ptr(pre_act), dt(dcol), B, T_in, T_out, Cc, K, stride, stream()), "col2im_1d")
    return dx


def group_pad(x, B, T, H, G, pad_l, pad_r, dtype):
    xg = torch.empty(B, G, pad_l + T + pad_r, H // G, dtype=dtype, device=x.device)
    check(lib().tav_group_pad(ptr(x), dt(x), ptr(xg), dt(dtype), B, T, H, G, pad_l, pad_r, stream()), "group_pad")
    return xg


def weight_norm_fwd(v, g, dtype):
    H, Cg, K = v.shape
    norms = torch.empty(K, dtype=torch.float32, device=v.device)
    part = workspace("wn_part", lib().tav_weight_norm_partials(H, Cg) * K, v.device)
    G = H // Cg
    w = torch.empty(G, Cg, K * Cg, dtype=dtype, device=v.device)
    wf = torch.empty(G, Cg, K * Cg, dtype=dtype, device=v.device)
    check(lib().tav_weight_norm_fwd(ptr(v), ptr(g), ptr(norms), ptr(part), ptr(w), ptr(wf), dt(dtype), H, Cg, K, stream()), "weight_norm_fwd")
    return w, wf, norms


def weight_norm_bwd(v, g, norms, dw_gemm):
    H, Cg, K = v.shape
    wsb = workspace("wn_bwd", H * Cg * K + lib().tav_weight_norm_partials(H, Cg) * K + K, v.device)
    dv = torch.empty_like(v)
    dg = torch.empty(K, dtype=torch.float32, device=v.device)
    check(lib().tav_weight_norm_bwd(ptr(v), ptr(g), ptr(norms), ptr(dw_gemm), ptr(wsb), ptr(dv), ptr(dg), H, Cg, K, 0, stream()), "weight_norm_bwd")
    return dv, dg
