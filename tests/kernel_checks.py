"""Per-kernel numerical checks of libtavhip against plain PyTorch references of the same op (run on the GPU).

Each check returns (name, max_abs_err, tolerance, ok).  Used by tests/test_kernels_gpu.py (asserts) and by
tools/gpu_kernel_check.py (prints a table without stopping at the first failure).
"""
import math

import torch
import torch.nn.functional as F

import tav_amd.ops as ops

DEV = "cuda"


def _rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def _res(name, got, ref, tol):
    got = got.float()
    ref = ref.float()
    err = (got - ref).abs().max().item()
    den = ref.abs().max().item() + 1e-12
    ok = bool(err <= tol * max(den, 1.0)) and bool(torch.isfinite(got).all())
    return (name, err / max(den, 1.0), tol, ok)


def tol_for(dtype):
    return 2e-2 if dtype == torch.bfloat16 else 2e-5


# ------------------------------------------------------------------------------------------------ GEMM
def check_gemm_nt(dtype, M=300, N=256, K=128, bias=True, act=0, resid=True, pre=False, out_f32=False, tile_m=0):
    a = _rnd(M, K, dtype=dtype, seed=1)
    b = _rnd(N, K, dtype=dtype, scale=0.1, seed=2)
    bi = _rnd(N, seed=3) if bias else None
    r = _rnd(M, N, seed=4) if resid else None
    out_dtype = torch.float32 if (out_f32 or dtype == torch.float32) else dtype
    res = ops.gemm_nt(a, b, bias=bi, act=act, resid=r, want_pre=pre, out_dtype=out_dtype, tile_m=tile_m)
    out, p = res if pre else (res, None)
    ref = a.float() @ b.float().t()
    if bias:
        ref = ref + bi
    ref_pre = ref
    if act == 1:
        ref = F.gelu(ref)
    if resid:
        ref = ref + r
    tol = tol_for(dtype if out_dtype != torch.float32 else torch.float32) if dtype == torch.float32 else (1e-2 if out_dtype == torch.bfloat16 else 2e-3)
    rs = [_res(f"gemm_nt[{dtype},M{M},N{N},K{K},b{int(bias)},a{act},r{int(resid)},f32out{int(out_f32)},tm{tile_m}]", out, ref, tol)]
    if pre:
        rs.append(_res("gemm_nt.pre", p, ref_pre, tol))
    return rs


def check_gemm_nt_mixed_schedule(M=46848 + 37, N=768, K=768):
    """The library may cover a launch's rows with whole rounds of 256 x 256 tiles plus 128-wide tiles over the rest (tav_gemm_nt_schedule).
    Every output element sums over K in the same order whatever the tile, so the mixed schedule must reproduce the single-tile launch
    (hint 17) BIT FOR BIT, for each epilogue flavour and every side tensor (bias, residual, second output, gelu' input)."""
    import ctypes as C
    dtype = torch.bfloat16
    a = _rnd(M, K, dtype=dtype, seed=41)
    b = _rnd(N, K, dtype=dtype, scale=0.1, seed=42)
    bi = _rnd(N, seed=43)
    r = _rnd(M, N, seed=44)
    u = _rnd(M, N, dtype=dtype, seed=45)
    g = ops.L.GemmNTArgs()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.in_dtype, g.out_dtype, g.nzb, g.nzg = M, N, K, K, K, N, 1, 1, 1, 1
    t, rf, tr = C.c_int32(), C.c_int32(), C.c_int32()
    assert ops.lib().tav_gemm_nt_schedule(C.byref(g), C.byref(t), C.byref(rf), C.byref(tr)) == 0
    split = 0 < rf.value < M and tr.value > 0
    rs = [(f"gemm_nt.mixed.plan[M{M},N{N},K{K}] is two launches", 0.0 if split else 1.0, 0.0, bool(split))]
    flavours = [("bias->bf16", dict(bias=bi)), ("bias+resid->f32", dict(bias=bi, resid=r, out_dtype=torch.float32)),
                ("gelu+pre", dict(bias=bi, act=3, want_pre=True)), ("*gelu'", dict(gelu_in=u, act=4)), ("->f32", dict(out_dtype=torch.float32))]
    for name, kw in flavours:
        mixed = ops.gemm_nt(a, b, tile_m=0, **kw)
        single = ops.gemm_nt(a, b, tile_m=17, **kw)
        mixed, single = (mixed if isinstance(mixed, tuple) else (mixed,)), (single if isinstance(single, tuple) else (single,))
        same = all(torch.equal(x, y) for x, y in zip(mixed, single))
        rs.append((f"gemm_nt.mixed[{name}] bitwise == single tile", 0.0 if same else 1.0, 0.0, bool(same)))
    ref = a.float() @ b.float().t() + bi
    rs.append(_res("gemm_nt.mixed[bias->bf16] vs torch", ops.gemm_nt(a, b, bias=bi), ref, 1e-2))
    return rs


def check_gemm_nt_gelu_bwd(dtype, M=200, N=384, K=256):
    a = _rnd(M, K, dtype=dtype, seed=5)
    b = _rnd(N, K, dtype=dtype, scale=0.1, seed=6)
    u = _rnd(M, N, dtype=dtype, seed=7)
    out = ops.gemm_nt(a, b, gelu_in=u)
    uu = u.float().requires_grad_(True)
    (gr,) = torch.autograd.grad(F.gelu(uu).sum(), uu)
    ref = (a.float() @ b.float().t()) * gr
    return [_res(f"gemm_nt.gelu_bwd[{dtype}]", out, ref, 1e-2 if dtype == torch.bfloat16 else 2e-5)]


def check_gemm_nt_gelu_derivative_pair(dtype, M=300, N=384, K=256, tile_m=0):
    """act 3 (forward FFN1: GELU out + gelu' stored in C_pre) and act 4 (dgrad: multiply by the stored derivative) against autograd."""
    a = _rnd(M, K, dtype=dtype, seed=11)
    b = _rnd(N, K, dtype=dtype, scale=0.1, seed=12)
    bi = _rnd(N, seed=13)
    h, gp = ops.gemm_nt(a, b, bias=bi, act=3, want_pre=True, tile_m=tile_m)
    pre = (a.float() @ b.float().t() + bi).requires_grad_(True)
    y = F.gelu(pre)
    (gr,) = torch.autograd.grad(y.sum(), pre)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5
    rs = [_res(f"gemm_nt.act3.gelu[{dtype},tm{tile_m}]", h, y.detach(), tol), _res(f"gemm_nt.act3.gelu'[{dtype},tm{tile_m}]", gp, gr, tol)]
    dy = _rnd(M, K, dtype=dtype, seed=14)
    w_t = _rnd(N, K, dtype=dtype, scale=0.1, seed=15)
    du = ops.gemm_nt(dy, w_t, gelu_in=gp, act=4, tile_m=tile_m)
    rs.append(_res(f"gemm_nt.act4[{dtype},tm{tile_m}]", du, (dy.float() @ w_t.float().t()) * gp.float(), tol))
    return rs


def check_fp8(M=700, N=384, K=1024, tile_m=0):
    """fp8 path (BASELINE config 5): per-tensor amax scaling, e4m3 quantisation (+ zero-padded transposed copy), the block-scaled MFMA GEMM with
    every epilogue flavour, and the weight gradient as the NT GEMM of the transposed copies.  References are computed from the DEQUANTISED
    operands, so the GEMM comparison is exact up to f32 summation order; the quantiser is compared with torch's own e4m3 conversion."""
    x = _rnd(M, K, dtype=torch.bfloat16, seed=21)
    w = _rnd(N, K, seed=22, scale=0.05)
    x8 = ops.fp8_quantize(x, want_t=True)
    w8 = ops.fp8_quantize(w)
    amax = x.float().abs().max()
    rs = [_res("fp8.amax", x8.scales[2:3], amax.reshape(1), 0.0), _res("fp8.scale", x8.scales[0:1], (448.0 / amax).reshape(1), 1e-6)]
    ref_q = (x.float() * x8.scales[0]).to(torch.float8_e4m3fn).float()
    rs.append(_res("fp8.quantize_vs_torch", x8.q.float(), ref_q, 0.0))
    # the row-major-only form takes the streaming kernel (16-B loads, no transposing tiles): same bytes; and so does its delayed-scaling form
    x8s = ops.fp8_quantize(x)
    rs.append(_res("fp8.quantize_stream == tiled", x8s.q.float(), x8.q.float(), 0.0))
    sts = ops.Fp8States(x.device, n=4)
    xa = ops.fp8_quantize(x, state=(sts, "x"))             # calibrates
    xb = ops.fp8_quantize(x, state=(sts, "x"))             # delayed: same scale, gathers the maximum
    rs.append(_res("fp8.quantize_stream delayed == calibrated", xb.q.float(), xa.q.float(), 0.0))
    rs.append(_res("fp8.delayed gathered amax", sts.dev[0, 3:4], amax.reshape(1), 0.0))
    xv = x[:, : K // 2]                                     # a strided view (row stride K, 16-B aligned): still the streaming kernel
    rs.append(_res("fp8.quantize_stream strided", ops.fp8_quantize(xv).q.float(), (xv.float() * (448.0 / xv.float().abs().max())).to(torch.float8_e4m3fn).float(), 0.0))
    kp = x8.qt.shape[1]
    rs.append(_res("fp8.quantize_t", x8.qt.float()[:, :M], x8.q.float().t(), 0.0))
    rs.append(_res("fp8.quantize_t_pad", x8.qt.float()[:, M:].abs().sum().reshape(1), torch.zeros(1, device=DEV), 0.0))
    xd, wd = x8.q.float() * x8.scales[1], w8.q.float() * w8.scales[1]
    bias = _rnd(N, seed=23)
    res = _rnd(M, N, seed=24)
    y = ops.gemm_nt_fp8(x8, w8, bias=bias, tile_m=tile_m)
    rs.append(_res(f"fp8.gemm_nt[bf16 out,tm{tile_m}]", y, xd @ wd.t() + bias, 1e-2))
    y32 = ops.gemm_nt_fp8(x8, w8, bias=bias, resid=res, out_dtype=torch.float32, tile_m=tile_m)
    rs.append(_res(f"fp8.gemm_nt[f32 out + resid,tm{tile_m}]", y32, xd @ wd.t() + bias + res, 2e-5))
    h, gp = ops.gemm_nt_fp8(x8, w8, bias=bias, act=3, want_pre=True, tile_m=tile_m)
    pre = (xd @ wd.t() + bias).requires_grad_(True)
    (gr,) = torch.autograd.grad(F.gelu(pre).sum(), pre)
    rs += [_res("fp8.gemm_nt.gelu", h, F.gelu(pre.detach()), 1e-2), _res("fp8.gemm_nt.gelu'", gp, gr, 1e-2)]
    # rough quantisation error of the whole product against the unquantised one (per-tensor e4m3, K = 1024): a sanity bound, not a parity claim
    rs.append(_res("fp8.gemm_nt.vs_unquantised", y32, x.float() @ w.t() + bias + res, 3e-2))
    dy = _rnd(M, N, dtype=torch.bfloat16, seed=25)
    dy8 = ops.fp8_quantize(dy, want_q=False, want_t=True)
    dw = ops.wgrad_fp8(dy8, x8)
    dyd, xdt = dy8.qt.float()[:, :M] * dy8.scales[1], x8.qt.float()[:, :M] * x8.scales[1]
    rs.append(_res("fp8.wgrad", dw, dyd @ xdt.t(), 1e-4))
    return rs


def check_gemm_tn(dtype, M=1000, N1=256, N2=384, nbatch=1):
    a = _rnd(nbatch * M, N1, dtype=dtype, seed=8)
    b = _rnd(nbatch * M, N2, dtype=dtype, seed=9)
    out, dbias = ops.gemm_tn(a, b, N1=N1, N2=N2, lda=N1, ldb=N2, rows_per_batch=M, nbatch=nbatch, a_zb=M * N1, b_zb=M * N2, want_bias=True)
    ref = a.float().t() @ b.float()
    return [_res(f"gemm_tn[{dtype},M{M},N1{N1},N2{N2},nb{nbatch}]", out, ref, 2e-3 if dtype == torch.bfloat16 else 2e-5),
            _res("gemm_tn.dbias", dbias, a.float().sum(0), 1e-5)]


def check_gemm_tn_grouped(dtype, M=777):
    """Four weight gradients over the same (ragged) token axis in one launch; operands are column slices of wider buffers."""
    shapes = [(384, 128), (128, 136), (256, 128), (128, 264)]
    wide_a = _rnd(M, 640, dtype=dtype, seed=21)
    pairs, refs = [], []
    for k, (n1, n2) in enumerate(shapes):
        a = wide_a[:, 64:64 + n1] if k == 0 else _rnd(M, n1, dtype=dtype, seed=22 + k)
        b = _rnd(M, n2, dtype=dtype, seed=30 + k)
        pairs.append((a, b))
        refs.append((a.float().t() @ b.float(), a.float().sum(0)))
    outs = ops.gemm_tn_grouped(pairs, want_bias=True)
    rs = []
    for k, ((dW, db), (rW, rb)) in enumerate(zip(outs, refs)):
        rs.append(_res(f"gemm_tn_grouped[{dtype},M{M}].dW{k}", dW, rW, 2e-3 if dtype == torch.bfloat16 else 2e-5))
        rs.append(_res(f"gemm_tn_grouped[{dtype},M{M}].db{k}", db, rb, 1e-5))
    return rs


def check_gemm_tn_grouped_big(M=5000, flags=1):
    """The 256-wide weight-gradient tile (bf16): ragged N1 / N2 (not multiples of 256), a ragged last K-tile, operands that are column slices
    of wider buffers, one to three token splits with the fixed-order slab reduce -- against f32 matmuls; and twice for bitwise reproducibility."""
    dtype = torch.bfloat16
    shapes = [(384, 136), (136, 264), (256, 512), (520, 128)]
    wide_a = _rnd(M, 640, dtype=dtype, seed=51)
    pairs, refs = [], []
    for k, (n1, n2) in enumerate(shapes):
        a = wide_a[:, 64:64 + n1] if k == 0 else _rnd(M, n1, dtype=dtype, seed=52 + k)
        b = _rnd(M, n2, dtype=dtype, seed=60 + k)
        pairs.append((a, b))
        refs.append((a.float().t() @ b.float(), a.float().sum(0)))
    rs = []
    for nsplit in (1, 2, 3):
        if (nsplit - 1) * (((M + nsplit - 1) // nsplit + 63) // 64 * 64) >= M:       # (the library refuses a split that would be empty)
            continue
        outs = ops.gemm_tn_grouped(pairs, want_bias=True, flags=flags | (nsplit << 8))
        again = ops.gemm_tn_grouped(pairs, want_bias=True, flags=flags | (nsplit << 8))
        for k, ((dW, db), (rW, rb)) in enumerate(zip(outs, refs)):
            rs.append(_res(f"gemm_tn_grouped.big[M{M},splits{nsplit}].dW{k}", dW, rW, 2e-3))
            rs.append(_res(f"gemm_tn_grouped.big[M{M},splits{nsplit}].db{k}", db, rb, 1e-5))
        same = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(outs, again))
        rs.append((f"gemm_tn_grouped.big[splits{nsplit}] bitwise repeatable", 0.0 if same else 1.0, 0.0, bool(same)))
    small = ops.gemm_tn_grouped(pairs, want_bias=True, flags=2)
    for k, ((dW, db), (rW, rb)) in enumerate(zip(small, refs)):
        rs.append(_res(f"gemm_tn_grouped.small(flag)[M{M}].dW{k}", dW, rW, 2e-3))
    return rs


def check_conv_as_gemm(dtype, B=2, T_in=203, Cc=64, k=3, s=2):
    """Conv1d(C,C,k,stride s) on channels-last activations as an NT GEMM over overlapping rows + its gradients."""
    T_out = (T_in - k) // s + 1
    x = _rnd(B, T_in, Cc, dtype=dtype, seed=10)
    w = _rnd(Cc, Cc, k, scale=0.1, seed=11)              # nn.Conv1d layout [co][ci][k], f32 parameter
    wn, wt, _ = ops.cast_conv_weight(w, dtype)
    y, pre = ops.gemm_nt(x, wn, act=1, want_pre=True, M=T_out, N=Cc, K=k * Cc, lda=s * Cc, ldb=k * Cc, ldc=Cc,
                         nzb=B, a_zb=T_in * Cc, c_zb=T_out * Cc, out_shape=(B, T_out, Cc))
    xr = x.float().permute(0, 2, 1).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    pre_ref = F.conv1d(xr, wr if dtype == torch.float32 else wr.to(dtype).float(), stride=s)
    y_ref = F.gelu(pre_ref)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5
    rs = [_res(f"conv_gemm.fwd[{dtype}]", y, y_ref.permute(0, 2, 1), tol)]
    dy = _rnd(B, T_out, Cc, dtype=dtype, seed=12)
    y_ref.backward(dy.float().permute(0, 2, 1))
    du = ops.gelu_bwd(pre, dy)
    dcol = ops.gemm_nt(du.view(B * T_out, Cc), wt, out_shape=(B * T_out, k * Cc))
    dx = ops.col2im_1d(dcol, B, T_in, T_out, Cc, k, s)
    rs.append(_res(f"conv_gemm.dx[{dtype}]", dx, xr.grad.permute(0, 2, 1), tol))
    dw = ops.gemm_tn(du, x, N1=Cc, N2=k * Cc, lda=Cc, ldb=s * Cc, rows_per_batch=T_out, nbatch=B, a_zb=T_out * Cc,
                     b_zb=T_in * Cc, perm_inner=Cc, perm_outer=k, out_shape=(Cc, Cc, k))
    rs.append(_res(f"conv_gemm.dw[{dtype}]", dw, wr.grad, 1e-2 if dtype == torch.bfloat16 else 2e-5))
    return rs


def check_conv_chain(dtype, B=3, T=407, Cc=64):
    """engine.ConvGemmFn as the wav2vec2 feature extractor chains it (HF wav2vec2:382-419, preset B: conv -> GELU, k = 3, 3, 2 at stride 2): the
    round-4 input gradient -- one NT GEMM per output phase over a zero-framed dy, gelu' in the epilogue, no column buffer / col2im -- against
    torch's conv1d autograd, and against the column-buffer path of rounds 1-3 (same products: bf16 differs only by where it rounds)."""
    from tav_amd import engine as E
    from tav_amd import runtime
    ectx = E.Ctx("fp32" if dtype == torch.float32 else "bf16")
    ks = [(3, 2), (3, 2), (2, 2)]
    ws = [torch.nn.Parameter(_rnd(Cc, Cc, k, scale=0.12, seed=300 + i)) for i, (k, _) in enumerate(ks)]
    bs = [torch.nn.Parameter(_rnd(Cc, scale=0.1, seed=310 + i)) for i in range(3)]
    x0 = _rnd(B, T, Cc, dtype=dtype, seed=320)

    def run(phase):
        E.CONV_PHASE_DGRAD[0] = phase
        for p_ in ws + bs:
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        h, u = x, None
        for (k, st), w, b in zip(ks, ws, bs):
            h, u = E.ConvGemmFn.apply(h, u, w, b, st, True, ectx)
        (h.float() * h.float()).sum().backward()
        return h.detach(), x.grad.detach(), [w.grad.detach().clone() for w in ws], [b.grad.detach().clone() for b in bs]
    try:
        y1, dx1, dw1, db1 = run(True)
        y0, dx0, dw0, db0 = run(False)
    finally:
        E.CONV_PHASE_DGRAD[0] = True
    xr = x0.float().permute(0, 2, 1).requires_grad_(True)
    wr = [w.detach().clone().requires_grad_(True) for w in ws]
    br = [b.detach().clone().requires_grad_(True) for b in bs]
    h = xr
    for (k, st), w, b in zip(ks, wr, br):
        h = F.gelu(F.conv1d(h, w if dtype == torch.float32 else w.to(dtype).float(), b, stride=st))
    (h * h).sum().backward()
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-5
    rs = [_res(f"conv_chain.y[{dtype}]", y1, h.permute(0, 2, 1), tol), _res(f"conv_chain.dx phase[{dtype}]", dx1, xr.grad.permute(0, 2, 1), tol),
          _res(f"conv_chain.dx col2im[{dtype}]", dx0, xr.grad.permute(0, 2, 1), tol)]
    for i in range(3):
        rs.append(_res(f"conv_chain.dw{i} phase[{dtype}]", dw1[i], wr[i].grad, tol))
        rs.append(_res(f"conv_chain.db{i} phase[{dtype}]", db1[i], br[i].grad, tol))
        rs.append(_res(f"conv_chain.dw{i} phase vs col2im[{dtype}]", dw1[i], dw0[i], tol))
    return rs


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, mask, mode, scale):
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    if mode == 1:
        s = s + mask[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    if mode == 2:
        p = p + mask[:, None, None, :]
    return torch.einsum("bhqk,bhkd->bhqd", p, v)


def check_attention(dtype, mode, B=2, S=200, nh=3, bwd=True, ref_style_mask=False, pre=False, spike=0.0):
    """ref_style_mask: the values PreFormer really produces (models/tav.py:383-397): {0,-65504} text, {65505,1} audio,
    {0} video.  The rank-1 term then dwarfs softmax(s)v, so the check is against an fp64 reference.
    pre: the q_prescaled convention of tav_attn_args (q holds q * scale * log2(e)); the reference sees the SAME rounded values divided
    by the factor, and dq is still the gradient w.r.t. the unscaled q.
    spike: scale factor on a few late (and one early) key rows, so a tile's maximum jumps past the running reference exponent by far
    more than the lazy-rescale threshold of the forward kernel -- the rare branch gets its own test (and an fp64 reference)."""
    H = nh * 64
    qkv = _rnd(B * S, 3 * H, dtype=dtype, seed=20 + mode)
    if spike:
        qkv = qkv.float()
        for row in {min(S - 1, 3), S // 2, max(0, S - 40), S - 1}:
            qkv.view(B, S, 3 * H)[:, row, H:2 * H] *= spike
        qkv = qkv.to(dtype)
    c2 = ops.ATTN_Q_PRESCALE
    if pre:
        qkv = qkv.clone()
        qkv[:, :H] = (qkv[:, :H].float() * c2).to(dtype)
    mask = None
    if mode == 1:
        mask = torch.zeros(B, S, device=DEV)
        mask[:, S - 37:] = torch.finfo(torch.float32).min
    if mode == 2:
        mask = torch.zeros(B, S, device=DEV)
        mask[:, : S // 3] = -0.5
        mask[:, S // 3: S // 2] = 2.0
        mask[0, S // 2:] = 1.0
        if ref_style_mask:
            mask = torch.zeros(B, S, device=DEV)
            mask[:, S // 4 - 9: S // 4] = -65504.0          # padded text tokens
            mask[:, S // 4: S // 4 + S // 2] = 65505.0      # valid audio frames
            mask[0, S // 4 + S // 2 - 20: S // 4 + S // 2] = 1.0   # padded audio frames of row 0
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    o, lse, corr = ops.attn_fwd(q, k, v, B, S, nh, key_mask=mask, mask_mode=mode, q_prescaled=pre)

    rdt = torch.float64 if (ref_style_mask or spike) else torch.float32

    def heads(t, div=1.0):
        return (t.to(rdt) / div).reshape(B, S, nh, 64).permute(0, 2, 1, 3).contiguous().requires_grad_(True)

    qr, kr, vr = heads(q, c2 if pre else 1.0), heads(k), heads(v)
    o_ref = _attn_ref(qr, kr, vr, mask.to(rdt) if mask is not None else None, mode, 0.125)
    tol = 2e-2 if dtype == torch.bfloat16 else 5e-5
    tag = f"{dtype},mode{mode},S{S},ref{int(ref_style_mask)},pre{int(pre)},spike{spike:g}"
    rs = [_res(f"attn.fwd[{tag}]", o, o_ref.permute(0, 2, 1, 3).reshape(B * S, H), tol)]
    lse_ref = torch.logsumexp(torch.einsum("bhqd,bhkd->bhqk", qr.detach(), kr.detach()) * 0.125 + (mask.to(rdt)[:, None, None, :] if mode == 1 else 0.0), dim=-1)
    rs.append(_res(f"attn.lse[{tag}]", lse, lse_ref, 2e-2 if dtype == torch.bfloat16 else 1e-5))
    if bwd:
        do = _rnd(B * S, H, dtype=dtype, seed=30 + mode)
        o_ref.backward(do.to(rdt).reshape(B, S, nh, 64).permute(0, 2, 1, 3))
        dqkv = ops.attn_bwd(q, k, v, o, do, lse, corr, B, S, nh, key_mask=mask, mask_mode=mode, q_prescaled=pre)

        def flat(t):
            return t.permute(0, 2, 1, 3).reshape(B * S, H)

        tolb = (6e-2 if mode == 2 else 3e-2) if dtype == torch.bfloat16 else 1e-4
        rs.append(_res(f"attn.dq[{tag}]", dqkv[:, :H], flat(qr.grad), tolb))
        rs.append(_res(f"attn.dk[{tag}]", dqkv[:, H:2 * H], flat(kr.grad), tolb))
        rs.append(_res(f"attn.dv[{tag}]", dqkv[:, 2 * H:], flat(vr.grad), tolb))
    return rs


# ------------------------------------------------------------------------------------------------ layer norm
def check_layernorm(x_dtype, W=768, rows=333, act=0):
    x = _rnd(rows, W, dtype=x_dtype, seed=40)
    gamma = 1.0 + 0.1 * _rnd(W, seed=41)
    beta = 0.1 * _rnd(W, seed=42)
    y32, ylp, mean, rstd = ops.ln_fwd(x, gamma, beta, 1e-5, want_f32=True, lp_dtype=torch.bfloat16, act=act)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (W,), gr, br, 1e-5)
    if act:
        ref = F.gelu(ref)
    rs = [_res(f"ln.fwd[{x_dtype},W{W},act{act}]", y32, ref, 2e-5), _res("ln.fwd.lp", ylp, ref, 1e-2)]
    dy = _rnd(rows, W, seed=43)
    add = _rnd(rows, W, seed=44)
    ref.backward(dy)
    dx32, dxlp, dg, db = ops.ln_bwd(dy, x, gamma, beta, mean, rstd, dx_add=add, want_f32=True, lp_dtype=torch.bfloat16, act=act)
    rs += [_res("ln.dx", dx32, xr.grad + add, 5e-5), _res("ln.dx.lp", dxlp, xr.grad + add, 1e-2),
           _res("ln.dgamma", dg, gr.grad, 1e-4), _res("ln.dbeta", db, br.grad, 1e-4)]
    return rs


def check_layernorm_deferred_param_reduce(n_ln=70):
    """ABI v5: inside an autograd backward pass ops.ln_bwd keeps the per-workgroup column sums and ONE tav_ln_param_reduce_multi launch per 64
    LayerNorms reduces them when the pass ends (ops.ln_flush, queued on the engine).  70 LayerNorms of three widths and ragged row counts
    (two launches): dgamma / dbeta must equal the immediate two-stage form bit for bit, and nothing may be pending afterwards."""
    cases = []
    for i in range(n_ln):
        W = (768, 512, 1024)[i % 3]
        rows = 37 + 61 * i
        x = _rnd(rows, W, seed=400 + i)
        gamma, beta = _rnd(W, seed=401 + i) * 0.1 + 1.0, _rnd(W, seed=402 + i) * 0.1
        _, _, mean, rstd = ops.ln_fwd(x, gamma, beta, 1e-5, want_f32=True, lp_dtype=None)
        dy = _rnd(rows, W, seed=403 + i)
        cases.append((x, gamma, beta, mean, rstd, dy))
    assert ops._graph_task_id() < 0
    want = [ops.ln_bwd(dy, x, g, b, m, r, want_f32=True)[2:] for (x, g, b, m, r, dy) in cases]       # outside a pass: immediate

    class Ln(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, g, b, i):
            ctx.i = i
            return x.clone()

        @staticmethod
        def backward(ctx, dy):
            x, g, b, m, r, _ = cases[ctx.i]
            dx, _, dg, db = ops.ln_bwd(dy.contiguous(), x, g, b, m, r, want_f32=True)
            return dx, dg, db, None

    leaves, total = [], None
    for i, (x, g, b, m, r, dy) in enumerate(cases):
        gp, bp = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        leaves.append((gp, bp))
        t = (Ln.apply(x.clone().requires_grad_(True), gp, bp, i) * dy).sum()
        total = t if total is None else total + t
    total.backward()
    torch.cuda.synchronize()
    assert not ops._ln_pending["items"]
    worst_g = max(float((gp.grad - w[0]).abs().max()) for (gp, _), w in zip(leaves, want))
    worst_b = max(float((bp.grad - w[1]).abs().max()) for (_, bp), w in zip(leaves, want))
    ok_g, ok_b = worst_g == 0.0, worst_b == 0.0
    return [(f"ln.deferred dgamma bitwise x{n_ln}", worst_g, 0.0, ok_g), (f"ln.deferred dbeta bitwise x{n_ln}", worst_b, 0.0, ok_b)]


# ------------------------------------------------------------------------------------------------ misc
def check_cast_weight():
    w = _rnd(300, 200, seed=50)
    n, t = ops.cast_weight(w, torch.bfloat16)
    return [_res("cast_weight.n", n, w.bfloat16(), 0.0), _res("cast_weight.t", t, w.bfloat16().t(), 0.0)]


def check_colsum():
    x = _rnd(1000, 768, dtype=torch.bfloat16, seed=51)
    return [_res("colsum", ops.colsum(x), x.float().sum(0), 1e-5)]


def check_text_embed(pad_id=1):
    B, S, W, V = 3, 50, 768, 1000
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, V, (B, S), generator=g)
    ids[:, S - 9:] = 1
    ids = ids.to(DEV)
    word, pos, typ = _rnd(V, W, seed=52), _rnd(S + 10, W, seed=53), _rnd(1, W, seed=54)
    gamma, beta = 1 + 0.1 * _rnd(W, seed=55), 0.1 * _rnd(W, seed=56)
    y32, _, pre, pos_ids, _, _ = ops.text_embed_fwd(ids, word, pos, typ, gamma, beta, 1e-5, pad_id)
    if pad_id >= 0:
        m = ids.ne(pad_id).int()
        pid = (torch.cumsum(m, 1) * m).long() + pad_id
    else:
        pid = torch.arange(S, device=DEV)[None].expand(B, S)
    ref = F.layer_norm(word[ids] + pos[pid] + typ[0], (W,), gamma, beta, 1e-5).reshape(B * S, W)
    return [_res(f"text_embed[pad{pad_id}]", y32, ref, 2e-5), _res("text_embed.pos_ids", pos_ids.float(), pid.float(), 0.0)]


def check_patchify(dtype=torch.float32):
    B, Fr, H, W, nkeep = 2, 4, 32, 48, 5
    video = _rnd(B, Fr, 3, H, W, seed=57)
    ntok = (Fr // 2) * (H // 16) * (W // 16)
    mask = torch.zeros(B, ntok, dtype=torch.bool)
    mask[0, [0, 3, 4, 7, 11]] = True
    mask[1, [1, 2, 5, 9, 10]] = True
    mask = mask.to(DEV)
    idx, counts = ops.mask_to_index(mask, True, nkeep)
    patches = ops.patchify(video, idx, dtype)
    wconv = _rnd(8, 3, 2, 16, 16, scale=0.05, seed=58)
    emb = F.conv3d(video.permute(0, 2, 1, 3, 4), wconv, stride=(2, 16, 16)).flatten(2).transpose(1, 2)   # [B, ntok, 8]
    ref = emb[mask].reshape(B * nkeep, 8)
    got = patches.float() @ wconv.reshape(8, -1).t()
    return [_res("patchify+gemm", got, ref, 1e-4), _res("mask_to_index.counts", counts.float(), torch.full((B,), float(nkeep), device=DEV), 0.0)]


def check_pool_head_ce():
    B, S, W = 3, 77, 768
    x = _rnd(B * S, W, seed=60)
    rs = [_res("mean_pool", ops.mean_pool_fwd(x, B, S), x.reshape(B, S, W).mean(1), 1e-5)]
    dy = _rnd(B, W, seed=61)
    dx, _ = ops.mean_pool_bwd(dy, B, S)
    rs.append(_res("mean_pool.bwd", dx, (dy / S)[:, None, :].expand(B, S, W).reshape(B * S, W), 1e-6))
    xh = _rnd(B, 3072, seed=62)
    Wh, bh = _rnd(7, 3072, scale=0.05, seed=63), _rnd(7, seed=64)
    y = ops.head_fwd(xh, Wh, bh)
    xr, Wr, br = xh.clone().requires_grad_(True), Wh.clone().requires_grad_(True), bh.clone().requires_grad_(True)
    yr = F.linear(xr, Wr, br)
    rs.append(_res("head.fwd", y, yr, 1e-5))
    tgt = torch.tensor([1, 6, 3], device=DEV)
    cw = torch.rand(7, device=DEV) + 0.5
    for weights in (None, cw):
        loss, dlog = ops.cross_entropy(y, tgt, weights)
        yl = y.clone().requires_grad_(True)
        lr = F.cross_entropy(yl, tgt, weight=weights)
        lr.backward()
        rs.append(_res(f"ce.loss[w{weights is not None}]", loss, lr.reshape(1), 1e-5))
        rs.append(_res(f"ce.dlogits[w{weights is not None}]", dlog, yl.grad, 1e-5))
    dyh = _rnd(B, 7, seed=65)
    yr.backward(dyh)
    dxh, dWh, dbh = ops.head_bwd(xh, Wh, dyh)
    rs += [_res("head.dx", dxh, xr.grad, 1e-5), _res("head.dW", dWh, Wr.grad, 1e-5), _res("head.db", dbh, br.grad, 1e-5)]
    t = _rnd(5, 768, seed=66)
    ty = ops.tanh_fwd(t)
    rs += [_res("tanh", ty, torch.tanh(t), 1e-6), _res("tanh.bwd", ops.tanh_bwd(ty, t), t * (1 - torch.tanh(t) ** 2), 1e-5)]
    return rs


def check_embed_add():
    rows, W = 500, 768
    x = _rnd(rows, W, seed=67)
    ids = torch.randint(0, 3, (rows,), generator=torch.Generator().manual_seed(1)).to(DEV)
    table = _rnd(3, W, seed=68)
    rs = [_res("embed_add.fwd", ops.embed_add_fwd(x, ids, table), x + table[ids], 1e-6)]
    dy = _rnd(rows, W, seed=69)
    ref = torch.zeros(3, W, device=DEV).index_add_(0, ids, dy)
    rs.append(_res("embed_add.bwd", ops.embed_add_bwd(dy, ids, 3), ref, 1e-5))
    rs.append(_res("scatter_add_rows", ops.scatter_add_rows(dy, ids, 3), ref, 1e-5))
    return rs


def check_scatter_deterministic():
    """The embedding-table gradient at the size of a real text batch: 4096 token rows, a padding id shared by a quarter of them, ids outside
    the table skipped; equal to index_add_ and bitwise identical from run to run (no atomics)."""
    rows, W, ntab = 4096, 768, 1000
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, ntab, (rows,), generator=g)
    ids[torch.rand(rows, generator=g) < 0.25] = 0
    ids[7] = 2
    ids = ids.to(DEV)
    dy = _rnd(rows, W, seed=91)
    ref = torch.zeros(ntab, W, device=DEV, dtype=torch.float64).index_add_(0, ids, dy.double()).float()
    a = ops.scatter_add_rows(dy, ids, ntab)
    b = ops.scatter_add_rows(dy, ids, ntab)
    bad = ids.clone()
    bad[5] = ntab + 3                                  # out-of-table index: skipped, never written out of bounds
    c = ops.scatter_add_rows(dy, bad, ntab)
    ref_c = torch.zeros(ntab, W, device=DEV, dtype=torch.float64).index_add_(0, ids[torch.arange(rows, device=DEV) != 5],
                                                                           dy.double()[torch.arange(rows, device=DEV) != 5]).float()
    return [_res("scatter_add_rows.large", a, ref, 2e-6), _res("scatter_add_rows.bitwise_repeat", (a != b).float().sum().reshape(1), torch.zeros(1, device=DEV), 0.0),
            _res("scatter_add_rows.bad_index_skipped", c, ref_c, 2e-6)]


def check_index_safety():
    """Rows of the video mask that keep FEWER tokens than nkeep (what the reference's collate can produce at batch > 1): keep_idx is still fully
    written with in-range indices, counts report the truth, and out-of-range indices handed to the gathers are clamped instead of faulting."""
    B, n, nkeep = 3, 96, 10
    mask = torch.zeros(B, n, dtype=torch.bool)
    mask[0, torch.arange(0, 40, 4)] = True              # exactly nkeep
    mask[1, [3, 50, 95]] = True                         # short row
    # row 2 keeps nothing
    idx, counts = ops.mask_to_index(mask.to(DEV), True, nkeep)
    exp = torch.zeros(B, nkeep)
    exp[0] = torch.arange(0, 40, 4).float()
    exp[1] = torch.tensor([3., 50., 95.] + [95.] * 7)
    rs = [_res("mask_to_index.padded", idx.float(), exp.to(DEV), 0.0), _res("mask_to_index.counts_short", counts.float(), torch.tensor([10., 3., 0.], device=DEV), 0.0)]
    table = _rnd(20, 64, seed=92)
    wild = torch.tensor([0, 19, -5, 20, 1 << 30], dtype=torch.int32, device=DEV)
    rs.append(_res("gather_rows.clamped", ops.gather_rows(table, wild), table[torch.tensor([0, 19, 0, 19, 19], device=DEV)], 0.0))
    video = _rnd(1, 2, 3, 32, 32, seed=93)
    ok = ops.patchify(video, torch.tensor([[3]], dtype=torch.int32, device=DEV), torch.float32)
    hi = ops.patchify(video, torch.tensor([[1 << 20]], dtype=torch.int32, device=DEV), torch.float32)
    rs.append(_res("patchify.clamped", hi, ok, 0.0))
    return rs


def check_conv0_gn(dtype):
    B, T_in, Cc, K, s = 2, 1605, 512, 10, 5
    T_out = (T_in - K) // s + 1
    wave = _rnd(B, T_in, scale=0.5, seed=70)
    w = _rnd(Cc, 1, K, scale=0.3, seed=71)
    gamma, beta = 1 + 0.1 * _rnd(Cc, seed=72), 0.1 * _rnd(Cc, seed=73)
    y0 = ops.conv0_fwd(wave, w, None, T_out, s, dtype)
    wr = w.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    c_ref = F.conv1d(wave[:, None], wr, stride=s)            # [B, C, T_out]
    tol = 1e-2 if dtype == torch.bfloat16 else 3e-5
    rs = [_res(f"conv0.fwd[{dtype}]", y0, c_ref.permute(0, 2, 1), tol)]
    y1, stats = ops.gn_gelu_fwd(y0, gamma, beta, 1e-5)
    c_in = y0.float().permute(0, 2, 1).detach().requires_grad_(True)
    g_ref = F.gelu(F.group_norm(c_in, Cc, gr, br, 1e-5))
    rs.append(_res(f"gn_gelu.fwd[{dtype}]", y1, g_ref.permute(0, 2, 1), tol))
    dy = _rnd(B, T_out, Cc, dtype=dtype, seed=74)
    g_ref.backward(dy.float().permute(0, 2, 1))
    dx, dg, db = ops.gn_gelu_bwd(y0, dy, gamma, beta, stats)
    tolb = 3e-2 if dtype == torch.bfloat16 else 2e-4
    rs += [_res(f"gn_gelu.dx[{dtype}]", dx, c_in.grad.permute(0, 2, 1), tolb), _res("gn_gelu.dgamma", dg, gr.grad, tolb),
           _res("gn_gelu.dbeta", db, br.grad, tolb)]
    c_ref.backward(dx.float().permute(0, 2, 1))
    dw, _ = ops.conv0_bwd_w(wave, dx, K, s, False)
    rs.append(_res(f"conv0.dw[{dtype}]", dw, wr.grad, 1e-4))
    return rs


def check_posconv(dtype):
    """grouped conv k=128, pad 64, drop last, GELU, + residual: forward and all gradients vs torch."""
    B, T, H, G, K = 2, 49, 256, 4, 128
    Cg = H // G
    x = _rnd(B, T, H, seed=80)                                   # f32 residual stream
    v = _rnd(H, Cg, K, scale=0.05, seed=81)
    g = (1.0 + 0.1 * _rnd(K, seed=82)).abs()
    bias = 0.1 * _rnd(H, seed=83)
    w, wf, norms = ops.weight_norm_fwd(v, g, dtype)
    xg = ops.group_pad(x.view(B * T, H), B, T, H, G, 64, 64, dtype)
    TP = T + 128
    y, pre = ops.gemm_nt(xg, w, bias=bias, act=1, resid=x.view(B * T, H), want_pre=True, out_dtype=torch.float32,
                         M=T, N=Cg, K=K * Cg, lda=Cg, ldb=K * Cg, ldc=H, nzb=B, nzg=G, a_zb=G * TP * Cg, a_zg=TP * Cg,
                         b_zg=Cg * K * Cg, c_zb=T * H, c_zg=Cg, bias_zg=Cg, out_shape=(B * T, H))
    xr = x.clone().requires_grad_(True)
    vr, gr_, br = v.clone().requires_grad_(True), g.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    wr = gr_[None, None, :] * vr / vr.pow(2).sum((0, 1), keepdim=True).sqrt()
    conv = F.conv1d(xr.permute(0, 2, 1), wr, br, padding=64, groups=G)[:, :, :-1]
    ref = xr + F.gelu(conv).permute(0, 2, 1)
    tol = 2e-2 if dtype == torch.bfloat16 else 5e-5
    rs = [_res(f"posconv.fwd[{dtype}]", y, ref.reshape(B * T, H), tol)]
    gout = _rnd(B * T, H, seed=84)
    ref.backward(gout.view(B, T, H))
    du = ops.gelu_bwd(pre, gout)                                  # f32 [B*T, H]
    dug = ops.group_pad(du, B, T, H, G, 63, 64, dtype)
    TPd = T + 127
    dx = ops.gemm_nt(dug, wf, resid=gout, out_dtype=torch.float32, M=T, N=Cg, K=K * Cg, lda=Cg, ldb=K * Cg, ldc=H, nzb=B, nzg=G,
                     a_zb=G * TPd * Cg, a_zg=TPd * Cg, b_zg=Cg * K * Cg, c_zb=T * H, c_zg=Cg, out_shape=(B * T, H))
    rs.append(_res(f"posconv.dx[{dtype}]", dx, xr.grad.reshape(B * T, H), tol))
    du_lp = ops.cast2d(du, dtype)
    dw = torch.empty(G, Cg, K * Cg, dtype=torch.float32, device=DEV)
    for gi in range(G):
        ops.gemm_tn(du_lp[:, gi * Cg:], xg[:, gi], out=dw[gi], N1=Cg, N2=K * Cg, lda=H, ldb=Cg, rows_per_batch=T, nbatch=B,
                    a_zb=T * H, b_zb=G * TP * Cg)
    dv, dg = ops.weight_norm_bwd(v, g, norms, dw)
    tolw = 3e-2 if dtype == torch.bfloat16 else 2e-4
    rs += [_res(f"posconv.dv[{dtype}]", dv, vr.grad, tolw), _res(f"posconv.dg[{dtype}]", dg, gr_.grad, tolw),
           _res(f"posconv.dbias[{dtype}]", ops.colsum(du), br.grad, tolw)]
    return rs


def all_checks():
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        out.append(lambda d=dtype: check_gemm_nt(d))
        out.append(lambda d=dtype: check_gemm_nt(d, M=1000, N=768, K=768, act=1, pre=True, resid=False))
        out.append(lambda d=dtype: check_gemm_nt(d, M=129, N=2304, K=768, resid=False))
        for tmh in (2, 3, 4):
            out.append(lambda d=dtype, t=tmh: check_gemm_nt(d, M=333, N=384, K=256, tile_m=t))
        if dtype == torch.bfloat16:          # 8-wave 256x128 tile (bf16 only): ragged M and N, short and long K, every epilogue
            out.append(lambda d=dtype: check_gemm_nt(d, M=333, N=384, K=256, tile_m=8))
            out.append(lambda d=dtype: check_gemm_nt(d, M=700, N=768, K=64, tile_m=8, out_f32=True))
            out.append(lambda d=dtype: check_gemm_nt(d, M=1000, N=3072, K=768, act=1, pre=True, resid=False, tile_m=8))
            out.append(lambda d=dtype: check_gemm_nt(d, M=513, N=132, K=1536, tile_m=8))
            # 8-wave 256x256 tile, two-pass epilogue: ragged M / N, short and long K, every epilogue flavour
            out.append(lambda d=dtype: check_gemm_nt(d, M=333, N=384, K=256, tile_m=16))
            out.append(lambda d=dtype: check_gemm_nt(d, M=700, N=768, K=64, tile_m=16, out_f32=True))
            out.append(lambda d=dtype: check_gemm_nt(d, M=1000, N=3072, K=768, act=1, pre=True, resid=False, tile_m=16))
            out.append(lambda d=dtype: check_gemm_nt(d, M=513, N=132, K=1536, tile_m=16))
            # interior tiles (the bounds-check-free pass) of the plain and residual flavours: no bias / bias, bf16 / f32 out, with and without the
            # f32 residual; the gelu pair below covers the other two flavours
            for kw in (dict(resid=False), dict(resid=False, bias=False), dict(resid=False, out_f32=True), dict(resid=False, bias=False, out_f32=True),
                       dict(out_f32=True), dict(bias=False, out_f32=True)):
                out.append(lambda d=dtype, k=kw: check_gemm_nt(d, M=700, N=768, K=192, tile_m=16, **k))
            # ... and every length of its 3 + 2 image ring's prologue / tail: 1, 2, 3 and 5 K-tiles
            for kk in (128, 192, 320):
                out.append(lambda d=dtype, k=kk: check_gemm_nt(d, M=300, N=260, K=k, tile_m=16))
        if dtype == torch.bfloat16:
            out.append(check_gemm_nt_mixed_schedule)
        if dtype == torch.bfloat16:
            out.append(check_gemm_tn_grouped_big)
            # the weight-gradient tile's ring with one, two and three K-tiles per split (the last one ragged)
            out.append(lambda: check_gemm_tn_grouped_big(M=130))
            out.append(lambda: check_gemm_tn_grouped_big(M=64))
        out.append(lambda d=dtype: check_gemm_nt_gelu_bwd(d))
        out.append(lambda d=dtype: check_gemm_nt_gelu_derivative_pair(d))
        if dtype == torch.bfloat16:
            out.append(lambda d=dtype: check_gemm_nt_gelu_derivative_pair(d, M=700, N=768, K=128, tile_m=16))
        out.append(lambda d=dtype: check_gemm_tn(d))
        out.append(lambda d=dtype: check_gemm_tn(d, M=249, N1=768, N2=512, nbatch=3))
        out.append(lambda d=dtype: check_gemm_tn_grouped(d))
        out.append(lambda d=dtype: check_gemm_tn_grouped(d, M=64))
        out.append(lambda d=dtype: check_conv_as_gemm(d))
        out.append(lambda d=dtype: check_conv_as_gemm(d, k=2, T_in=100))
        out.append(lambda d=dtype: check_conv_chain(d))
        out.append(lambda d=dtype: check_conv_chain(d, B=2, T=1000, Cc=128))      # even length: the phases end on different rows
        for mode in (0, 1, 2):
            out.append(lambda d=dtype, m=mode: check_attention(d, m))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=64, nh=1))
        out.append(lambda d=dtype: check_attention(d, 2, B=1, S=481, nh=12))
        out.append(lambda d=dtype: check_attention(d, 2, B=2, S=481, nh=12, ref_style_mask=True))
        # size extremes of the path: 1 token, the text model's longest sequence (514 positions -> 512 tokens, pre-softmax mask), the
        # 32-frame video of BASELINE config 5 (3136 tokens before masking), 10 s audio (499 frames)
        out.append(lambda d=dtype: check_attention(d, 0, B=2, S=1, nh=2))
        out.append(lambda d=dtype: check_attention(d, 1, B=1, S=512, nh=2))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=3136, nh=1))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=499, nh=2))
        # tav_attn_args.q_prescaled (what the engine's layers run): every mask mode, the ragged / single-tile / long cases, the reference-style
        # post-softmax mask, and key spikes that force the forward kernel's lazy rescale late in the key loop (moderate and extreme jumps)
        for mode in (0, 1, 2):
            out.append(lambda d=dtype, m=mode: check_attention(d, m, pre=True))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=64, nh=1, pre=True))
        out.append(lambda d=dtype: check_attention(d, 0, B=2, S=1, nh=2, pre=True))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=3136, nh=1, pre=True))
        out.append(lambda d=dtype: check_attention(d, 0, B=2, S=1464, nh=2, pre=True))
        # odd slice / tile counts: the one-dimensional XCD-banded grid (attn_tile) must stay a bijection when tiles * heads * batch % 8 != 0
        out.append(lambda d=dtype: check_attention(d, 0, B=3, S=300, nh=5, pre=True))
        out.append(lambda d=dtype: check_attention(d, 2, B=3, S=135, nh=3))
        out.append(lambda d=dtype: check_attention(d, 2, B=2, S=481, nh=12, ref_style_mask=True, pre=True))
        out.append(lambda d=dtype: check_attention(d, 0, B=2, S=328, nh=2, pre=True, spike=6.0))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=328, nh=2, pre=True, spike=300.0))
        out.append(lambda d=dtype: check_attention(d, 2, B=1, S=300, nh=2, pre=True, spike=40.0))
        out.append(lambda d=dtype: check_attention(d, 0, B=1, S=328, nh=2, spike=40.0))
        out.append(lambda d=dtype: check_layernorm(d))
        out.append(lambda d=dtype: check_layernorm(d, W=512, act=1))
        out.append(lambda d=dtype: check_layernorm(d, W=1024))
        if dtype == torch.float32:
            out.append(check_layernorm_deferred_param_reduce)
        out.append(lambda d=dtype: check_conv0_gn(d))
        out.append(lambda d=dtype: check_posconv(d))
    out.append(lambda: check_gemm_nt(torch.bfloat16, out_f32=True))
    out += [check_cast_weight, check_colsum, check_text_embed, lambda: check_text_embed(-1), check_patchify,
            check_pool_head_ce, check_embed_add, check_scatter_deterministic, check_index_safety, check_fp8,
            lambda: check_fp8(M=1000, N=768, K=256, tile_m=16), lambda: check_fp8(M=130, N=132, K=128, tile_m=4)]
    return out
