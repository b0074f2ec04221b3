"""Focused, low-noise timings for A/B of two builds on ONE box (TAV_LIB selects the library; see tools/ab_build.sh).
usage: python tools/gpu_ab.py [attn] [gemm] [tn]   -- prints min / median over 7 repetitions of 50 launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402

import tav_amd.ops as ops  # noqa: E402

dev = "cuda"
what = set(sys.argv[1:]) or {"attn", "gemm", "tn"}
B = int(os.environ.get("TAV_B", "8"))


def timeit(fn, iters=50, reps=7):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def rnd(*s, dtype=torch.bfloat16):
    return torch.randn(*s, device=dev).to(dtype)


tag = os.path.basename(os.environ.get("TAV_LIB", "libtavhip.so"))
if "attn" in what:
    for pre in (False, True):         # q_prescaled: the convention the engine's layers use (tav_attn_args.q_prescaled)
        for (name, S, mode) in [("video", 1464, 0), ("fusion", 481, 2), ("audio", 249, 0), ("text", 128, 1)] + ([("video-L", 2927, 0)] if os.environ.get("TAV_ATTN_LONG") else []):
            nh, H = 12, 768
            qkv = rnd(B * S, 3 * H)
            mask = torch.zeros(B, S, device=dev) if mode else None
            q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
            lo, med = timeit(lambda: ops.attn_fwd(q, k, v, B, S, nh, key_mask=mask, mask_mode=mode, q_prescaled=pre))
            fl = 4 * B * nh * S * S * 64
            print(f"[{tag}] attn_fwd pre{int(pre)} {name:7s} S={S:5d}: min {lo:8.1f} us  med {med:8.1f} us  {fl / lo / 1e6:7.1f} TF")
            o, lse, aux = ops.attn_fwd(q, k, v, B, S, nh, key_mask=mask, mask_mode=mode, q_prescaled=pre)
            do = rnd(B * S, H)
            lo, med = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, aux if mode == 2 else None, B, S, nh, key_mask=mask, mask_mode=mode, q_prescaled=pre))
            print(f"[{tag}] attn_bwd pre{int(pre)} {name:7s} S={S:5d}: min {lo:8.1f} us  med {med:8.1f} us  {2.0 * fl / lo / 1e6:7.1f} TF (8 B h S^2 d; executes 14)")
if "gemm" in what:
    for (name, M, N, K) in [("video qkv", B * 1464, 2304, 768), ("video out", B * 1464, 768, 768), ("video ffn1", B * 1464, 3072, 768),
                            ("video ffn2", B * 1464, 768, 3072), ("fusion qkv", B * 481, 2304, 768), ("audio ffn1", B * 249, 3072, 768),
                            ("conv1", B * 7999, 512, 1536), ("square 4096", 4096, 4096, 4096)]:
        a, b = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        lo, med = timeit(lambda: ops.gemm_nt(a, b, bias=bias))
        print(f"[{tag}] gemm_nt {name:12s} M={M:6d} N={N:5d} K={K:5d}: min {lo:8.1f} us  med {med:8.1f} us  {2 * M * N * K / lo / 1e6:7.1f} TF")
if "layer" in what:       # the eight NT GEMMs of ONE video-encoder layer (forward + dgrad) with their real epilogues, and their sum
    M, H, F = B * int(os.environ.get("TAV_S", "1464")), 768, 3072     # TAV_S: tokens per utterance (1464 video, 481 fusion, 249 audio, 128 text)
    x_lp, w_qkv, w_o, w_1, w_2 = rnd(M, H), rnd(3 * H, H), rnd(H, H), rnd(F, H), rnd(H, F)
    w_qkv_t, w_o_t, w_1_t, w_2_t = rnd(H, 3 * H), rnd(H, H), rnd(H, F), rnd(F, H)
    b3, b1, bf = torch.randn(3 * H, device=dev), torch.randn(H, device=dev), torch.randn(F, device=dev)
    res = torch.randn(M, H, device=dev)
    pad = int(os.environ.get("TAV_PAD", "0"))             # experiment: row stride of the wide [M, 3072] / [M, 2304] activations = width + pad elements

    def wide(n, dtype=torch.bfloat16):
        return rnd(M, n + pad, dtype=dtype)[:, :n] if pad else rnd(M, n, dtype=dtype)

    h_lp, u_lp, dqkv = wide(F), wide(F), wide(3 * H)
    o_qkv, o_h, o_du = wide(3 * H), wide(F), wide(F)
    tmh = int(os.environ.get("TAV_TM", "0"))
    cases = [("qkv      bias            -> bf16", lambda: ops.gemm_nt(x_lp, w_qkv, bias=b3, out=o_qkv, tile_m=tmh), 2 * M * 3 * H * H),
             ("out-proj bias+resid      -> f32 ", lambda: ops.gemm_nt(x_lp, w_o, bias=b1, resid=res, out_dtype=torch.float32, tile_m=tmh), 2 * M * H * H),
             ("ffn1     bias+gelu+gelu' -> bf16", lambda: ops.gemm_nt(x_lp, w_1, bias=bf, act=3, want_pre=True, out=o_h, tile_m=tmh), 2 * M * F * H),
             ("ffn2     bias+resid      -> f32 ", lambda: ops.gemm_nt(h_lp, w_2, bias=b1, resid=res, out_dtype=torch.float32, tile_m=tmh), 2 * M * H * F),
             ("d ffn2   * gelu'         -> bf16", lambda: ops.gemm_nt(x_lp, w_2_t, gelu_in=u_lp, act=4, out=o_du, tile_m=tmh), 2 * M * F * H),
             ("(same shape, plain)      -> bf16", lambda: ops.gemm_nt(x_lp, w_2_t, out=o_du, tile_m=tmh), 0),
             ("d ffn1                   -> f32 ", lambda: ops.gemm_nt(h_lp, w_1_t, out_dtype=torch.float32, tile_m=tmh), 2 * M * H * F),
             ("d out-proj               -> bf16", lambda: ops.gemm_nt(x_lp, w_o_t, tile_m=tmh), 2 * M * H * H),
             ("d qkv                    -> f32 ", lambda: ops.gemm_nt(dqkv, w_qkv_t, out_dtype=torch.float32, tile_m=tmh), 2 * M * H * 3 * H)]
    tot_t = tot_f = 0.0
    for name, fn, fl in cases:
        lo, med = timeit(fn, iters=20, reps=5)
        if fl:
            tot_t += lo
            tot_f += fl
        print(f"[{tag}] layer-NT {name}: min {lo:8.1f} us  med {med:8.1f} us  {fl / lo / 1e6:7.1f} TF")
    print(f"[{tag}] layer-NT total (b={B}, tile hint {tmh}): {tot_t:8.1f} us  {tot_f / tot_t / 1e6:7.1f} TF")
if "tn" in what:
    for (name, M, N1, N2) in [("video dWqkv", B * 1464, 2304, 768), ("video dWo", B * 1464, 768, 768), ("video dW1", B * 1464, 3072, 768),
                              ("video dW2", B * 1464, 768, 3072), ("text dW1", B * 128, 3072, 768)]:
        a, b = rnd(M, N1), rnd(M, N2)
        lo, med = timeit(lambda: ops.gemm_tn(a, b))
        print(f"[{tag}] gemm_tn {name:12s} M={M:6d} N1={N1:5d} N2={N2:5d}: min {lo:8.1f} us  med {med:8.1f} us  {2 * M * N1 * N2 / lo / 1e6:7.1f} TF")
if "tng" in what:         # the four weight gradients of one encoder layer in one grouped launch (what the engine runs), per branch
    for (name, S) in [("video", 1464), ("fusion", 481), ("audio", 249), ("text", 128)]:
        M, H, F = B * S, 768, 3072
        dqkv, a_lp, dy1, o_lp, du, c_lp, dy2, h_lp = rnd(M, 3 * H), rnd(M, H), rnd(M, H), rnd(M, H), rnd(M, F), rnd(M, H), rnd(M, H), rnd(M, F)
        pairs = [(dqkv, a_lp), (dy1, o_lp), (du, c_lp), (dy2, h_lp)]
        lo, med = timeit(lambda: ops.gemm_tn_grouped(pairs, want_bias=True), iters=20, reps=5)
        fl = 2 * M * (3 * H * H + H * H + 2 * F * H)
        print(f"[{tag}] gemm_tn_grouped {name:7s} layer M={M:6d}: min {lo:8.1f} us  med {med:8.1f} us  {fl / lo / 1e6:7.1f} TF")
        lo, med = timeit(lambda: ops.gemm_tn_grouped(pairs, want_bias=False), iters=20, reps=5)
        print(f"[{tag}]   (no bias sums)  {name:7s} layer M={M:6d}: min {lo:8.1f} us  med {med:8.1f} us  {fl / lo / 1e6:7.1f} TF")
        for fl_name, flg in (("128-wide tiles", 2), ("256-wide x1", 1 | (1 << 8)), ("256-wide x2", 1 | (2 << 8)), ("256-wide x3", 1 | (3 << 8)), ("256-wide x7", 1 | (7 << 8))):
            if (flg >> 8) > 1 and M // (flg >> 8) < 64:
                continue
            lo, med = timeit(lambda: ops.gemm_tn_grouped(pairs, want_bias=True, flags=flg), iters=20, reps=5)
            print(f"[{tag}]   {fl_name:15s} {name:7s} layer M={M:6d}: min {lo:8.1f} us  med {med:8.1f} us  {fl / lo / 1e6:7.1f} TF")
if "tiles" in what:       # NT GEMM: workgroup tile sweep (hint 2/3/4 = 64/96/128 x 128 with 4 waves, 8 = 256 x 128 and 16 = 256 x 256 with 8 waves)
    for (name, M, N, K) in [("video qkv", B * 1464, 2304, 768), ("video out", B * 1464, 768, 768), ("video ffn1", B * 1464, 3072, 768),
                            ("video ffn2", B * 1464, 768, 3072), ("video dffn1", B * 1464, 768, 3072), ("video dqkv", B * 1464, 768, 2304),
                            ("fusion qkv", B * 481, 2304, 768), ("fusion ffn1", B * 481, 3072, 768), ("audio ffn1", B * 249, 3072, 768),
                            ("conv1", B * 7999, 512, 1536), ("square 4096", 4096, 4096, 4096)]:
        a, b = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        row = []
        for tm in (0, 17, 2, 3, 4, 8, 16):      # 0 = library's plan (may mix tiles), 17 = its best single tile
            lo, _ = timeit(lambda: ops.gemm_nt(a, b, bias=bias, tile_m=tm), iters=30, reps=5)
            row.append(f"tm{tm} {lo:6.1f}")
        print(f"[{tag}] gemm_nt {name:12s} M={M:6d} N={N:5d} K={K:5d}: " + "  ".join(row) + "  us")
if "ring" in what:        # small-grid NT GEMMs (text / audio / fusion shapes): LDS ring depth sweep (hint = tm + 16 * depth)
    for (name, M, N, K) in [("text out", B * 128, 768, 768), ("text ffn2", B * 128, 768, 3072), ("text qkv", B * 128, 2304, 768), ("text ffn1", B * 128, 3072, 768),
                            ("audio out", B * 249, 768, 768), ("audio ffn2", B * 249, 768, 3072), ("audio ffn1", B * 249, 3072, 768),
                            ("fusion out", B * 481, 768, 768), ("fusion ffn2", B * 481, 768, 3072), ("fusion qkv", B * 481, 2304, 768)]:
        a, b = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        row = []
        lo, _ = timeit(lambda: ops.gemm_nt(a, b, bias=bias, out_dtype=torch.float32 if N == 768 else None), iters=30, reps=5)
        row.append(f"auto {lo:5.1f}")
        for tm in (2, 3, 4):
            for nst in (2, 3, 4):
                lo, _ = timeit(lambda: ops.gemm_nt(a, b, bias=bias, tile_m=tm + 32 * nst, out_dtype=torch.float32 if N == 768 else None), iters=30, reps=5)
                row.append(f"tm{tm}s{nst} {lo:5.1f}")
        print(f"[{tag}] gemm_nt {name:11s} M={M:5d} N={N:4d} K={K:4d}: " + " ".join(row))
if "ln" in what:          # LayerNorm backward (with dgamma/dbeta) at the four branch sizes
    for (name, rows) in [("video", B * 1464), ("fusion", B * 481), ("audio", B * 249), ("text", B * 128)]:
        x = torch.randn(rows, 768, device=dev)
        g, bt = torch.ones(768, device=dev), torch.zeros(768, device=dev)
        _, _, mean, rstd = ops.ln_fwd(x, g, bt, 1e-12, want_f32=False, lp_dtype=torch.bfloat16)
        dy = torch.randn(rows, 768, device=dev)
        lo, med = timeit(lambda: ops.ln_bwd(dy, x, g, bt, mean, rstd, want_f32=True, lp_dtype=torch.bfloat16))
        print(f"[{tag}] ln_bwd {name:7s} rows={rows:6d}: min {lo:6.1f} us  med {med:6.1f} us")
        lo, med = timeit(lambda: ops.ln_fwd(x, g, bt, 1e-12, want_f32=False, lp_dtype=torch.bfloat16))
        print(f"[{tag}] ln_fwd {name:7s} rows={rows:6d}: min {lo:6.1f} us  med {med:6.1f} us")
if "gn" in what:          # wav2vec2 front-end GroupNorm + GELU over the conv0 output [B, 15999, 512] (bf16), forward and backward
    x = rnd(B, 15999, 512)
    g, bt = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    y, stats = ops.gn_gelu_fwd(x, g, bt, 1e-5)
    dy = rnd(B, 15999, 512)
    lo, med = timeit(lambda: ops.gn_gelu_fwd(x, g, bt, 1e-5), iters=20, reps=5)
    print(f"[{tag}] gn_gelu fwd (stats + finalize + apply) B={B}: min {lo:6.1f} us  med {med:6.1f} us")
    lo, med = timeit(lambda: ops.gn_gelu_bwd(x, dy, g, bt, stats), iters=20, reps=5)
    print(f"[{tag}] gn_gelu bwd (stats + finalize + param grads + apply) B={B}: min {lo:6.1f} us  med {med:6.1f} us")
if "adamw" in what:       # fused clip + AdamW over a BERT-base-like parameter list (110 M parameters)
    from tav_amd.optim import FusedAdamW
    shapes = [(30522, 768)] + [(768, 768)] * 48 + [(3072, 768)] * 12 + [(768, 3072)] * 12 + [(768,)] * 100 + [(3072,)] * 12
    ps = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.02) for s in shapes]
    for p_ in ps:
        p_.grad = torch.randn_like(p_) * 1e-3
    opt = FusedAdamW(ps, lr=1e-4, weight_decay=1e-2)
    n = sum(p_.numel() for p_ in ps)
    lo, med = timeit(lambda: opt.clip_and_step(1.0), iters=20, reps=7)
    print(f"[{tag}] clip+adamw {n / 1e6:.0f} M params: min {lo:7.1f} us  med {med:7.1f} us  {n * 32 / lo / 1e6:6.2f} TB/s (32 B/param incl. the norm pass)")
