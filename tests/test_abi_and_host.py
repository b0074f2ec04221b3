"""CPU: the C-ABI library loads and exports every symbol include/tavhip.h declares (no compute calls without a GPU),
argument validation returns TAV_ERR_* before anything is launched, the product path refuses to run without the GPU /
library (no CPU fallback), and the host-side logic (masks, samplers, schedules, key remaps, synthetic batches)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest
import torch

import tav_amd  # noqa: F401
from oracle import tav_oracle as O
from tav_amd import _lib, synthetic
from tav_amd import config as cfgmod
from tav_amd.models.tav import PreFormer, TAVForMAE, collate_batch, remap_reference_keys
from tav_amd.train_model.tav_train import CosineWarmRestarts
from tav_amd.utils.global_functions import MySampler, Metrics, arg_parse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tavhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tav_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    h = _lib.lib()
    names = _declared()
    assert len(names) >= 50
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/tavhip.h but not exported by libtavhip.so"
    assert set(names) == set(_lib.declared_symbols()), set(names) ^ set(_lib.declared_symbols())
    assert h.tav_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    h = _lib.lib()
    g = _lib.GemmNTArgs()
    assert h.tav_gemm_nt(C.byref(g), None) == -1                       # TAV_ERR_NULL
    a = _lib.AttnArgs()
    assert h.tav_attn_fwd(C.byref(a), None) == -1
    assert h.tav_ln_fwd(C.byref(_lib.LnArgs()), None) == -1
    assert h.tav_error_string(-2).decode() == "unsupported shape"
    cr, ns = C.c_int32(), C.c_int32()
    assert h.tav_gemm_tn_splits(768, 768, 11712, 1, C.byref(cr), C.byref(ns)) == 0
    assert cr.value % 64 == 0 and ns.value * cr.value >= 11712
    assert h.tav_gemm_tn_splits(512, 1536, 7999, 8, C.byref(cr), C.byref(ns)) == 0 and ns.value % 8 == 0


def test_nt_schedule_mixes_tiles_only_when_the_last_round_is_short():
    """tav_gemm_nt_schedule is host arithmetic: the video stack's N = 768 products at batch 32 (2.14 rounds of 256 x 256 tiles) take
    two whole rounds of big tiles and hand the remaining rows to the 128-wide tile; batched, fp32 and forced-tile calls never split."""
    h = _lib.lib()

    def plan(M, N, K, in_dt=1, out_dt=1, nzb=1, hint=0, resid=False):
        g = _lib.GemmNTArgs()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, K, N
        g.in_dtype, g.out_dtype, g.nzb, g.nzg, g.tile_m_hint = in_dt, out_dt, nzb, 1, hint
        if resid:
            g.resid = 1                                  # only tested for NULL-ness by the planner
        t, r, tr = C.c_int32(), C.c_int32(), C.c_int32()
        assert h.tav_gemm_nt_schedule(C.byref(g), C.byref(t), C.byref(r), C.byref(tr)) == 0
        return t.value, r.value, tr.value

    M = 32 * 1464
    t, r, tr = plan(M, 768, 768)
    assert t == 16 and r == 170 * 256 and tr in (2, 3, 4)          # 510 big tiles = two rounds, 3328 rows left
    assert plan(M, 768, 768, hint=17)[1:] == (M, 0)                # one tile for all rows on request
    assert plan(M, 768, 768, hint=4) == (4, M, 0)
    assert plan(M, 768, 768, in_dt=0, out_dt=0)[1:] == (M, 0)      # fp32 operands: 128-wide tiles only
    assert plan(M, 768, 768, nzb=2)[1:] == (M, 0)                  # batched calls are never split
    assert plan(4 * 128, 768, 768)[1:] == (4 * 128, 0)             # the text branch: under one round, nothing to split
    t, r, tr = plan(256 * 256, 256, 768)                           # exactly one full round of big tiles: nothing left over
    assert r == 256 * 256 and tr == 0
    g = _lib.GemmNTArgs()
    t_, r_, tr_ = C.c_int32(), C.c_int32(), C.c_int32()
    assert h.tav_gemm_nt_schedule(C.byref(g), C.byref(t_), C.byref(r_), C.byref(tr_)) == -2     # M = 0: unsupported shape


def test_nt_schedule_invariants_over_random_shapes():
    """Whatever the shape, the plan is well formed: the first launch covers a positive number of rows <= M; a second launch only follows whole
    256-row tiles of the 256 x 256 tile, takes a 128-wide tile, and never appears for f32 operands, batched calls or forced tiles."""
    import random
    h = _lib.lib()
    rnd = random.Random(1234)
    for _ in range(400):
        M = rnd.choice([1, 7, 128, 255, 256, 257, 1000, 4096, 11712, 23424, 46848, 46885, 255968]) + rnd.randrange(0, 3)
        N = 4 * rnd.randrange(1, 1025)
        K = 64 * rnd.randrange(1, 65)
        in_dt = rnd.choice([0, 1])
        out_dt = 0 if in_dt == 0 else rnd.choice([0, 1])
        g = _lib.GemmNTArgs()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, K, N
        g.in_dtype, g.out_dtype, g.nzb, g.nzg = in_dt, out_dt, rnd.choice([1, 1, 1, 3]), 1
        g.tile_m_hint = rnd.choice([0, 0, 0, 2, 3, 4, 16, 17])
        g.act = rnd.choice([0, 0, 1, 3])
        if rnd.random() < 0.3:
            g.resid = 1
        t, r, tr = C.c_int32(), C.c_int32(), C.c_int32()
        rc = h.tav_gemm_nt_schedule(C.byref(g), C.byref(t), C.byref(r), C.byref(tr))
        assert rc == 0, (rc, M, N, K)
        assert t.value in (2, 3, 4, 8, 16) and 0 < r.value <= M
        if tr.value:
            assert t.value == 16 and r.value % 256 == 0 and r.value < M and tr.value in (2, 3, 4)
            assert in_dt == 1 and g.nzb == 1 and g.tile_m_hint == 0
        else:
            assert r.value == M
        if in_dt == 0:
            assert t.value in (2, 3, 4)


def test_grouped_wgrad_workspace_plan_is_host_arithmetic():
    """tav_gemm_tn_grouped_ws_bytes: a 768-wide layer's four gradients at batch 32 (46848 rows, bf16) take the 256-wide tile with two token
    splits -- two f32 slabs of every gradient plus the bias partials [2 splits][N2 / 256 tiles][N1]; small or f32 problems only need the
    bias partials of the 128-wide tile rows; forcing flags change the plan."""
    h = _lib.lib()
    H, F = 768, 3072
    shapes = [(3 * H, H), (H, H), (F, H), (H, F)]

    def nbytes(rows, dtype=1, flags=0, bias=True):
        probs = (_lib.GemmTNProblem * 4)()
        for k, (n1, n2) in enumerate(shapes):
            probs[k].A, probs[k].B, probs[k].out, probs[k].dbias = 1, 1, 1, (1 if bias else None)
            probs[k].N1, probs[k].N2, probs[k].lda, probs[k].ldb = n1, n2, n1, n2
        return h.tav_gemm_tn_grouped_ws_bytes(probs, 4, rows, dtype, flags)

    elems = sum(a * b for a, b in shapes)
    big_bias = sum(2 * ((n2 + 255) // 256) * n1 for n1, n2 in shapes)
    small_bias = sum(((n2 + 127) // 128) * n1 for n1, n2 in shapes)
    assert nbytes(32 * 1464) == 4 * (2 * elems + big_bias)
    assert nbytes(32 * 1464, bias=False) == 4 * 2 * elems
    assert nbytes(4 * 1464) == 4 * small_bias                     # too few rows for slabs to pay: 128-wide tiles, bias partials only
    assert nbytes(32 * 1464, dtype=0) == 4 * small_bias           # f32 operands: 128-wide only
    assert nbytes(32 * 1464, flags=2) == 4 * small_bias           # forced 128-wide
    assert nbytes(4 * 1464, flags=1 | (3 << 8)) == 4 * (3 * elems + sum(3 * ((n2 + 255) // 256) * n1 for n1, n2 in shapes))
    assert h.tav_gemm_tn_grouped_ws(None, 4, 100, 1, None, 0, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtavhip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    cfg = cfgmod.preset("B-tiny")
    model = TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg)
    (tx, au, vi), _ = synthetic.make_batch(cfg, 2, s_text=8, t_audio=3200, n_visual_true=4)
    with pytest.raises(RuntimeError, match="no CPU fallback|GPU"):
        model(tx["input_ids"], tx["attention_mask"], au["audio_features"], vi["visual_embeds"], vi["attention_mask"], torch.zeros(2, 21, 768),
              torch.zeros(2, 21, dtype=torch.long), torch.zeros(2, 1, 1, 21))


def test_state_dict_keys_follow_reference_names():
    cfg = cfgmod.preset("A-tiny")
    model = TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=False, num_layers=12), cfg)
    keys = set(model.state_dict())
    for k in ["embedding.weight", "bert_norm.weight", "rand_norm.bias", "vid_norm.weight", "aud_norm.weight", "linear1.weight", "wav_2_768_2.bias",
              "random_mae_encoder.layer.0.layernorm_before.weight", "random_mae_encoder.layer.1.attention.attention.q_bias",
              "random_mae_encoder.layer.1.attention.attention.key.weight", "random_mae_encoder.layer.0.attention.output.dense.bias",
              "random_mae_encoder.layer.0.intermediate.dense.weight", "random_mae_encoder.layer.0.output.dense.weight",
              "bert.embeddings.word_embeddings.weight", "bert.encoder.layer.0.attention.self.query.weight", "bert.pooler.dense.bias",
              "wav2vec2.feature_extractor.conv_layers.0.conv.weight", "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
              "wav2vec2.encoder.layers.0.attention.k_proj.bias", "videomae.embeddings.patch_embeddings.projection.weight",
              "videomae.encoder.layer.0.attention.attention.query.bias"]:
        assert k in keys, k
    assert "random_mae_encoder.layer.0.attention.attention.key.bias" not in keys      # reference: k has no bias (utils/TAVFormer.py:347)
    assert not model.embedding.weight.requires_grad                                      # learn_PosEmbeddings=False (models/tav.py:436)
    pre = PreFormer(cfg)
    assert {"masked_spec_embed", "wav_2_768.weight", "bert.embeddings.LayerNorm.weight"} <= set(pre.state_dict())
    # reference randomize_model (models/tav.py:461-471): fusion LayerNorm = (1, 0), biases 0
    L = model.random_mae_encoder.layer[0]
    assert torch.all(L.layernorm_before.weight == 1) and torch.all(L.layernorm_before.bias == 0) and torch.all(L.output.dense.bias == 0)


def test_randomize_model_matches_oracle_init_rule():
    """M1 (reference models/tav.py:442, :461-471): from one RNG state the product's `.apply(randomize_model)` leaves the fusion stack with
    exactly the weights the oracle's restatement of the rule gives (which oracle/validate_vs_reference.py pins, bit for bit, against the
    reference's own function body): same traversal, so the same number of xavier draws per Linear, the last one staying."""
    from tav_amd.utils.TAVFormer import VideoMAEEncoder
    cfg = cfgmod.preset("B-tiny")
    model = TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg)
    fcfg, nl = dict(cfg["fusion"]), cfg["fusion"]["layers"]
    torch.manual_seed(5)
    a = VideoMAEEncoder(fcfg, nl)
    st = torch.get_rng_state()
    a.apply(model.randomize_model)
    torch.manual_seed(5)
    b = VideoMAEEncoder(fcfg, nl)
    torch.set_rng_state(st)
    O.randomize_model_(b)
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    w = sa["layer.0.intermediate.dense.weight"]
    bound = math.sqrt(6.0 / (w.shape[0] + w.shape[1]))                       # xavier_uniform
    assert w.abs().max() <= bound and w.abs().max() > 0.9 * bound


def test_capture_guard_refuses_edges_that_crash_end_capture():
    """runtime.check_edge / stream_wait: while a capture is active, a dependency whose two ends are both branch streams, or that touches a
    stream the capture never registered, raises (round 3: such a graph dumped core in hipStreamEndCapture); fork / join through the
    origin stays legal.  Host logic only -- the streams here are stand-ins with identity."""
    from tav_amd import runtime

    class FakeStream:
        def __init__(self, name):
            self.name, self.waits = name, []

        def __repr__(self):
            return f"<stream {self.name}>"

        def wait_stream(self, other):
            self.waits.append(("stream", other))

        def wait_event(self, ev):
            self.waits.append(("event", ev))

    origin, a, b, stray = FakeStream("origin"), FakeStream("audio"), FakeStream("video"), FakeStream("stray")
    runtime.stream_wait(a, b)                                   # no capture: anything goes
    assert not runtime.capture_active()
    with runtime.guard_only(origin, branches=[a, b]):
        assert runtime.capture_active()
        runtime.stream_wait(a, origin, event="ev")              # fork: branch waits for an event recorded on the origin
        runtime.stream_wait(origin, a)                          # join: origin waits for the branch
        runtime.stream_wait(origin, origin)
        with pytest.raises(RuntimeError, match="two branch streams.*audio.*video"):
            runtime.stream_wait(a, b)
        with pytest.raises(RuntimeError, match="stray.*neither the capture's origin nor one of its registered branches"):
            runtime.stream_wait(stray, origin)
        with pytest.raises(RuntimeError, match="stray"):
            runtime.stream_wait(origin, stray, event="ev")
        n_before = len(a.waits)
        with pytest.raises(RuntimeError):
            runtime.stream_wait(a, stray)
        assert len(a.waits) == n_before                        # refused BEFORE the dependency is made
    assert not runtime.capture_active()
    assert a.waits[0] == ("stream", b) and ("event", "ev") in a.waits and origin.waits[0] == ("stream", a)


def test_gradient_arena_slots_are_single_use():
    """engine._arena_into: a layer differentiated twice within one data-parallel segment must not write its second weight gradient into the
    arena slot the first one already owns (AccumulateGrad would then add the slot to itself); the second request gets no destinations."""
    from tav_amd import engine, runtime
    H, F = 4, 8
    mk = lambda *s: torch.nn.Parameter(torch.zeros(*s))           # noqa: E731
    wq, wk, wv, wo, w1, w2 = mk(H, H), mk(H, H), mk(H, H), mk(H, H), mk(F, H), mk(H, F)
    bq, bk, bv, bo, b1, b2 = mk(H), mk(H), mk(H), mk(H), mk(F), mk(H)
    order = [wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2]
    flat = torch.zeros(sum(p.numel() for p in order))
    off, views = 0, {}
    for p in order:
        views[p.data_ptr()] = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    runtime.grad_slots.clear()
    runtime.grad_slots.update(views)
    try:
        first = engine._arena_into(wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2)
        assert all(d is not None for pair in first for d in pair)
        assert first[3][0].shape == (3 * H, H) and first[3][0].data_ptr() == views[wq.data_ptr()].data_ptr()     # Wq | Wk | Wv as one fused destination
        second = engine._arena_into(wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2)
        assert second is None or all(d is None for pair in second for d in pair)
    finally:
        runtime.grad_slots.clear()


def test_remap_reference_keys():
    sd = {"videomae.encoder.layer.0.attention.attention.q_bias": torch.ones(4), "videomae.encoder.layer.0.attention.attention.v_bias": torch.ones(4) * 2,
          "wav2vec2.encoder.pos_conv_embed.conv.weight_g": torch.ones(1, 1, 3), "wav2vec2.encoder.pos_conv_embed.conv.weight_v": torch.ones(2, 2, 3),
          "bert.embeddings.position_ids": torch.arange(4), "random_mae_encoder.layer.0.attention.attention.q_bias": torch.zeros(4), "linear1.weight": torch.zeros(1)}
    out = remap_reference_keys(sd)
    assert torch.all(out["videomae.encoder.layer.0.attention.attention.query.bias"] == 1)
    assert torch.all(out["videomae.encoder.layer.0.attention.attention.key.bias"] == 0)
    assert torch.all(out["videomae.encoder.layer.0.attention.attention.value.bias"] == 2)
    assert "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0" in out
    assert "bert.embeddings.position_ids" not in out and "random_mae_encoder.layer.0.attention.attention.q_bias" in out


def test_preformer_mask_helpers_match_oracle():
    cfg = cfgmod.preset("B-tiny")
    pre = PreFormer(cfg)
    am = torch.ones(3, 16000)
    am[0, 12000:] = 0
    am[2, 401:] = 0
    T = pre.wav2vec2.conv_out_len(16000)
    assert T == 49
    got = pre._get_feature_vector_attention_mask(T, am)
    ref = O.feature_vector_attention_mask(T, am, cfg["audio"])
    assert torch.equal(got, ref) and got.dtype == torch.bool
    assert got.sum(1).tolist() == [int(O.w2v2_conv_out_lengths(torch.tensor(n), cfg["audio"])) for n in (12000, 16000, 401)]
    assert pre.wav2vec2.conv_out_len(80000) == 249 and pre.wav2vec2.conv_out_len(160000) == 499      # SURVEY.md §8 derived S


def test_synthetic_batch_contract():
    cfg = cfgmod.preset("B")
    (tx, au, vi), labels = synthetic.make_batch(cfg, 3, seed=1234)
    assert tx["input_ids"].shape == (3, 128) and tx["input_ids"].dtype == torch.int64 and tx["attention_mask"].dtype == torch.float32
    assert (tx["input_ids"][:, 96:] == cfg["text"]["pad_id"]).all() and (tx["attention_mask"][:, 96:] == 0).all() and (tx["attention_mask"][:, :96] == 1).all()
    assert au["audio_features"].shape == (3, 80000) and (au["audio_features"][0, 64000:] == 0).all() and au["attention_mask"][0].sum() == 64000
    assert vi["visual_embeds"].shape == (3, 16, 3, 224, 224) and vi["attention_mask"].dtype == torch.bool
    assert vi["attention_mask"].sum(1).tolist() == [104, 104, 104]                                   # equal per-row counts (SURVEY hard parts)
    assert labels.dtype == torch.float32 and labels.min() >= 0 and labels.max() <= 6
    (tx2, _, _), _ = synthetic.make_batch(cfg, 3, seed=1234)
    assert torch.equal(tx["input_ids"], tx2["input_ids"])


def test_collate_batch_contract():
    torch.manual_seed(0)
    items = []
    for n in (3000, 2000):
        text = {"input_ids": torch.randint(3, 100, (1, 70)), "attention_mask": torch.ones(1, 70)}
        items.append(([text, torch.randn(n), torch.randn(16, 3, 32, 32)], 3))
    (tx, au, vi), labels = collate_batch(items, "train")
    assert tx["input_ids"].shape == (2, 70) and au["audio_features"].shape == (2, 3000) and au["attention_mask"][1].sum() == 2000
    assert (au["audio_features"][1, 2000:] == 0).all()
    assert vi["visual_embeds"].shape == (2, 16, 3, 32, 32) and vi["attention_mask"].shape == (2, 32)
    assert vi["attention_mask"][0].sum() == vi["attention_mask"][1].sum()
    assert labels.tolist() == [3.0, 3.0]


def test_sampler_schedule_metrics_flags():
    s = MySampler([0.5, 0.25, 0.25, 1.0], 4, replacement=True, epoch=0, epoch_switch=2)
    a = list(iter(s))
    b = list(iter(s))
    assert len(a) == 4 and b == [0, 1, 2, 3] and s.epoch == 2                       # multinomial epoch, then sequential epoch

    class _Opt:
        lr = 1e-3
    o = _Opt()
    sch = CosineWarmRestarts(o, T_0=2)
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    ref = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ref_opt, T_0=2)
    for e in (0.0, 0.25, 1.0, 1.75, 2.0, 3.5):
        sch.step(e)
        ref.step(e)
        assert math.isclose(o.lr, ref.get_last_lr()[0], rel_tol=1e-9, abs_tol=1e-12)

    m = Metrics(3)
    m.update_metrics(torch.tensor([0, 1, 1, 2]), torch.tensor([0, 1, 2, 2]))
    *_, acc, f1m, f1w, rec, prec, cm = m.compute_scores("val")
    assert cm.tolist() == [[1, 0, 0], [0, 1, 0], [0, 1, 1]] and abs(acc - (1 + 1 + 0.5) / 3) < 1e-9
    args = arg_parse("x", ["-l", "0.001", "--batch_size", "4", "--clip", "2.0", "--preset", "B"])
    assert args.learning_rate == 0.001 and args.batch_size == 4 and args.clip == 2.0 and args.loss == "NewCrossEntropy" and args.T_max == 2


# ---- checkpoint format and train-loop parity (host logic, CPU) -----------------------------------------
def test_cosine_warm_restarts_matches_torch():
    from tav_amd.train_model.tav_train import CosineWarmRestarts

    class _Opt:
        lr = 3e-4
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.AdamW([p], lr=3e-4)
    ref = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ref_opt, T_0=2)
    mine = CosineWarmRestarts(_Opt(), T_0=2)
    for e in (0.0, 0.25, 1.0, 1.9, 2.0, 3.5, 4.0):
        ref.step(e)
        mine.step(e)
        assert abs(mine.get_last_lr()[0] - ref.get_last_lr()[0]) < 1e-12
    sd = mine.state_dict()
    assert {"T_0", "T_i", "T_mult", "eta_min", "T_cur", "base_lrs", "last_epoch", "_last_lr"} <= set(sd) and set(sd) <= set(ref.state_dict())
    other = CosineWarmRestarts(_Opt(), T_0=7)
    other.load_state_dict(ref.state_dict())                       # a scheduler state saved by the reference resumes here
    assert other.T_0 == 2 and abs(other.get_last_lr()[0] - ref.get_last_lr()[0]) < 1e-12


def test_fused_adamw_state_dict_is_torch_format():
    from tav_amd.optim import FusedAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.AdamW(ps, lr=1e-3, weight_decay=1e-2)
    for p in ps:
        p.grad = torch.randn_like(p)
    ref.step()
    ref.step()
    sd = ref.state_dict()
    opt = FusedAdamW(ps, lr=5.0, weight_decay=0.5)
    opt.load_state_dict(sd)                                       # torch checkpoint -> fused optimizer
    assert opt.lr == 1e-3 and opt.weight_decay == 1e-2 and opt.step_count == 2
    out = opt.state_dict()
    assert out["param_groups"][0]["params"] == [0, 1] and set(out["state"]) == {0, 1}
    for i in (0, 1):
        assert torch.equal(out["state"][i]["exp_avg"], sd["state"][i]["exp_avg"]) and torch.equal(out["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
        assert float(out["state"][i]["step"]) == 2.0
    fresh = torch.optim.AdamW(ps, lr=1.0)
    fresh.load_state_dict(out)                                    # and back: fused state -> torch.optim.AdamW
    assert fresh.param_groups[0]["lr"] == 1e-3 and float(fresh.state[ps[0]]["step"]) == 2.0


def test_best_pt_round_trip(tmp_path):
    """save_model / load_model write and read the reference's best.pt layout (utils/global_functions.py:199-258), including a checkpoint
    that carries transformers-4.2x VideoMAE key names."""
    from tav_amd.models.tav import PreFormer, TAVForMAE
    from tav_amd.optim import FusedAdamW
    from tav_amd.train_model.tav_train import CosineWarmRestarts
    from tav_amd.utils import global_functions as G
    cfg = cfgmod.preset("B-tiny")
    args = dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12)
    torch.manual_seed(1)
    model, pre = TAVForMAE(args, cfg), PreFormer(cfg)
    crit = G.NewCrossEntropyLoss(torch.ones(7))
    opt = FusedAdamW(list(model.parameters()) + list(pre.parameters()), lr=2e-5, weight_decay=1e-4)
    sched = CosineWarmRestarts(opt, T_0=2)
    sched.step(0.5)
    f = G.save_model(model, pre, opt, crit, sched, 3, 17, str(tmp_path), 2400)
    ck = torch.load(f, weights_only=False)
    assert {"epoch", "step", "model_state_dict", "optimizer_state_dict", "loss", "scheduler", "PREFormer"} == set(ck) and ck["epoch"] == 3 and ck["step"] == 17
    want = {k: v.clone() for k, v in model.state_dict().items()}
    # rewrite the VideoMAE attention biases the way a 4.2x-era reference checkpoint names them, then reload through the remap
    sd = ck["model_state_dict"]
    for k in [k for k in sd if k.startswith("videomae.") and k.endswith(".attention.attention.query.bias")]:
        base = k[: -len("query.bias")]
        sd[base + "q_bias"] = sd.pop(k)
        sd[base + "v_bias"] = sd.pop(base + "value.bias")
        sd.pop(base + "key.bias")
    torch.save(ck, f)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    opt.lr = 1.0
    G.load_model(model, pre, opt, crit, str(tmp_path))
    for k, v in model.state_dict().items():
        if not k.endswith(".attention.attention.key.bias"):
            assert torch.equal(v, want[k]), k
    import math
    assert abs(opt.lr - 2e-5 * (1 + math.cos(math.pi * 0.25)) / 2) < 1e-12      # the learning rate the scheduler had set when the checkpoint was written


def test_grad_accum_loop_steps_like_the_reference(monkeypatch):
    """train_model/tav_train.py:87-119: loss / dialogue length, an optimizer step after EVERY batch and a second (gradient-free) one at each
    dialogue end; validation + best-checkpoint bookkeeping on the last batch."""
    from tav_amd.train_model import tav_train as T

    class Loss:
        def __init__(self, v):
            self.v = v

        def __truediv__(self, d):
            return Loss(self.v / d)

        def item(self):
            return self.v

        def backward(self):
            calls.append(("backward", self.v))

    class Stepper:
        reducer = None
        opt = None

        def update(self, clip=True):
            calls.append(("update",) if clip else ("update-unclipped",))

    class Sched:
        def step(self, e):
            calls.append(("sched", round(e, 4)))

    class DS:
        grad, grad_sum, ctr = [2, 3], [2, 5], 0              # two dialogues: 2 and 3 utterances (utils/data_loaders.py:47-57)

        def retGradAccum(self, i):
            r, s = self.grad[self.ctr], self.grad_sum[self.ctr]
            if i + 1 == self.grad_sum[self.ctr]:
                self.ctr += 1
            if self.ctr == len(self.grad):
                self.ctr = 0
            return r, s

    class DL(list):
        dataset = DS()
    calls = []
    dl = DL([(None, None)] * 5)
    monkeypatch.setattr(T, "get_statistics", lambda *a, **k: Loss(6.0))
    monkeypatch.setattr(T, "validate", lambda *a, **k: 0.5)
    monkeypatch.setattr(T, "log", lambda *a, **k: None)
    T.PATIENCE_ITER = 0
    best = T.grad_accum(1, dl, None, None, None, None, Stepper(), Sched(), 10, None, 100, 2400, None)
    assert best == 0.5
    assert [c[1] for c in calls if c[0] == "backward"] == [3.0, 3.0, 2.0, 2.0, 2.0]
    seq = [c[0] for c in calls]
    # batch index 1 ends dialogue 1 (accum_sum = 2: (1+1) % 2 == 0), batch index 4 ends dialogue 2 ((4+1) % 5 == 0) and is the last one
    assert seq == ["backward", "update", "sched",
                   "backward", "update", "sched", "update-unclipped", "sched",       # reference :102-106: optimizer.step() without clip_grad_norm_
                   "backward", "update", "sched",
                   "backward", "update", "sched",
                   "backward", "update", "sched", "update-unclipped", "sched"]


def test_zero_grad_readings_torch_1_10_and_torch_2():
    """FusedAdamW.zero_grad / TrainStep.zero_to_none: the reference's `model.zero_grad()` zero-FILLS under its pinned torch 1.10
    (requirements.txt:100) and sets None under torch >= 2; both are offered, the default is the torch-2 reading."""
    from tav_amd.optim import FusedAdamW
    ps = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2))]
    opt = FusedAdamW(ps)
    ps[0].grad = torch.full((3,), 2.0)
    opt.zero_grad(set_to_none=False)
    assert torch.equal(ps[0].grad, torch.zeros(3)) and ps[1].grad is None            # (a gradient that never existed stays None, as in torch)
    assert [p for p in opt._active()] == [ps[0]]                                      # the zero tensor still takes part in the next step
    opt.zero_grad()
    assert ps[0].grad is None and opt._active() == []


def test_collate_batch_device_contract():
    """the device-side collate yields the same contract as collate_batch (shapes, dtypes, zero padding, 0/1 audio mask
    matching the lengths) with exactly n_visual_true True tokens per row (pure torch: checked on the CPU device here)."""
    from tav_amd.models.tav import collate_batch, collate_batch_device, sample_video_mask
    torch.manual_seed(0)
    lens = [700, 1000, 400]
    batch = [([{"input_ids": torch.randint(3, 100, (12,)), "attention_mask": torch.ones(12)}, torch.randn(n), torch.randn(16, 3, 32, 32)], i % 7)
             for i, n in enumerate(lens)]
    (t, a, v), lab = collate_batch_device(batch, "train", device="cpu", n_visual_true=4)
    (t0, a0, v0), lab0 = collate_batch(batch, "train")
    assert t["input_ids"].dtype == torch.int64 and torch.equal(t["input_ids"], t0["input_ids"]) and torch.equal(t["attention_mask"], t0["attention_mask"])
    assert torch.equal(a["audio_features"], a0["audio_features"]) and torch.equal(a["attention_mask"], a0["attention_mask"])
    assert a["attention_mask"].sum(1).tolist() == [float(n) for n in lens] and float(a["audio_features"][2, 400:].abs().max()) == 0.0
    assert torch.equal(v["visual_embeds"], v0["visual_embeds"]) and v["attention_mask"].dtype == torch.bool and v["attention_mask"].shape == v0["attention_mask"].shape
    assert v["attention_mask"].sum(1).tolist() == [4, 4, 4]
    assert torch.equal(lab, lab0) and lab.dtype == torch.float32
    m = sample_video_mask(64, 1568)
    assert m.sum(1).unique().tolist() == [105] and 0.2 < m[:, :784].float().mean() * 15 < 2.0       # round(1568/15) per row, spread over the row


def test_specaugment_distribution_matches_hf_compute_mask_indices():
    """PreFormer._mask_hidden_states (train=True; reference models/tav.py:269-306 -> HF _compute_mask_indices, wav2vec2:101) draws its spans
    with torch ops on the device; the HF function is the reference's sampler.  Statistical comparison on the CPU: no masked frame
    ever lies in a row's padding, the masked fraction per row matches HF's within sampling error, and one epsilon is drawn per call (the span
    counts of equally long rows agree within a call)."""
    from transformers.models.wav2vec2.modeling_wav2vec2 import _compute_mask_indices
    from tav_amd.models.tav import PreFormer
    pre = PreFormer.__new__(PreFormer)                       # only the sampler is exercised: no encoder weights needed
    torch.nn.Module.__init__(pre)
    Ha, B, T = 16, 6, 249
    pre.masked_spec_embed = torch.nn.Parameter(torch.full((Ha,), 7.0))
    lens = torch.tensor([249, 249, 200, 120, 60, 249])
    amask = (torch.arange(T)[None, :] < lens[:, None])
    torch.manual_seed(0)
    np.random.seed(0)
    trials = 300
    ours, hf = torch.zeros(B), torch.zeros(B)
    for _ in range(trials):
        hidden = torch.zeros(B * T, Ha)
        out = pre._mask_hidden_states(hidden, B, T, amask, training=True).view(B, T, Ha)
        sel = out[:, :, 0] == 7.0
        assert not (sel & ~amask).any()                       # never inside the padding
        # equal-length rows 0, 1, 5 take the same NUMBER of spans within one call (one epsilon per call): their masked counts differ only by overlaps
        ours += sel.float().sum(1)
        ref = torch.from_numpy(_compute_mask_indices((B, T), mask_prob=0.05, mask_length=10, attention_mask=amask.long(), min_masks=2))
        assert not (ref & ~amask).any()
        hf += ref.float().sum(1)
    ours, hf = ours / trials, hf / trials
    assert torch.allclose(ours, hf, rtol=0.08), (ours, hf)
    assert (pre._mask_hidden_states(torch.zeros(B * T, Ha), B, T, amask, training=False) == 0).all()      # train=False: untouched
    # feature axis (reference :292-304; off under the default mask_feature_prob = 0): spans of channels zeroed for EVERY frame of a row
    Hf = 64
    pre.masked_spec_embed = torch.nn.Parameter(torch.full((Hf,), 7.0))
    pre.cfg = {"audio": {"mask_time_prob": 0.0, "mask_feature_prob": 0.2, "mask_feature_length": 4, "mask_feature_min_masks": 1}}
    ours_f, hf_f = 0.0, 0.0
    for _ in range(200):
        out = pre._mask_hidden_states(torch.ones(B * T, Hf), B, T, amask, training=True).view(B, T, Hf)
        z = out == 0
        assert torch.equal(z, z[:, :1, :].expand_as(z))           # the same channels on every frame of a row
        ours_f += z[:, 0, :].float().sum(1).mean().item()
        hf_f += float(_compute_mask_indices((B, Hf), mask_prob=0.2, mask_length=4, min_masks=1).sum(1).mean())
    assert abs(ours_f - hf_f) / hf_f < 0.08, (ours_f, hf_f)


def test_emitted_isa_discipline():
    """tools/check_isa.py on the built library: every M0 write belongs to one of common.h's LDS-DMA sequences and nothing
    else touches M0 (the GEMM loops do not save it); every v_readfirstlane_b32 keeps its wait states (to_sgpr's hand-placed s_nops).  Plus
    the checker itself on hand-written disassembly with each violation."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    ci = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ci)
    if not os.path.exists(ci.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    if not os.path.exists(os.path.join(ROOT, "multi-modal-emotion_amd", "libtavhip.so")):
        pytest.skip("libtavhip.so has not been built (python -c 'import __graft_entry__ as g; g.build()')")
    problems, stats = ci.check(os.path.join(ROOT, "multi-modal-emotion_amd", "libtavhip.so"))
    assert stats["lds_dma"] > 1000 and stats["readfirstlane"] > 100 and stats["m0_writes"] >= stats["lds_dma"]
    assert not problems, problems[:5]
    good = """
0000000000001000 <k>:
\ts_mov_b32 m0, s5
\ts_nop 0
\tglobal_load_lds_dwordx4 v1, s[2:3]
\tv_mov_b32_e32 v7, v3
\ts_nop 1
\tv_readfirstlane_b32 s9, v7
\ts_nop 4
\tglobal_load_dword v2, v1, s[8:9]
"""
    assert ci.check_text([("good", good)])[0] == []
    bad_m0 = good.replace("\ts_nop 0\n\tglobal_load_lds", "\tglobal_load_lds")                       # no wait state between the M0 write and the DMA
    bad_user = good + "\ts_movrels_b32 s1, s2\n"                                                     # a compiler-made M0 user
    bad_idx = good + "\ts_set_gpr_idx_on s4, 0x1\n"                                                  # gfx9 VGPR indexing: uses M0 without naming it
    bad_lane = good.replace("\ts_nop 1\n\tv_readfirstlane", "\tv_readfirstlane")                     # VALU write -> lane read, no wait state
    bad_vmem = good.replace("\ts_nop 4\n", "\ts_nop 1\n")                                            # VMEM reads the SGPR 2 wait states later
    for txt in (bad_m0, bad_user, bad_idx, bad_lane, bad_vmem):
        assert ci.check_text([("bad", txt)])[0], txt


def test_head_mask_layouts_and_oracle_head_mask():
    """VideoMAEEncoder's head_mask entry (reference utils/TAVFormer.py:190, :368-370): the layouts HF's get_head_mask produces become per-head
    factors, anything finer is refused; the oracle's restatement multiplies the probabilities before the post-softmax mask is added."""
    from tav_amd.utils.TAVFormer import VideoMAEEncoder
    enc = VideoMAEEncoder(dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12), 1)
    h = torch.arange(12, dtype=torch.float32) / 11.0
    assert torch.equal(enc._head_scale(h, 3), h)
    assert torch.equal(enc._head_scale(h.view(1, 12, 1, 1), 3), h)
    hb = torch.rand(3, 12, 1, 1)
    assert torch.equal(enc._head_scale(hb, 3), hb.view(3, 12))
    assert torch.equal(enc._head_scale(torch.tensor(0.5), 3), torch.full((12,), 0.5))
    with pytest.raises(NotImplementedError):
        enc._head_scale(torch.ones(1, 12, 5, 5), 3)
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 4, 768), None, head_mask=[h], output_attentions=True)      # no GPU: the module refuses, there is no CPU path
    # oracle: probs * head_mask, THEN + mask; a zeroed head returns the mask alone
    sd = {"e." + k: v for k, v in enc.state_dict().items()}
    x = torch.randn(2, 5, 768)
    m = torch.zeros(2, 1, 1, 5)
    m[..., :2] = -0.25
    hm = torch.ones(1, 1, 12, 1, 1)
    hm[0, 0, 3] = 0.0
    probs = []
    y = O.fusion_encoder(sd, "e", x, m, dict(layers=1, heads=12, eps=1e-12), head_mask=hm, probs_out=probs)
    assert y.shape == x.shape and probs[0].shape == (2, 12, 5, 5)
    assert torch.allclose(probs[0][:, 3], m.expand(2, 1, 5, 5)[:, 0])
    assert torch.allclose(probs[0][:, 0].sum(-1), torch.full((2, 5), 1.0 - 0.5), atol=1e-5)


def test_bucket_norm_is_used_only_when_the_gradients_live_in_the_buckets():
    """FusedAdamW.norm_buffers (set by the data-parallel graph chain) must not be trusted by a later step whose gradients are elsewhere."""
    import torch
    from tav_amd.optim import FusedAdamW
    flat = torch.zeros(10 + 6 + 3)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 6)]
    opt = FusedAdamW(ps)
    opt.norm_buffers = [flat[:16]]
    ps[0].grad, ps[1].grad = flat[:10], flat[10:16]
    assert opt._grads_in_norm_buffers(ps)                       # both gradients are views of the bucket, which holds nothing else
    ps[1].grad = torch.zeros(6)                                 # a plain backward re-allocated one of them
    assert not opt._grads_in_norm_buffers(ps)
    ps[1].grad = flat[10:16]
    opt.norm_buffers = [flat[:19]]                              # the bucket holds 3 elements that are nobody's gradient
    opt._norm_ok_key = None
    assert not opt._grads_in_norm_buffers(ps)
