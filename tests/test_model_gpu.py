"""-m gpu: the product path (libtavhip through the reference's nn.Module interface) against the CPU oracle and the
committed golden vectors, plus size-independent properties at the benchmark's full input sizes.
Tolerances (BASELINE.json north_star): 1e-3 relative for the fp32 policy, 1e-2 for bf16 -- logits, loss, global grad-norm."""
import os

import numpy as np
import pytest
import torch

import closed_form as cf
import tav_amd  # noqa: F401
from oracle import tav_oracle as O
from tav_amd import config as C
from tav_amd import engine as E
from tav_amd import runtime, synthetic
from tav_amd.models.tav import PreFormer, TAVForMAE
from tav_amd.optim import FusedAdamW, grad_norm
from tav_amd.utils.TAVFormer import TransformerEncoder, VideoMAEEncoder

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "tav_golden.npz"))
ARGS = dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12)


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _run_product(pre, model, batch, labels, check="val"):
    tav, emb, amask = pre(input_ids=batch["input_ids"], audio_features=batch["audio_features"], video_embeds=batch["video_embeds"],
                          text_mask=batch["text_mask"], audio_mask=batch["audio_mask"], visual_mask=batch["visual_mask"], device="cuda", train=False)
    logits = model(batch["input_ids"], batch["text_mask"], batch["audio_features"], batch["video_embeds"], batch["visual_mask"], tav, emb, amask,
                   batch_size=len(labels), check=check)
    loss = E.CrossEntropyFn.apply(logits, labels.long().cuda(), None)
    return tav, emb, amask, logits, loss


def _as_batch(tx, au, vi):
    return dict(input_ids=tx["input_ids"], text_mask=tx["attention_mask"], audio_features=au["audio_features"], audio_mask=au["attention_mask"],
                video_embeds=vi["visual_embeds"], visual_mask=vi["attention_mask"])


@pytest.mark.parametrize("preset", ["A", "B"])
def test_golden_closed_form_fp32(gpu, preset):
    """Closed-form weights/inputs: product (fp32 policy) reproduces what the reference-side modules produced."""
    cfg = C.preset(preset + "-tiny")
    runtime.set_precision("fp32")
    pre = cf.fill_module_(PreFormer(cfg)).cuda()
    model = cf.fill_module_(TAVForMAE(ARGS, cfg)).cuda()
    batch, labels = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=16, image=32, vocab=cfg["text"]["vocab"], pad_id=cfg["text"]["pad_id"], nkeep_fusion=4)
    tav, emb, amask, logits, loss = _run_product(pre, model, batch, labels)
    loss.backward()
    assert rel(tav, GOLD[f"{preset}_pre_tav"]) < 1e-4
    assert (emb.cpu().numpy() == GOLD[f"{preset}_pre_tav_embed"]).all()
    assert rel(amask, GOLD[f"{preset}_pre_attention_mask"]) == 0.0
    assert rel(logits, GOLD[f"{preset}_logits"]) < 1e-3
    assert abs(loss.item() - GOLD[f"{preset}_loss"][0]) / GOLD[f"{preset}_loss"][0] < 1e-3
    gn = grad_norm(list(pre.parameters()) + list(model.parameters())).item()
    assert abs(gn - GOLD[f"{preset}_gradnorm"][0]) / GOLD[f"{preset}_gradnorm"][0] < 1e-3
    assert rel(model.linear1.weight.grad[:, :16], GOLD[f"{preset}_grad_linear1"]) < 1e-3


@pytest.mark.parametrize("S,mname", [(8, "none"), (37, "zeros"), (37, "refstyle")])
def test_fusion_encoder_golden_fp32(gpu, S, mname):
    runtime.set_precision("fp32")
    enc = cf.fill_module_(VideoMAEEncoder(dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12), 2)).cuda()
    x = (cf.tensor_for(f"fusion_x{S}", (2, S, 768), kind="bias") * 20).cuda().requires_grad_(True)
    m = torch.zeros(2, 1, 1, S)
    if mname == "refstyle":
        m[..., : S // 4] = O.FP16_MIN
        m[..., S // 4: S // 2] = 65505.0
        m[0, ..., S // 2 - 1] = 1.0
    y = enc(x, None if mname == "none" else m.cuda())
    y.square().mean().backward()
    assert rel(y, GOLD[f"fusion_S{S}_{mname}_y"]) < 1e-4
    # backward of the post-softmax mask (reference utils/TAVFormer.py:372-383 under autograd): input gradient and a weight-gradient corner
    assert rel(x.grad, GOLD[f"fusion_S{S}_{mname}_dx"]) < 1e-3
    assert rel(enc.layer[0].intermediate.dense.weight.grad[:8, :8], GOLD[f"fusion_S{S}_{mname}_dW1_l0"]) < 1e-3


@pytest.mark.parametrize("mname,policy,tol", [("none", "fp32", 1e-4), ("masked", "fp32", 1e-4), ("masked", "bf16", 2e-2)])
def test_fusion_encoder_head_mask_and_attentions(gpu, mname, policy, tol):
    """VideoMAEEncoder.forward(head_mask=, output_attentions=True, output_hidden_states=True, return_dict=False) -- reference
    utils/TAVFormer.py:171-223, :368-375, :389 -- against the oracle's restatement: output, hidden states, the returned probabilities
    (head factor applied before the post-softmax mask) and the gradients through the head-masked context."""
    runtime.set_precision(policy)
    cfg = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12)
    enc = cf.fill_module_(VideoMAEEncoder(cfg, 2)).cuda()
    B, S = 2, 37
    x = (cf.tensor_for("fusion_hm_x", (B, S, 768), kind="bias") * 20).cuda().requires_grad_(True)
    m = None
    if mname == "masked":
        m = torch.zeros(B, 1, 1, S)
        m[..., : S // 4] = -0.03
        m[1, ..., S // 2:] = 0.02
    g = torch.Generator().manual_seed(3)
    hm = torch.rand(2, 1, 12, 1, 1, generator=g) * 1.5              # HF get_head_mask layout: [layers, 1, heads, 1, 1]
    hm[0, 0, 3] = 0.0                                               # a pruned head
    out, hidden, attn = enc(x, None if m is None else m.cuda(), head_mask=hm.cuda(), output_attentions=True, output_hidden_states=True, return_dict=False)
    out.square().mean().backward()
    sd = {k: v.detach().cpu().float() for k, v in enc.state_dict().items()}
    x_ref = x.detach().cpu().float().requires_grad_(True)
    probs = []
    y_ref = O.fusion_encoder({"e." + k: v.requires_grad_(True) for k, v in sd.items()}, "e", x_ref, m, dict(layers=2, heads=12, eps=1e-12), head_mask=hm, probs_out=probs)
    y_ref.square().mean().backward()
    assert rel(out, y_ref) < tol
    assert len(hidden) == 3 and rel(hidden[0], x) == 0.0 and rel(hidden[2], out) == 0.0
    assert len(attn) == 2 and attn[0].shape == (B, 12, S, S)
    for a, pr in zip(attn, probs):
        assert rel(a, pr) < (tol if policy == "fp32" else 5e-2)     # exponentials of bf16 dot products of 20x-scaled inputs
    assert float(attn[0][:, 3].abs().max()) <= (0.03 if m is not None else 0.0) + 1e-6      # the pruned head returns the mask alone
    assert rel(x.grad, x_ref.grad) < 10 * tol
    with pytest.raises(NotImplementedError):
        enc(x, None, head_mask=[torch.ones(1, 12, S, S), None])     # per-query / per-key head masks are not built


@pytest.mark.parametrize("preset", ["A", "B"])
def test_encoder_goldens_fp32(gpu, preset):
    """The three encoders on their own against what the HF modules the reference calls produced (closed-form weights, fp32 policy):
    bert pooled output (models/tav.py:485), wav2vec2 last hidden state (:476), videomae mean over kept tokens (:480-481)."""
    cfg = C.preset(preset + "-tiny")
    runtime.set_precision("fp32")
    model = cf.fill_module_(TAVForMAE(ARGS, cfg)).cuda()
    batch, _ = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=16, image=32, vocab=cfg["text"]["vocab"], pad_id=cfg["text"]["pad_id"], nkeep_fusion=4)
    with torch.no_grad():
        _, pooled = model.bert(batch["input_ids"].cuda(), batch["text_mask"].cuda())
        aud, _, Sa = model.wav2vec2(batch["audio_features"].cuda())
        vid, Sv = model.videomae(batch["video_embeds"].cuda(), batch["visual_mask"].cuda())
    assert rel(pooled, GOLD[f"{preset}_text_pooled"]) < 1e-4
    assert rel(aud.view(2, Sa, -1), GOLD[f"{preset}_audio_last"]) < 1e-4
    assert rel(vid.view(2, Sv, -1).mean(1), GOLD[f"{preset}_video_mean"]) < 1e-4


@pytest.mark.parametrize("preset,policy,tol", [("B", "fp32", 1e-3), ("A", "fp32", 1e-3), ("B", "bf16", 1e-2), ("A", "bf16", 1e-2)])
def test_parity_vs_oracle_random_weights(gpu, preset, policy, tol):
    """Seeded random weights (BASELINE.md §3 protocol): logits, loss, global grad-norm and per-tensor gradients vs the oracle."""
    cfg = C.preset(preset + "-tiny")
    runtime.set_precision(policy)
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)
    (tx, au, vi), lab = synthetic.make_batch(cfg, 2, s_text=32, t_audio=16000, n_visual_true=8)
    batch = _as_batch(tx, au, vi)
    sdp = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
    sdm = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    o_logits, o_loss = O.tav_step(sdm, sdp, cfg, batch, lab.long())
    o_loss.backward()
    pre.cuda()
    model.cuda()
    _, _, _, logits, loss = _run_product(pre, model, batch, lab)
    loss.backward()
    assert rel(logits, o_logits) < tol
    assert abs(loss.item() - o_loss.item()) / abs(o_loss.item()) < tol
    o_gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in list(sdp.values()) + list(sdm.values()) if v.requires_grad and v.grad is not None)).item()
    gn = grad_norm(list(pre.parameters()) + list(model.parameters())).item()
    assert abs(gn - o_gn) / o_gn < tol
    gmax = max(v.grad.abs().max().item() for v in list(sdp.values()) + list(sdm.values()) if getattr(v, "grad", None) is not None)
    worst = 0.0
    for mod, sdo in ((pre, sdp), (model, sdm)):
        for k, p in mod.named_parameters():
            og = sdo[k].grad
            assert (p.grad is None) == (og is None), f"gradient presence differs for {k}"     # same set of trained parameters as the reference
            if og is not None:
                worst = max(worst, (p.grad.detach().cpu() - og).abs().max().item() / (og.abs().max().item() + 1e-3 * gmax))
    assert worst < (3e-2 if policy == "bf16" else 1e-3), worst


def test_full_size_properties_bf16(gpu):
    """BASELINE-sized inputs (text 128, audio 80000, video 16x3x224x224, 104/1464 video tokens), preset B depth-reduced to fit the test budget:
    (1) deterministic (bitwise) across runs, (2) utterances are independent: batch of 4 == two batches of 2 (what DP sharding relies on),
    (3) permuting the batch permutes the logits."""
    cfg = C.preset("B")
    for k in ("text", "audio", "video", "fusion"):
        cfg[k]["layers"] = 2
    runtime.set_precision("bf16")
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)
    pre.cuda()
    model.cuda()
    (tx, au, vi), lab = synthetic.make_batch(cfg, 4, device="cuda")
    au["attention_mask"][:] = 1            # equal audio lengths so that padding does not couple the split batches
    batch = _as_batch(tx, au, vi)

    def fwd(sel):
        b = {k: v[sel] for k, v in batch.items()}
        with torch.no_grad():
            return _run_product(pre, model, b, lab[sel])[3]

    full = fwd(torch.arange(4))
    again = fwd(torch.arange(4))
    assert torch.equal(full, again)
    halves = torch.cat([fwd(torch.tensor([0, 1])), fwd(torch.tensor([2, 3]))])
    assert rel(halves, full) < 2e-3
    perm = torch.tensor([2, 0, 3, 1])
    assert rel(fwd(perm), full[perm]) < 2e-3
    assert torch.isfinite(full).all()


def test_dropout_train_mode(gpu):
    cfg = C.preset("B-tiny")
    runtime.set_precision("fp32")
    pre, model = PreFormer(cfg).cuda(), TAVForMAE(ARGS, cfg).cuda()
    (tx, au, vi), lab = synthetic.make_batch(cfg, 2, s_text=16, t_audio=8000, n_visual_true=4, device="cuda")
    batch = _as_batch(tx, au, vi)
    a = _run_product(pre, model, batch, lab, check="train")[3]
    b = _run_product(pre, model, batch, lab, check="train")[3]
    c = _run_product(pre, model, batch, lab, check="val")[3]
    assert not torch.equal(a, b) and torch.isfinite(a).all() and torch.isfinite(c).all()      # fresh mask per call (models/tav.py:497-498)


def test_fused_adamw_matches_torch(gpu):
    torch.manual_seed(0)
    shapes = [(768, 768), (3072,), (7, 3072), (5,), (50, 3, 10)]
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref = torch.optim.AdamW(qs, lr=1e-3, weight_decay=1e-2)
    opt = FusedAdamW(ps, lr=1e-3, weight_decay=1e-2)
    for step in range(3):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p) * (3.0 if step == 1 else 0.1)
            p.grad, q.grad = g.clone(), g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(qs, 1.0)
        ref.step()
        n = opt.clip_and_step(1.0)
        assert abs(n.item() - n_ref.item()) / n_ref.item() < 1e-5
        for p, q in zip(ps, qs):
            assert rel(p, q) < 1e-5


def test_fused_adamw_bucket_norm_and_misaligned_pieces(gpu):
    """Round 4, the two properties the sharded optimizer rests on.  (a) FusedAdamW.norm_buffers: the clip norm taken over flat gradient buckets
    (chunks counted from each bucket's start) is the per-parameter norm up to summation order, and tav_sum_partials over the exchanged partial
    array reproduces tav_sumsq_chunked's own second stage bit for bit.  (b) adamw_chunk_kernel computes the same bits whether an element falls
    into its 16-byte or its scalar loop: a tensor updated whole equals the same tensor updated as two pieces cut at an odd element."""
    import ctypes as Ct
    from tav_amd._lib import check, lib, ptr, stream
    from tav_amd.optim import bucket_norm_tables
    torch.manual_seed(1)
    shapes = [(768, 768), (3072,), (7, 3072), (5,), (50, 3, 10), (40000,)]
    n_all = sum(int(np.prod(s)) for s in shapes)
    flat = torch.randn(n_all + len(shapes), device="cuda")
    ps, off = [], 0
    for s_ in shapes:
        p = torch.nn.Parameter(torch.randn(s_, device="cuda"))
        k = p.numel()
        p.grad = flat[off:off + k].view_as(p)
        off += k
        ps.append(p)
    a, b = FusedAdamW(ps, lr=1e-3), FusedAdamW([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
    for q, p in zip(b.params, ps):
        q.grad = p.grad
    b.norm_buffers = [flat[:n_all]]
    na, nb = a.clip_and_step(1.0).item(), b.clip_and_step(1.0).item()
    assert abs(na - nb) / na < 1e-6 and abs(na - flat[:n_all].double().norm().item()) / na < 1e-6
    for p, q in zip(ps, b.params):
        assert rel(p, q) < 1e-6
    # second stage alone == second stage inside tav_sumsq_chunked
    _, t_g, t_s, t_c, nbuf, nch, part = bucket_norm_tables([flat[:n_all]], int(lib().tav_optim_chunk_elems()))
    out1, out2 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    check(lib().tav_sumsq_chunked(ptr(t_g), ptr(t_s), ptr(t_c), nbuf, nch, ptr(part), ptr(out1), stream()), "sumsq_chunked")
    check(lib().tav_sum_partials(ptr(part), nch, ptr(out2), stream()), "sum_partials")
    assert torch.equal(out1, out2) and nch == (n_all + 16383) // 16384
    # (b) whole tensor vs two pieces cut at element 16384 + 1 (the second piece starts 4 bytes off a 16-byte boundary)
    n, cut = 3 * 16384 + 11, 16384 + 1
    w0, g0 = torch.randn(n, device="cuda"), torch.randn(n, device="cuda") * 1e-3
    whole = FusedAdamW([torch.nn.Parameter(w0.clone())], lr=1e-3, weight_decay=1e-2)
    whole.params[0].grad = g0.clone()
    wp = torch.nn.Parameter(w0.clone())
    pieces = [torch.nn.Parameter(wp.detach()[:cut]), torch.nn.Parameter(wp.detach()[cut:])]
    split = FusedAdamW(pieces, lr=1e-3, weight_decay=1e-2)
    for _ in range(3):
        whole.params[0].grad = g0.clone()
        pieces[0].grad, pieces[1].grad = g0[:cut].clone(), g0.clone()[cut:]       # (the second gradient view is misaligned too)
        whole.clip_and_step(None)
        split.clip_and_step(None)
    torch.cuda.synchronize()
    assert torch.equal(whole.params[0].detach(), wp.detach())
    assert torch.equal(whole.state[whole.params[0]][0][cut:], split.state[pieces[1]][0])


def test_grad_accum_second_step_under_both_zero_grad_readings(gpu):
    """reference train_model/tav_train.py:96-106: step, zero_grad, then -- at a dialogue end -- a second unclipped optimizer.step().  Under torch 1.10
    (the reference's pin) zero_grad zero-fills and that step decays the weights and moves them along the momentum; under torch >= 2 it is a no-op.
    The fused optimizer reproduces torch.optim.AdamW in both readings."""
    torch.manual_seed(0)
    shapes = [(300, 64), (77,), (7, 128)]
    for like_1_10 in (True, False):
        ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        ref = torch.optim.AdamW(qs, lr=1e-2, weight_decay=1e-1)
        opt = FusedAdamW(ps, lr=1e-2, weight_decay=1e-1)
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_norm_(qs, 1.0)
        ref.step()
        ref.zero_grad(set_to_none=not like_1_10)
        opt.clip_and_step(1.0)
        opt.zero_grad(set_to_none=not like_1_10)
        before = [p.detach().clone() for p in ps]
        ref.step()                                   # the dialogue-end step: no clipping (:103)
        opt.clip_and_step(None)
        for p, q, b in zip(ps, qs, before):
            assert rel(p, q) < 1e-5
            moved = (p.detach() - b).abs().max().item()
            assert (moved > 1e-4) if like_1_10 else (moved == 0.0)


@pytest.mark.parametrize("policy,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_transformer_encoder_vs_oracle_and_golden(gpu, policy, tol):
    """Row T1 (alternative fusion stack): forward vs the golden made from the reference class, forward+backward vs the oracle."""
    runtime.set_precision(policy)
    te = TransformerEncoder(768, num_layers=1).eval()
    if policy == "fp32":
        cf.fill_module_(te)                     # closed-form weights: comparable with the committed golden
    else:
        synthetic.seeded_init_(te, 3)           # bf16: seeded random weights (closed-form ones are ill-conditioned)
    sd = {"t." + k: v.detach().clone().requires_grad_(True) for k, v in te.state_dict().items()}
    te.cuda()
    x = cf.tensor_for("te_x", (2, 11, 768), kind="bias") * 20
    m = torch.zeros(2, 1, 1, 11)
    m[..., 8:] = O.FP16_MIN
    xg = x.cuda().requires_grad_(True)
    y = te(xg, m.cuda())
    if policy == "fp32":
        assert rel(y, GOLD["transformer_encoder_early0_y"]) < 1e-3
    xo = x.clone().requires_grad_(True)
    yo = O.transformer_encoder(sd, "t", xo, m, 1, 12, False)
    assert rel(y, yo) < tol
    w = cf.tensor_for("te_w", (2, 11, 768), kind="bias")
    (y * w.cuda()).sum().backward()
    (yo * w).sum().backward()
    assert rel(xg.grad, xo.grad) < tol
    gmax = max(v.grad.abs().max().item() for v in sd.values())
    for k, p in te.named_parameters():          # error relative to the tensor's own scale, floored by 1e-3 of the largest gradient
        og = sd["t." + k].grad
        assert (p.grad.detach().cpu() - og).abs().max().item() / (og.abs().max().item() + 1e-3 * gmax) < 5 * tol, k
    # training mode: dropout active (the reference never .eval()s this module), deterministic per seed, different across calls
    te.train()
    a, b = te(xg.detach(), m.cuda()), te(xg.detach(), m.cuda())
    assert torch.isfinite(a).all() and not torch.equal(a, b)


# ---- SURVEY.md §8(f) rows 1-2 --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["textcls", "textaudio"])
@pytest.mark.parametrize("preset", ["A", "B"])
def test_single_and_dual_models_golden_fp32(gpu, preset, tag):
    """Text-only classifier (reference SingleModels/models/text.py) and the text+audio dual classifier on libtavhip, fp32 policy,
    closed-form weights: logits / loss / grad-norm against the goldens made from the HF modules the reference calls."""
    from tav_amd.DoubleModels.models.text_audio import BertAudioClassifier
    from tav_amd.SingleModels.models.text import BertClassifier
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset(preset + "-tiny")
    runtime.set_precision("fp32")
    batch, labels = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=16, image=32, vocab=cfg["text"]["vocab"], pad_id=cfg["text"]["pad_id"], nkeep_fusion=4)
    args = dict(output_dim=7, dropout=0.5)
    model = cf.fill_module_(BertClassifier(args, config=cfg) if tag == "textcls" else BertAudioClassifier(args, config=cfg)).cuda()
    if tag == "textcls":
        logits = model(batch["input_ids"], batch["text_mask"], "val")
    else:
        logits = model(batch["input_ids"], batch["text_mask"], batch["audio_features"], check="val")
    loss = CrossEntropyLoss()(logits, labels)
    loss.backward()
    gn = grad_norm(list(model.parameters())).item()
    assert rel(logits.detach().cpu(), torch.as_tensor(GOLD[f"{preset}_{tag}_logits"])) < 1e-3
    assert abs(loss.item() - GOLD[f"{preset}_{tag}_loss"][0]) / GOLD[f"{preset}_{tag}_loss"][0] < 1e-3
    assert abs(gn - GOLD[f"{preset}_{tag}_gradnorm"][0]) / GOLD[f"{preset}_{tag}_gradnorm"][0] < 1e-3


@pytest.mark.parametrize("tag", ["textcls", "textaudio"])
def test_single_and_dual_models_bf16_vs_oracle(gpu, tag):
    """bf16 policy, seeded-random weights, 10 s audio for the dual model's long-sequence audio attention (Sa = 499
    at T = 160000, BASELINE config 4): logits and loss within 1e-2 of the CPU oracle; dropout active under check="train"."""
    from tav_amd.DoubleModels.models.text_audio import BertAudioClassifier
    from tav_amd.SingleModels.models.text import BertClassifier
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset("B")
    cfg["text"]["layers"] = 2
    cfg["audio"]["layers"] = 2
    runtime.set_precision("bf16")
    args = dict(output_dim=7, dropout=0.5)
    model = BertClassifier(args, config=cfg) if tag == "textcls" else BertAudioClassifier(args, config=cfg)
    synthetic.seeded_init_(model, 3)
    (tx, au, _), lab = synthetic.make_batch(cfg, 2, s_text=128, t_audio=160000 if tag == "textaudio" else 8000, n_visual_true=104)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        if tag == "textcls":
            o_logits = O.text_classifier_forward(sd, cfg, tx["input_ids"], tx["attention_mask"])
        else:
            o_logits = O.text_audio_forward(sd, cfg, tx["input_ids"], tx["attention_mask"], au["audio_features"])
        o_loss = torch.nn.functional.cross_entropy(o_logits, lab.long())
    model.cuda()

    def run(check):
        if tag == "textcls":
            return model(tx["input_ids"], tx["attention_mask"], check)
        return model(tx["input_ids"], tx["attention_mask"], au["audio_features"], check=check)

    logits = run("val")
    loss = CrossEntropyLoss()(logits, lab)
    loss.backward()
    assert rel(logits.detach().cpu(), o_logits) < 1e-2
    assert abs(loss.item() - o_loss.item()) / o_loss.item() < 1e-2
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) > 30 and all(torch.isfinite(g).all() for g in grads)
    a, b = run("train"), run("train")
    assert not torch.equal(a, b)


def test_entrypoints_run_one_tiny_epoch(gpu, capsys):
    """tav_nn.py / SingleModels/text_nn.py / DoubleModels/text_audio_nn.py mains (reference entrypoints of the same names): one epoch on
    synthetic utterances with the tiny preset; losses must be finite."""
    import tav_amd.tav_nn as tav_nn
    from tav_amd.DoubleModels import text_audio_nn
    from tav_amd.SingleModels import text_nn
    argv = ["--preset", "B-tiny", "--epoch", "1", "--batch_size", "2", "--synthetic", "4", "--dtype", "bf16", "--loss", "CrossEntropy"]
    try:
        for mod in (text_nn, text_audio_nn, tav_nn):
            mod.main(argv)
            out = capsys.readouterr().out
            assert "nan" not in out.lower()
    finally:
        C.set_default_preset("A")


def test_rccl_reducer_single_rank(gpu, monkeypatch):
    """The bucketed gradient all-reduce on the real RCCL backend with ONE rank (the only multi-process GPU test a one-GPU box allows):
    hooks, bucket planning, side stream, AVG collective, re-pointed .grad views.  Loss and gradients must equal the plain path
    (bitwise, except the scatter-add embedding tables), over two steps (the first plans the buckets, the second overlaps them)."""
    import torch.distributed as dist
    from tav_amd.train_model.tav_train import TrainStep
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset("B-tiny")
    runtime.set_precision("bf16")
    inp, lab = synthetic.make_batch(cfg, 2, s_text=16, t_audio=8000, n_visual_true=4, device="cuda")

    def run(ddp):
        torch.manual_seed(0)                  # seeded_init_ scales by the std of the default init: make that identical too
        pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
        synthetic.seeded_init_(pre, 1)
        synthetic.seeded_init_(model, 2)
        pre.cuda()
        model.cuda()
        monkeypatch.setenv("TAV_DDP_SINGLE_RANK", "1" if ddp else "0")
        stepper = TrainStep(model, pre, CrossEntropyLoss(), lr=1e-4, bucket_mb=1.0)
        assert (stepper.reducer is not None) == bool(ddp)
        if ddp == "manual":                   # graph-mode reducer (bench.py with N > 1): pack after backward, collectives issued by the caller
            stepper.reducer.set_manual(True, bucket_mb=4.0)
        out = []
        for _ in range(2):
            loss = stepper.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4)
            if ddp == "manual":
                stepper.reducer.pack_all()
                stepper.reducer.reduce_packed()
            torch.cuda.synchronize()
            out.append((loss.item(), {k: p.grad.clone() for k, p in list(model.named_parameters()) + list(pre.named_parameters()) if p.grad is not None}))
            stepper.update()
        if stepper.reducer is not None:
            assert len(stepper.reducer.buckets) > 3
            stepper.reducer.remove()
        return out

    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        plain, ddp, manual = run(False), run(True), run("manual")
    finally:
        if created:
            dist.destroy_process_group()
    for step, ((l0, g0), (l1, g1)) in enumerate(list(zip(plain, ddp)) + list(zip(plain, manual))):
        step = step % 2
        assert g0.keys() == g1.keys()
        if step == 0:                         # identical weights: everything but the atomically accumulated embedding tables is bitwise equal
            assert l0 == l1
            for k in g0:
                if "embedding" in k:
                    assert rel(g0[k], g1[k]) < 1e-5, k
                else:
                    assert torch.equal(g0[k], g1[k]), k
        else:                                 # after one update (embedding-table rounding differs in the last bit): equal to rounding
            assert abs(l0 - l1) / abs(l0) < 1e-4
            gmax = max(v.abs().max().item() for v in g0.values())
            for k in g0:                      # (floor: a key bias has a zero true gradient -- the softmax is invariant to it -- and holds rounding noise only)
                e = (g0[k].double() - g1[k].double()).abs().max().item() / (g0[k].abs().max().item() + 1e-3 * gmax)
                assert e < 2e-2, (k, e)


def test_best_pt_resume_continues(gpu, tmp_path):
    """a run interrupted after two steps and resumed from best.pt (model, PreFormer, AdamW moments + step count) takes
    the same third step as the uninterrupted run."""
    from tav_amd.train_model.tav_train import CosineWarmRestarts, TrainStep
    from tav_amd.utils import global_functions as G
    cfg = C.preset("B-tiny")
    runtime.set_precision("fp32")
    inp, lab = synthetic.make_batch(cfg, 2, s_text=16, t_audio=8000, n_visual_true=4, device="cuda")

    def build():
        torch.manual_seed(0)
        pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
        synthetic.seeded_init_(pre, 1)
        synthetic.seeded_init_(model, 2)
        return pre.cuda(), model.cuda()

    def steps(stepper, n):
        for _ in range(n):
            stepper.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4)
            stepper.update()

    crit = G.CrossEntropyLoss()
    pre_a, model_a = build()
    a = TrainStep(model_a, pre_a, crit, lr=1e-3, weight_decay=1e-2)
    steps(a, 3)
    pre_b, model_b = build()
    b = TrainStep(model_b, pre_b, crit, lr=1e-3, weight_decay=1e-2)
    steps(b, 2)
    G.save_model(model_b, pre_b, b.opt, crit, CosineWarmRestarts(b.opt, 2), 0, 1, str(tmp_path), 2400)
    pre_c, model_c = build()
    c = TrainStep(model_c, pre_c, crit, lr=7.0, weight_decay=0.0)
    G.load_model(model_c, pre_c, c.opt, crit, str(tmp_path))
    assert c.opt.step_count == 2 and c.opt.lr == 1e-3
    steps(c, 1)
    worst = 0.0
    for (k, pa), (_, pc) in zip(list(model_a.named_parameters()) + list(pre_a.named_parameters()), list(model_c.named_parameters()) + list(pre_c.named_parameters())):
        worst = max(worst, rel(pc, pa))         # not bitwise: the embedding tables' atomic scatter-add differs in the last bit from run to run
    assert worst < 2e-4
    # and the resumed state matters: stepping from the checkpointed weights with FRESH moments lands somewhere else
    pre_d, model_d = build()
    d = TrainStep(model_d, pre_d, crit, lr=1e-3, weight_decay=1e-2)
    G.load_model(model_d, pre_d, None, crit, str(tmp_path))
    steps(d, 1)
    assert max(rel(pd, pa) for (_, pa), (_, pd) in zip(model_a.named_parameters(), model_d.named_parameters())) > 2e-3


# ---- full depth, full input size: BASELINE configs 2 and 4 -----------------------------------------------------
_ORACLE_CACHE = {}


def _oracle_full(preset, seed=0, batch_size=2, cfg=None, tag=None):
    """Oracle fwd + loss + bwd at the benchmark's sizes (text 128, audio 80000, video 16x3x224x224 with 104/1464 tokens), every encoder at its
    full depth: computed once per (preset, seed, batch) -- a few seconds of CPU at batch 2 -- and shared by the policies that are compared with it.
    `seed` moves the weights AND the inputs (seed 0 = the round-1/2 case)."""
    key = (tag or preset, seed, batch_size)
    if key not in _ORACLE_CACHE:
        cfg = C.preset(preset) if cfg is None else cfg
        torch.manual_seed(seed)
        pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
        synthetic.seeded_init_(pre, 1 + 10 * seed)
        synthetic.seeded_init_(model, 2 + 10 * seed)
        nvt = 104 * cfg["video"]["frames"] // 16 + (1 if cfg["video"]["frames"] == 32 else 0)     # 104 of 1568, 209 of 3136
        (tx, au, vi), lab = synthetic.make_batch(cfg, batch_size, seed=1234 + seed, n_visual_true=nvt)   # reference-style masks: {0,-65504} text, {65505,1} audio (row 0 padded)
        batch = _as_batch(tx, au, vi)
        sdp = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
        sdm = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
        o_logits, o_loss = O.tav_step(sdm, sdp, cfg, batch, lab.long())
        o_loss.backward()
        grads = {("pre", k): v.grad for k, v in sdp.items() if v.requires_grad and v.grad is not None}
        grads.update({("model", k): v.grad for k, v in sdm.items() if v.requires_grad and v.grad is not None})
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
        state = ({k: v.detach() for k, v in sdp.items()}, {k: v.detach() for k, v in sdm.items()})
        _ORACLE_CACHE[key] = (cfg, batch, lab, state, o_logits.detach(), o_loss.item(), gn, grads)
    return _ORACLE_CACHE[key]


def _compare_with_oracle(oracle, policy, tol, label, grad_tol=None, check_optimizer=True):
    """Product (current precision policy = `policy`) against one `_oracle_full` record: logits, loss, clip_grad_norm_ value within `tol`,
    every parameter gradient compared (same set of trained parameters).  Returns (logits, loss, grad-norm, worst-tensor) errors."""
    cfg, batch, lab, (sdp, sdm), o_logits, o_loss, o_gn, o_grads = oracle
    runtime.set_precision(policy)
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    pre.load_state_dict(sdp)
    model.load_state_dict(sdm)
    pre.cuda()
    model.cuda()
    tav, emb, amask, logits, loss = _run_product(pre, model, batch, lab)
    assert amask.abs().max().item() > 6e4                                      # the reference-style mask really is in play (fp16 min / 65505)
    loss.backward()
    torch.cuda.synchronize()
    e_logits, e_loss = rel(logits, o_logits), abs(loss.item() - o_loss) / abs(o_loss)
    params = list(pre.parameters()) + list(model.parameters())
    gn = grad_norm(params).item()
    e_gn = abs(gn - o_gn) / o_gn
    gmax = max(g.abs().max().item() for g in o_grads.values())
    worst, worst_k, n_cmp, table = 0.0, None, 0, []
    for tag, mod in (("pre", pre), ("model", model)):
        for k, p in mod.named_parameters():
            og = o_grads.get((tag, k))
            assert (p.grad is None) == (og is None), f"gradient presence differs for {tag}.{k}"
            if og is not None:
                ae = (p.grad.detach().cpu() - og).abs().max().item()
                e = ae / (og.abs().max().item() + 1e-3 * gmax)
                table.append((e, f"{tag}.{k}", og.abs().max().item(), ae))
                n_cmp += 1
                if e > worst:
                    worst, worst_k = e, f"{tag}.{k}"
    for e, k, m, ae in sorted(table, reverse=True)[:6]:
        print(f"    {k:80s} err {e:.2e}  |ref|max {m:.3e}  abs err {ae:.3e}  (gmax {gmax:.3e})")
    print(f"[{label} {policy}] logits {e_logits:.2e} loss {e_loss:.2e} grad-norm {e_gn:.2e} worst tensor {worst:.2e} ({worst_k}), {n_cmp} gradients")
    assert e_logits < tol and e_loss < tol and e_gn < tol, (e_logits, e_loss, e_gn)
    assert worst < (grad_tol if grad_tol is not None else (5e-2 if policy != "fp32" else 1e-3)), (worst, worst_k)
    if check_optimizer:
        opt = FusedAdamW(params, lr=1e-6, weight_decay=1e-4)
        n = opt.clip_and_step(1.0)                                             # train_model/tav_train.py:61-62 on the same gradients
        assert abs(n.item() - o_gn) / o_gn < tol
    return e_logits, e_loss, e_gn, worst


@pytest.mark.parametrize("preset,policy,tol", [("B", "fp32", 1e-3), ("B", "bf16", 1e-2), ("A", "fp32", 1e-3), ("A", "bf16", 1e-2)])
def test_full_depth_full_size_parity(gpu, preset, policy, tol):
    """BASELINE config 2 at FULL depth (12-layer fusion stack hard-coded at reference models/tav.py:441-442, 12/12/12 or 6/24/12 encoder layers) and
    full input sizes, against the CPU oracle on identical seeded weights and inputs: logits, loss, clip_grad_norm_ value within the north_star
    tolerance, every parameter gradient compared (same set of trained parameters), the fused clip+AdamW reporting the same norm."""
    _compare_with_oracle(_oracle_full(preset), policy, tol, f"full-depth {preset}")


def test_config4_text_audio_10s_full_depth(gpu):
    """BASELINE config 4: text+audio dual classifier, 10 s audio (T = 160000 -> 499 frames: the long-sequence audio attention), batch 16, every
    encoder at full depth (preset B), bf16 policy: logits, loss and gradient norm against the CPU oracle."""
    from tav_amd.DoubleModels.models.text_audio import BertAudioClassifier
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset("B")
    runtime.set_precision("bf16")
    torch.manual_seed(0)
    model = BertAudioClassifier(dict(output_dim=7, dropout=0.5), config=cfg)
    synthetic.seeded_init_(model, 3)
    (tx, au, _), lab = synthetic.make_batch(cfg, 16, s_text=128, t_audio=160000, with_video=False)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    o_logits = O.text_audio_forward(sd, cfg, tx["input_ids"], tx["attention_mask"], au["audio_features"])
    o_loss = torch.nn.functional.cross_entropy(o_logits, lab.long())
    o_loss.backward()
    o_gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.requires_grad and v.grad is not None)).item()
    model.cuda()
    logits = model(tx["input_ids"], tx["attention_mask"], au["audio_features"], check="val")
    loss = CrossEntropyLoss()(logits, lab)
    loss.backward()
    gn = grad_norm([p for p in model.parameters() if p.grad is not None]).item()
    e = (rel(logits, o_logits.detach()), abs(loss.item() - o_loss.item()) / abs(o_loss.item()), abs(gn - o_gn) / o_gn)
    print(f"[config 4, b=16, T=160000, 12+12 layers, bf16] logits {e[0]:.2e} loss {e[1]:.2e} grad-norm {e[2]:.2e}")
    assert max(e) < 1e-2, e
    for k, p in model.named_parameters():
        assert (p.grad is None) == (sd[k].grad is None), k


def test_backward_is_bitwise_deterministic(gpu):
    """run the same fwd+bwd twice (branches on their own streams) and compare every gradient bit for bit -- no kernel on the path
    accumulates with atomics (the embedding-table gradients are segment sums, the weight gradients fixed-order slab sums)."""
    cfg = C.preset("B")
    for k in ("text", "audio", "video", "fusion"):
        cfg[k]["layers"] = 2
    runtime.set_precision("bf16")
    torch.manual_seed(0)
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)
    pre.cuda()
    model.cuda()
    (tx, au, vi), lab = synthetic.make_batch(cfg, 4, device="cuda")
    batch = _as_batch(tx, au, vi)
    runs = []
    for _ in range(2):
        for p in list(pre.parameters()) + list(model.parameters()):
            p.grad = None
        _, _, _, logits, loss = _run_product(pre, model, batch, lab)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), {k: p.grad.clone() for k, p in list(model.named_parameters()) + list(pre.named_parameters()) if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1].keys() == runs[1][1].keys() and len(runs[0][1]) > 100
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_unequal_visual_rows_raise_not_fault(gpu):
    """A visual mask whose rows keep different numbers of tokens (what the reference's collate can emit at batch > 1, models/tav.py:206-217):
    a Python error when the host counts, and no device fault when the caller vouches for the count (n_visual_true) but is wrong."""
    cfg = C.preset("B-tiny")
    runtime.set_precision("bf16")
    pre, model = PreFormer(cfg).cuda(), TAVForMAE(ARGS, cfg).cuda()
    (tx, au, vi), lab = synthetic.make_batch(cfg, 2, s_text=16, t_audio=8000, n_visual_true=4, device="cuda")
    vi["attention_mask"][1, :] = False
    vi["attention_mask"][1, :2] = True                                          # row 1: 2 True instead of 4 (the total, 6, still divides by 2)
    batch = _as_batch(tx, au, vi)
    with pytest.raises(ValueError, match="same number"):
        _run_product(pre, model, batch, lab)
    tav, emb, amask = pre(input_ids=batch["input_ids"], audio_features=batch["audio_features"], video_embeds=batch["video_embeds"], text_mask=batch["text_mask"],
                          audio_mask=batch["audio_mask"], visual_mask=batch["visual_mask"], device="cuda", train=False, n_visual_true=4)
    logits = model(batch["input_ids"], batch["text_mask"], batch["audio_features"], batch["video_embeds"], batch["visual_mask"], tav, emb, amask,
                   batch_size=2, check="val", n_visual_true=4)
    torch.cuda.synchronize()                                                   # would surface a memory fault
    assert torch.isfinite(logits).all()


def test_specaugment_on_device_is_capturable(gpu):
    """PreFormer(train=True): the SpecAugment time mask (reference models/tav.py:269-306) is drawn on the device -- no host read -- so the call can
    be captured into a hipGraph; each replay draws a fresh mask (graph-safe Philox offsets), rows keep their span budget (2 spans of 10 frames at
    249 frames) and padding frames are never chosen as span starts."""
    cfg = C.preset("B-tiny")
    runtime.set_precision("bf16")
    torch.manual_seed(0)
    pre = PreFormer(cfg).cuda()
    (tx, au, vi), _ = synthetic.make_batch(cfg, 3, s_text=16, t_audio=80000, n_visual_true=4, device="cuda")
    kw = dict(input_ids=tx["input_ids"], audio_features=au["audio_features"], video_embeds=vi["visual_embeds"], text_mask=tx["attention_mask"],
              audio_mask=au["attention_mask"], visual_mask=vi["attention_mask"], device="cuda", n_visual_true=4)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        clean, _, _ = pre(train=False, **kw)
        pre(train=True, **kw)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with runtime.capture(g, s):
            noisy, _, _ = pre(train=True, **kw)
        outs = []
        for _ in range(2):
            g.replay()
            torch.cuda.synchronize()
            outs.append(noisy.clone())
    St, Sa = 16, 249
    for o in outs:                                           # every row got masked frames (a span also moves its neighbours through the positional conv)
        assert (o[:, St:St + Sa] != clean[:, St:St + Sa]).any(-1).any(dim=1).all() and torch.isfinite(o).all()
    assert not torch.equal(outs[0], outs[1])                                                # a new draw per replay
    assert torch.equal(outs[0][:, :St], clean[:, :St]) and torch.equal(outs[0][:, St + Sa:], clean[:, St + Sa:])   # text / video tokens untouched


def test_collate_device_feeds_captured_step(gpu):
    """collate_batch_device builds the batch ON the GPU (padding, audio length mask, equal-count video token mask) and its
    output drives PreFormer + TAVForMAE + loss + backward captured into ONE hipGraph; replaying the graph on a second collated batch (copied into
    the captured input buffers) reproduces the eager result on that batch."""
    from tav_amd.models.tav import collate_batch_device
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset("B-tiny")
    runtime.set_precision("bf16")
    torch.manual_seed(0)
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)
    pre.cuda()
    model.cuda()
    crit = CrossEntropyLoss()

    def items(seed, lens):
        g = torch.Generator().manual_seed(seed)
        out = []
        for L in lens:
            ids = torch.randint(3, cfg["text"]["vocab"], (16,), generator=g)
            out.append(([{"input_ids": ids, "attention_mask": torch.ones(16)}, torch.randn(L, generator=g) * 0.1, torch.randn(16, 3, 32, 32, generator=g)],
                        float(torch.randint(0, 7, (1,), generator=g))))
        return out

    gen = torch.Generator(device="cuda").manual_seed(7)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        inp, lab = collate_batch_device(items(1, [8000, 6400]), "train", device="cuda", n_visual_true=4, generator=gen)
        assert inp[1]["audio_features"].is_cuda and inp[2]["attention_mask"].sum(1).tolist() == [4, 4]
        assert inp[1]["attention_mask"].sum(1).tolist() == [8000.0, 6400.0]

        def step():
            loss = __import__("tav_amd.train_model.tav_train", fromlist=["get_statistics"]).get_statistics(inp, lab, model, pre, crit, None, check="val", epoch=0, n_visual_true=4)
            loss.backward()
            return loss

        for _ in range(2):
            for p in list(pre.parameters()) + list(model.parameters()):
                p.grad = None
            step()
        torch.cuda.synchronize()
        for p in list(pre.parameters()) + list(model.parameters()):
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with runtime.capture(graph, s):
            static_loss = step()
        inp2, lab2 = collate_batch_device(items(2, [7000, 8000]), "train", device="cuda", n_visual_true=4, generator=gen)
        for d, d2 in zip(inp, inp2):
            for k in d:
                d[k].copy_(d2[k])
        lab.copy_(lab2)
        graph.replay()
        torch.cuda.synchronize()
        replayed = static_loss.item()
        g_replayed = model.linear1.weight.grad.clone()
        for p in list(pre.parameters()) + list(model.parameters()):
            p.grad = None
        eager = step()
        torch.cuda.synchronize()
    assert abs(replayed - eager.item()) <= 1e-6 * abs(eager.item())
    assert torch.equal(g_replayed, model.linear1.weight.grad)


@pytest.mark.parametrize("mode", ["chain", "single"])
def test_graphed_ddp_step_single_rank(gpu, monkeypatch, mode):
    """mode "single" (round 4): the same step as ONE hipGraph, the bucket all-reduces captured as raw RCCL calls through the C ABI on the reducer
    branch -- same assertions.  mode "chain":
    bench.py's data-parallel step (ddp.GraphedStep: forward + backward cut into segments, one hipGraph each, the bucket all-reduces issued
    eagerly on the reducer stream in between, optimizer graph last) on the real RCCL backend with ONE rank.  Learning rate 0 keeps the weights
    fixed, so every replay must reproduce the plain eager step bit for bit: same loss, and every gradient -- read back from the packed,
    all-reduced buckets the optimizer graph consumes -- equal to the plain backward's; the optimizer graph itself must have run (step counter)."""
    import torch.distributed as dist
    from tav_amd import engine
    from tav_amd.ddp import GraphedStep
    from tav_amd.train_model.tav_train import TrainStep
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset("B")
    for k in ("text", "audio", "video", "fusion"):
        cfg[k]["layers"] = 4
    cfg["video"]["image"] = 32
    runtime.set_precision("bf16")

    def build(ddp):
        torch.manual_seed(0)
        pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
        synthetic.seeded_init_(pre, 1)
        synthetic.seeded_init_(model, 2)
        pre.cuda()
        model.cuda()
        monkeypatch.setenv("TAV_DDP_SINGLE_RANK", "1" if ddp else "0")
        return pre, model, TrainStep(model, pre, CrossEntropyLoss(), lr=0.0, weight_decay=1e-2, clip=1.0)

    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            inp, lab = synthetic.make_batch(cfg, 2, s_text=16, t_audio=8000, n_visual_true=4, device="cuda")
            pre_a, model_a, plain = build(False)
            loss_a = plain.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4).item()
            ref = [None if p.grad is None else p.grad.clone() for p in plain.params]
            pre_b, model_b, st = build(True)
            assert st.reducer is not None
            # one eager step first (hook-mode reducer), as bench.py's warm-up does: the optimizer state must exist before the capture
            st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4)
            st.update()
            torch.cuda.synchronize()
            engine.bump_weight_epoch()
            st.opt.zero_grad()
            g = GraphedStep(st, lambda: st.forward_loss(inp, lab, check="val", epoch=0, n_visual_true=4), s, segments=4, mode=mode)
            assert g.seg.nseg == 3 and len(g.graphs) == (3 if mode == "chain" else 1)          # 4-layer stacks are cut before layers 1 and 3
            assert sum(len(pl) for pl, _ in g.flats) == len(st.opt.state)          # every trained parameter sits in exactly one bucket
            steps0 = st.opt.step_count
            losses = [g.run().item() for _ in range(3)]            # (the capture itself executed nothing)
            torch.cuda.synchronize()
            assert st.opt.step_count == steps0 + 3
            index = {id(p): i for i, p in enumerate(st.params)}
            n_cmp = 0
            for plist, flat in g.flats:
                off = 0
                for p in plist:
                    got = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
                    assert torch.equal(got, ref[index[id(p)]]), f"gradient of parameter #{index[id(p)]} differs"
                    n_cmp += 1
            st.reducer.remove()
            if g.comm is not None:
                g.comm.destroy()
    finally:
        if created:
            dist.destroy_process_group()
    print("graphed ddp:", g.describe(), loss_a, losses, n_cmp, "gradients compared")
    assert all(l == loss_a for l in losses), (loss_a, losses)
    assert n_cmp == sum(r is not None for r in ref)


# ---- BASELINE config 5: videomae-large geometry, 32 frames, fp8 GEMM operands ----------------------------------------------------------
def test_videomae_large_golden_fp32(gpu):
    """VideoEncoder at the VideoMAE-large widths (1024 / 16 heads / 4096) on 32 frames, fp32 policy, closed-form weights, against the vectors
    made from the Hugging Face VideoMAEModel (reference models/tav.py:456,480; generator: oracle/validate_vs_reference.py section 5)."""
    from tav_amd.encoders import VideoEncoder
    runtime.set_precision("fp32")
    vl = dict(layers=2, hidden=1024, heads=16, inter=4096, frames=32, image=32, patch=16, tubelet=2, eps=1e-12)
    enc = cf.fill_module_(VideoEncoder(vl)).cuda()
    batch, _ = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=32, image=32, vocab=1000, pad_id=0, nkeep_fusion=8)
    with torch.no_grad():
        out, S = enc(batch["video_embeds"].cuda(), batch["visual_mask"].cuda())
        emb, n = enc.embed(batch["video_embeds"].cuda(), (~batch["visual_mask"]).cuda())
    assert (S, n) == (56, 8)
    assert rel(out.view(2, S, -1).mean(1), GOLD["L_video_mean"]) < 1e-4
    assert rel(out.view(2, S, -1)[:, 0], GOLD["L_video_tok0"]) < 1e-4
    assert rel(emb.view(2, n, -1), GOLD["L_video_embed_fusion"]) < 1e-4


@pytest.mark.parametrize("policy,tol", [("fp32", 1e-3), ("bf16", 1e-2), ("fp8", 1e-2), ("fp8-all", 4e-2)])
def test_config5_parity_vs_oracle(gpu, policy, tol):
    """BASELINE config 5 (preset B with videomae-large on 32 frames, Linear(1024, 768) bridges): every width true, the video at its full size (32 x 3 x
    224 x 224: 3136 tubelet tokens, 2927 in the video encoder, 209 to the fusion stack), depth 2 per stack.  Logits, loss and global gradient norm
    against the fp32 CPU oracle on identical
    seeded weights: fp32 policy 1e-3, bf16 1e-2, and the fp8 policy -- e4m3 operands with per-tensor scales in the video encoder's linear layers,
    forward, dgrad and wgrad (91 % of the FLOPs of this configuration) -- 1e-2 as well.  "fp8-all" (every stack on e4m3, an experiment, not
    the shipped policy) is held to 4e-2: the un-pooled text token carries e4m3 rounding straight into the logits."""
    cfg = C.preset("B5")
    for k in ("text", "audio", "video", "fusion"):
        cfg[k]["layers"] = 2
    cfg["text"]["vocab"] = 1000
    runtime.set_precision(policy)
    torch.manual_seed(0)
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    assert "vid_2_768.weight" in pre.state_dict() and "vid_2_768_2.weight" in model.state_dict()
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)
    (tx, au, vi), lab = synthetic.make_batch(cfg, 2, s_text=32, t_audio=16000, n_visual_true=209)
    batch = _as_batch(tx, au, vi)
    if "c5" not in _ORACLE_CACHE:                          # the oracle run (a few seconds of CPU) is shared by the four policies
        sdp = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
        sdm = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
        o_logits, o_loss = O.tav_step(sdm, sdp, cfg, batch, lab.long())
        o_loss.backward()
        o_gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in list(sdp.values()) + list(sdm.values()) if v.requires_grad and v.grad is not None)).item()
        _ORACLE_CACHE["c5"] = (sdp, sdm, o_logits.detach(), o_loss.detach(), o_gn)
    sdp, sdm, o_logits, o_loss, o_gn = _ORACLE_CACHE["c5"]
    pre.cuda()
    model.cuda()
    _, _, _, logits, loss = _run_product(pre, model, batch, lab)
    loss.backward()
    gn = grad_norm(list(pre.parameters()) + list(model.parameters())).item()
    e = (rel(logits, o_logits), abs(loss.item() - o_loss.item()) / abs(o_loss.item()), abs(gn - o_gn) / o_gn)
    gmax = max(v.grad.abs().max().item() for v in list(sdp.values()) + list(sdm.values()) if getattr(v, "grad", None) is not None)
    worst, worst_k = 0.0, None
    for mod, sdo, tag in ((pre, sdp, "pre"), (model, sdm, "model")):
        for k, p in mod.named_parameters():
            og = sdo[k].grad
            assert (p.grad is None) == (og is None), f"gradient presence differs for {k}"
            if og is not None:
                ek = (p.grad.detach().cpu() - og).abs().max().item() / (og.abs().max().item() + 1e-3 * gmax)
                if ek > worst:
                    worst, worst_k = ek, f"{tag}.{k}"
    print(f"[config 5, {policy}] logits {e[0]:.2e} loss {e[1]:.2e} grad-norm {e[2]:.2e} worst tensor {worst:.2e} ({worst_k})")
    assert max(e) < tol, e
    assert worst < {"fp32": 1e-3, "bf16": 3e-2, "fp8": 0.15, "fp8-all": 0.3}[policy], (worst, worst_k)


def test_fp8_delayed_scaling_equals_its_calibration_pass(gpu):
    """Round 4, fp8 policy: the first pass over a quantisation site calibrates (amax passes + quantiser, current scaling); every later pass is ONE
    launch that quantises with the scale the previous step left and gathers the new maximum (tav_fp8_quantize_delayed), rolled by the optimizer
    step (tav_fp8_roll_states).  With weights and inputs unchanged the maxima are unchanged, so passes 2 and 3 -- delayed, with a roll in between --
    must reproduce the calibration pass bit for bit (logits and every gradient); and a tensor that GREW past last step's maximum saturates
    instead of overflowing."""
    from tav_amd import ops
    cfg = C.preset("B5")
    for k in ("text", "audio", "video", "fusion"):
        cfg[k]["layers"] = 2
    cfg["text"]["vocab"] = 1000
    cfg["video"]["image"] = 64
    runtime.set_precision("fp8")
    torch.manual_seed(0)
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    synthetic.seeded_init_(pre, 1)
    synthetic.seeded_init_(model, 2)
    nvt = 16
    (tx, au, vi), lab = synthetic.make_batch(cfg, 2, s_text=32, t_audio=16000, n_visual_true=nvt)
    batch = _as_batch(tx, au, vi)
    pre.cuda()
    model.cuda()
    params = list(pre.parameters()) + list(model.parameters())

    def one_pass():
        for p in params:
            p.grad = None
        _, _, _, logits, loss = _run_product(pre, model, batch, lab)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]

    sts = runtime.ctx().cache._fp8_states
    assert sts is None or not sts.calibrated
    l0, g0 = one_pass()                                   # calibrates every site
    sts = runtime.ctx().cache._fp8_states
    assert sts is not None and len(sts.calibrated) == len(sts.index) >= 2 * (4 + 4)          # per video layer: four dY, four dgrad weight operands
    assert float(sts.dev[:len(sts.index), 3].abs().max()) == 0.0                              # (calibration gathers nothing)
    l1, g1 = one_pass()                                   # delayed, scales of the calibration pass
    assert float(sts.dev[:len(sts.index), 3].max()) > 0.0                                     # the delayed kernels gathered their maxima
    ops.fp8_roll_all()
    torch.cuda.synchronize()
    assert float(sts.dev[:len(sts.index), 3].abs().max()) == 0.0
    l2, g2 = one_pass()                                   # delayed, scales rolled from pass 2's maxima (= the same maxima)
    for lg, gg in ((l1, g1), (l2, g2)):
        assert torch.equal(lg, l0)
        assert all((a is None and b is None) or torch.equal(a, b) for a, b in zip(gg, g0))
    # saturation: quantise a tensor 8x larger than the state's maximum with the stale scale
    x = torch.randn(256, 512, device="cuda").bfloat16()
    key = (sts, ("probe", "x"))
    a = ops.fp8_quantize(x, state=key)                    # calibrates
    b = ops.fp8_quantize((x.float() * 8).bfloat16(), state=key)      # delayed, scale too large by 8x
    torch.cuda.synchronize()
    qa, qb = a.q.float(), b.q.float()
    assert torch.isfinite(qb).all() and float(qb.abs().max()) == 448.0 and float(qa.abs().max()) == 448.0
    runtime.set_precision("bf16")


def test_c_abi_allreduce_bucket_single_rank(gpu):
    """the C ABI's own gradient-bucket collective (RCCL, tav_allreduce_bucket) with one rank: communicator from a unique id,
    in-place mean enqueued on the caller's stream (f32 and bf16), communicator destroyed."""
    import ctypes
    from tav_amd import _lib
    from tav_amd._lib import ptr, stream
    h = _lib.lib()
    ver = ctypes.c_int32()
    assert h.tav_comm_rccl_version(ctypes.byref(ver)) == 0 and ver.value >= 20000      # RCCL is resolved lazily (dlopen): which one did we get?
    print(f"[c-abi collective] RCCL runtime version {ver.value} (header at build time: /opt/rocm/include/rccl/rccl.h)")
    uid = ctypes.create_string_buffer(128)
    assert h.tav_comm_unique_id(uid) == 0
    comm = ctypes.c_void_p()
    assert h.tav_comm_init_rank(ctypes.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        for dtype, code in ((torch.float32, 0), (torch.bfloat16, 1)):
            x = torch.randn(1 << 20, device="cuda").to(dtype)
            ref = x.clone()
            assert h.tav_allreduce_bucket(ptr(x), x.numel() * x.element_size(), code, comm, stream()) == 0
            torch.cuda.synchronize()
            assert torch.equal(x, ref)                      # the mean over one rank
        # ... and CAPTURED into a hipGraph on a side stream forked from the capture's origin (what ddp.GraphedStep's single-graph mode does):
        # a raw RCCL call on the caller's stream, no process-group watchdog thread around it
        x = torch.randn(1 << 20, device="cuda")
        y = torch.empty_like(x)
        origin, side = torch.cuda.Stream(), torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(origin):
            with runtime.capture(g, origin, branches=[side], capture_error_mode="thread_local"):
                y.copy_(x).mul_(2.0)
                runtime.stream_wait(side, origin)
                assert h.tav_allreduce_bucket(ptr(y), y.numel() * 4, 0, comm, side.cuda_stream) == 0
                runtime.stream_wait(origin, side)
                y.add_(1.0)
        for k in range(3):
            x.fill_(float(k))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(y, torch.full_like(y, 2.0 * k + 1.0))
        assert h.tav_allreduce_bucket(None, 4, 0, comm, stream()) == -1 and h.tav_allreduce_bucket(ptr(x), 3, 0, comm, stream()) == -2
    finally:
        assert h.tav_comm_destroy(comm) == 0


# ---- optimizer sharded over data-parallel ranks: the HIP kernels, two ranks on ONE GPU over gloo ----------------------
def _worker_sharded_step(rank, world, port, use_graphs, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tav_amd import engine
        from tav_amd.ddp import GraphedStep
        from tav_amd.optim import ShardedAdamW
        from tav_amd.train_model.tav_train import TrainStep
        from tav_amd.utils.global_functions import CrossEntropyLoss
        torch.cuda.set_device(0)
        cfg = C.preset("B-tiny")                                  # every width of preset B, 2 layers per stack, 1000-word vocabulary: 70 M parameters
        runtime.set_precision("bf16")                             # (the exchanges cross the host over gloo here: the buckets' size is the test's time)

        def build():
            torch.manual_seed(0)
            pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
            synthetic.seeded_init_(pre, 1)
            synthetic.seeded_init_(model, 2)
            pre.cuda()
            model.cuda()
            return TrainStep(model, pre, CrossEntropyLoss(), lr=1e-3, weight_decay=1e-2, clip=1.0)

        s = torch.cuda.Stream()
        res = {}
        with torch.cuda.stream(s):
            inp, lab = synthetic.make_batch(cfg, 2, seed=100 + rank, s_text=16, t_audio=8000, n_visual_true=4, device="cuda")   # this rank's utterances
            for name, sharded in (("replicated", False), ("sharded", True)):
                st = build()
                assert st.reducer is not None
                st.forward_backward(inp, lab, check="val", epoch=0, n_visual_true=4)       # bench.py's eager warm-up step: hook-mode all-reduce,
                st.update()                                                                # replicated AdamW -> the moments the shards inherit
                torch.cuda.synchronize()
                st.opt.zero_grad()
                g = GraphedStep(st, lambda: st.forward_loss(inp, lab, check="val", epoch=0, n_visual_true=4), s, segments=4, mode="chain",
                                use_graphs=use_graphs, shard_optimizer=sharded)
                losses, norms = [], []
                for _ in range(2):
                    losses.append(g.run().item())
                    norms.append(st.opt.last_norm.item())
                torch.cuda.synchronize()
                norm = norms
                extra = {}
                if sharded:
                    assert isinstance(st.opt, ShardedAdamW) and len(st.opt.state) == 0
                    own, tot = st.opt.owned_elements()
                    # the moments this rank holds against the same elements of the replicated optimizer's (the gather into a complete state_dict is
                    # plain torch copies + the same exchange as the parameters: tests/test_ddp_gloo.py)
                    index = {id(p): i for i, p in enumerate(st.params)}
                    ref_mv = res["replicated"]["moments"]
                    sd_same = all(torch.equal(st.opt._m[a:a + n], ref_mv[index[id(p)]][0].view(-1)[off:off + n]) and
                                  torch.equal(st.opt._v[a:a + n], ref_mv[index[id(p)]][1].view(-1)[off:off + n]) for (p, off, n, _, _, a) in st.opt._mine)
                    extra = dict(own=own, tot=tot, desc=g.describe(), state_elems=st.opt._m.numel(), sd_same=sd_same, pieces=len(st.opt._mine))
                else:
                    extra = dict(moments={i: tuple(t.clone() for t in st.opt.state[p]) for i, p in enumerate(st.params) if p in st.opt.state})
                res[name] = dict(params=[p.detach().clone() for p in st.params], losses=losses, norm=norm, steps=st.opt.step_count, **extra)
                st.reducer.remove()
                engine.bump_weight_epoch()
        a, b = res["replicated"], res["sharded"]
        same = all(torch.equal(x, y) for x, y in zip(a["params"], b["params"]))
        worst = max((x - y).abs().max().item() for x, y in zip(a["params"], b["params"]))
        sd_same = b["sd_same"] and b["pieces"] > 0
        moved = max((x - y).abs().max().item() for x, y in zip(a["params"], build().params))
        got = [None] * world
        dist.all_gather_object(got, dict(same=same, worst=worst, sd_same=sd_same, moved=moved, losses=(a["losses"], b["losses"]), norms=(a["norm"], b["norm"]),
                                         steps=(a["steps"], b["steps"]), own=b["own"], tot=b["tot"], state_elems=b["state_elems"], desc=b["desc"]))
        if rank == 0:
            out.put(got)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("use_graphs", [True])      # (False -- the eager chain -- passes too, 4 min; its host logic is what tests/test_ddp_gloo.py runs on the CPU)
def test_sharded_optimizer_two_ranks_one_gpu(gpu, use_graphs):
    """optim.ShardedAdamW with the HIP kernels behind ddp.GraphedStep(shard_optimizer=True): two ranks (two processes on this GPU, gloo), each with its own
    utterances, one eager warm-up step with the replicated optimizer, then two steps of the chain.  Against the same chain with the all-reduce and the
    replicated FusedAdamW: parameters, losses, the clipped gradient norm and the moments each rank holds are bit-equal on both ranks; each rank holds
    about half of the moments.  use_graphs=True is bench.py's form (segment graphs + three optimizer graphs, the exchanges between them)."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded_step, args=(r, 2, port, use_graphs, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(420)
            assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
        got = q.get(timeout=10)
    finally:
        for p in procs:                       # (a rank that hangs must not outlive the test: these exact children, nothing else)
            if p.is_alive():
                p.kill()
                p.join(10)
    print("sharded optimizer:", got[0]["desc"], "| worst", [g["worst"] for g in got], "| norms", got[0]["norms"], "| moved", got[0]["moved"])
    for g in got:
        assert g["same"] and g["sd_same"], g["worst"]
        assert g["losses"][0] == g["losses"][1] and g["norms"][0] == g["norms"][1] and g["steps"][0] == g["steps"][1]
        assert g["moved"] > 1e-4                                  # the steps did change the parameters
        assert 0.4 < g["own"] / g["tot"] < 0.6 and g["own"] <= g["state_elems"] < g["own"] + 64 * 600      # (each piece's moments padded to 256 bytes)
    assert got[0]["own"] + got[1]["own"] == got[0]["tot"]
    assert got[0]["losses"] != got[1]["losses"]                   # two different batches went in
