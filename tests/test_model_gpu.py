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
    (y.square().mean()).backward() if False else None
    assert rel(y, GOLD[f"fusion_S{S}_{mname}_y"]) < 1e-4


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
            for k in g0:
                assert rel(g0[k], g1[k]) < 2e-2, k


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
