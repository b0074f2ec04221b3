"""-m gpu: oracle parity AT THE BENCHMARKED SIZES (VERDICT r02 "next round" item 1).

The headline (global batch 32 on one GPU) and BASELINE configs[1] (batch 8) run tile plans that no batch-2 comparison reaches
(256x256 NT tiles, the mixed 256/128 schedule, the 256-wide grouped weight-gradient tile with f32 slabs), and config 5 runs 24 video
layers.  Here:
  * batch 8, preset B, bf16, full depth and sizes: DIRECT comparison with the CPU oracle (reference models/tav.py:344-417,473-504,
    train_model/tav_train.py:56-65 restated in oracle/tav_oracle.py) -- logits, loss, clip_grad_norm_ value, every gradient;
  * batch 32: the product's logits / per-row losses equal the concatenation of sixteen batch-2 runs (the size the oracle checks in
    test_model_gpu.py::test_full_depth_full_size_parity), grad(b=32) equals the mean of the sixteen split gradients, and rows 0..7 of the
    batch-32 logits equal the oracle's batch-8 logits (the batch-8 oracle batch IS rows 0..7 of the batch-32 one);
  * three seeds (weights and inputs) for the bf16 full-depth gate, worst case reported;
  * config 5 (videomae-large, 32 frames) at its real depth -- 24 video layers, 12/12/12 elsewhere, full text / audio / video sizes --
    fp32 / bf16 / fp8 against the oracle.
Tolerances: BASELINE.json north_star (1e-3 relative fp32, 1e-2 bf16 / fp8) on logits, loss, grad-norm.
"""
import pytest
import torch

import tav_amd  # noqa: F401
from oracle import tav_oracle as O  # noqa: F401  (the oracle is only the checker)
from tav_amd import config as C
from tav_amd import runtime, synthetic
from tav_amd.models.tav import PreFormer, TAVForMAE
from tav_amd.optim import grad_norm

from test_model_gpu import ARGS, _as_batch, _compare_with_oracle, _oracle_full, _run_product, rel

pytestmark = pytest.mark.gpu


def _batch32(cfg, seed=0):
    (tx, au, vi), lab = synthetic.make_batch(cfg, 32, seed=1234 + seed)
    return _as_batch(tx, au, vi), lab


def _first_rows(batch, lab, n):
    return {k: v[:n].clone() for k, v in batch.items()}, lab[:n].clone()


def _oracle_b8_of_b32():
    """Oracle record for rows 0..7 of the batch-32 synthetic batch (same weights as `_oracle_full("B")`: seeds 1 / 2)."""
    import test_model_gpu as T
    key = ("B-rows0-7-of-32", 0, 8)
    if key not in T._ORACLE_CACHE:
        cfg = C.preset("B")
        torch.manual_seed(0)
        pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
        synthetic.seeded_init_(pre, 1)
        synthetic.seeded_init_(model, 2)
        batch, lab = _first_rows(*_batch32(cfg), 8)
        sdp = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
        sdm = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
        o_logits, o_loss = O.tav_step(sdm, sdp, cfg, batch, lab.long())
        o_loss.backward()
        grads = {("pre", k): v.grad for k, v in sdp.items() if v.requires_grad and v.grad is not None}
        grads.update({("model", k): v.grad for k, v in sdm.items() if v.requires_grad and v.grad is not None})
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
        state = ({k: v.detach() for k, v in sdp.items()}, {k: v.detach() for k, v in sdm.items()})
        T._ORACLE_CACHE[key] = (cfg, batch, lab, state, o_logits.detach(), o_loss.item(), gn, grads)
    return T._ORACLE_CACHE[key]


def test_batch8_direct_oracle_bf16(gpu):
    """BASELINE configs[1]: preset B, bf16, batch 8, every stack at full depth, full input sizes -- against the oracle itself (~25 s of CPU)."""
    _compare_with_oracle(_oracle_b8_of_b32(), "bf16", 1e-2, "full-depth B, batch 8 (rows 0..7 of the batch-32 workload)")


def test_batch32_equals_sixteen_oracle_checked_b2_runs_bf16(gpu):
    """The headline workload (global batch 32 on one GPU, preset B, bf16, full depth): logits and per-row losses equal the concatenation of
    sixteen batch-2 runs (<= 2e-3), every parameter gradient equals the mean of the sixteen split gradients, and rows 0..7 equal the
    oracle's batch-8 logits within the bf16 budget.  Ties the big-tile plans (256x256 NT, mixed schedule, 256-wide grouped weight gradients
    with f32 slabs -- gemm.hip) to the oracle without a 32-utterance CPU run."""
    cfg, _, _, (sdp, sdm), o_logits8, _, _, _ = _oracle_b8_of_b32()
    runtime.set_precision("bf16")
    pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
    pre.load_state_dict(sdp)
    model.load_state_dict(sdm)
    pre.cuda()
    model.cuda()
    batch, lab = _batch32(cfg)
    batch = {k: v.cuda() for k, v in batch.items()}
    lab = lab.cuda()
    named = [(f"pre.{k}", p) for k, p in pre.named_parameters()] + [(f"model.{k}", p) for k, p in model.named_parameters()]

    def run(sel):
        for _, p in named:
            p.grad = None
        b = {k: v[sel] for k, v in batch.items()}
        _, _, _, logits, loss = _run_product(pre, model, b, lab[sel])
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), {k: p.grad.detach().clone() for k, p in named if p.grad is not None}

    logits32, g32 = run(slice(0, 32))
    gn32 = grad_norm([p for _, p in named]).item()
    parts, gsum = [], None
    for j in range(16):
        lg, g = run(slice(2 * j, 2 * j + 2))
        parts.append(lg)
        assert g.keys() == g32.keys()
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    logits2 = torch.cat(parts)
    e_logits = rel(logits32, logits2)
    ce = torch.nn.functional.cross_entropy
    rows32 = ce(logits32.float(), lab.long(), reduction="none")
    rows2 = ce(logits2.float(), lab.long(), reduction="none")
    e_rows = ((rows32 - rows2).abs().max() / rows2.abs().max()).item()
    gmax = max(v.abs().max().item() for v in g32.values())
    worst, worst_k = 0.0, None
    sq = 0.0
    for k in g32:
        m = gsum[k] / 16.0
        e = (g32[k] - m).abs().max().item() / (m.abs().max().item() + 1e-3 * gmax)
        sq += float((m.double() ** 2).sum())
        if e > worst:
            worst, worst_k = e, k
    e_gn = abs(gn32 - sq ** 0.5) / sq ** 0.5
    e_oracle = rel(logits32[:8], o_logits8)
    print(f"[batch 32 vs sixteen batch-2 runs, bf16] logits {e_logits:.2e} per-row loss {e_rows:.2e} grad-norm {e_gn:.2e} "
          f"worst gradient tensor {worst:.2e} ({worst_k}); rows 0..7 vs the oracle's batch-8 logits {e_oracle:.2e}")
    assert e_logits < 2e-3 and e_rows < 2e-3, (e_logits, e_rows)
    assert e_gn < 2e-3 and worst < 1e-2, (e_gn, worst, worst_k)
    assert e_oracle < 1e-2, e_oracle


@pytest.mark.parametrize("seed", [1, 2])
def test_full_depth_bf16_more_seeds(gpu, seed):
    """The bf16 full-depth gate on two more seeds (weights and inputs; seed 0 is test_full_depth_full_size_parity): the round-2 result sat at
    8.2e-3 of a 1e-2 budget on one seed.  (Round 4: the suite's wall time was the CPU oracle running on a thread pool sized for the HOST's
    core count instead of the cgroup's 16 -- tests/conftest.py caps it; 823 s -> 291 s for the whole -m gpu suite.)"""
    _compare_with_oracle(_oracle_full("B", seed=seed), "bf16", 1e-2, f"full-depth B seed {seed}")


def _c5_oracle(seed=0):
    cfg = C.preset("B5")                                       # 12 text / 12 audio / 24 video (1024-16-4096, 32 frames) / 12 fusion layers
    return _oracle_full("B5", seed=seed, batch_size=2, cfg=cfg, tag="B5-full")


@pytest.mark.parametrize("policy,tol,gtol", [("fp32", 1e-3, 1e-3), ("bf16", 1e-2, 5e-2), ("fp8", 1e-2, 0.2)])
def test_config5_full_depth_parity(gpu, policy, tol, gtol):
    """BASELINE configs[4] at its REAL depth: videomae-large, 24 layers, 32 x 3 x 224 x 224 frames (3136 tubelet tokens, 2927 in the video
    encoder, 209 to the fusion stack), text 128, audio 5 s, every other stack at full depth, batch 2; fp32 / bf16 / fp8 policies against the
    fp32 CPU oracle (reference models/tav.py:456,480 scaled up; ~35 s of CPU for the oracle, shared by the three policies)."""
    _compare_with_oracle(_c5_oracle(), policy, tol, "config 5 full depth", grad_tol=gtol)


def test_per_rank_batch4_full_depth_bf16(gpu):
    """BASELINE configs[2] as ONE RANK sees it: global batch 32 over 8 GPUs = 4 utterances per rank, preset B, bf16, every stack at full depth,
    full input sizes -- directly against the CPU oracle (logits, loss, clip_grad_norm_ value, every gradient; ~14 s of CPU).  The tile plans at
    M = 4 x {128, 249, 481, 1464} tokens differ from the batch-2 and batch-8 cases (small-grid 4-deep rings, 128-wide weight-gradient tiles)."""
    _compare_with_oracle(_oracle_full("B", seed=0, batch_size=4), "bf16", 1e-2, "full-depth B, batch 4 (one rank of configs[2])")


def test_config0_text_classifier_b16_s128_full_depth_bf16(gpu):
    """BASELINE configs[0] at its stated size: SingleModels/text_nn.py's classifier (reference SingleModels/models/text.py:41-69: BERT-base pooled
    output -> Linear(768, 7)), 12 layers, batch 16, sequence 128 (last quarter of every row padded), bf16 policy against the fp32 CPU oracle:
    logits, loss, clip_grad_norm_ value within 1e-2, every parameter gradient compared."""
    from tav_amd.SingleModels.models.text import BertClassifier
    from tav_amd.utils.global_functions import CrossEntropyLoss
    cfg = C.preset("B")
    assert cfg["text"]["layers"] == 12 and cfg["text"]["hidden"] == 768
    runtime.set_precision("bf16")
    model = BertClassifier(dict(output_dim=7, dropout=0.5), config=cfg)
    synthetic.seeded_init_(model, 3)
    (tx, _, _), lab = synthetic.make_batch(cfg, 16, s_text=128, text_only=True)
    assert tx["input_ids"].shape == (16, 128) and float(tx["attention_mask"].sum()) < 16 * 128
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    o_logits = O.text_classifier_forward(sd, cfg, tx["input_ids"], tx["attention_mask"])
    o_loss = torch.nn.functional.cross_entropy(o_logits, lab.long())
    o_loss.backward()
    o_grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    o_gn = torch.sqrt(sum((g.double() ** 2).sum() for g in o_grads.values())).item()
    model.cuda()
    logits = model(tx["input_ids"], tx["attention_mask"], "val")
    loss = CrossEntropyLoss()(logits, lab.cuda())
    loss.backward()
    torch.cuda.synchronize()
    gn = grad_norm(list(model.parameters())).item()
    e_logits, e_loss, e_gn = rel(logits.detach().cpu(), o_logits.detach()), abs(loss.item() - o_loss.item()) / abs(o_loss.item()), abs(gn - o_gn) / o_gn
    gmax = max(g.abs().max().item() for g in o_grads.values())
    worst, worst_k = 0.0, None
    for k, p in model.named_parameters():
        og = o_grads.get(k)
        assert (p.grad is None) == (og is None), k
        if og is not None:
            e = (p.grad.detach().cpu() - og).abs().max().item() / (og.abs().max().item() + 1e-3 * gmax)
            if e > worst:
                worst, worst_k = e, k
    print(f"[configs[0] text classifier b=16 S=128 12 L bf16] logits {e_logits:.2e} loss {e_loss:.2e} grad-norm {e_gn:.2e} worst tensor {worst:.2e} ({worst_k})")
    assert e_logits < 1e-2 and e_loss < 1e-2 and e_gn < 1e-2, (e_logits, e_loss, e_gn)
    assert worst < 5e-2, (worst, worst_k)


@pytest.mark.parametrize("policy,tol,gtol", [("bf16", 1e-2, 5e-2), ("fp8", 1e-2, 0.2)])
def test_config5_full_depth_parity_second_seed(gpu, policy, tol, gtol):
    """Config 5 at its real depth on ANOTHER seed (weights and inputs).  Round 4 found the round-3 fp8 policy (QKV forward on e4m3) inside the 1e-2 budget
    on seed 0 only -- 1.13e-2 / 1.36e-2 on seeds 1 / 2 (profiles/r04_fp8_seeds.txt) -- and took every forward GEMM off e4m3; one seed is not a gate."""
    _compare_with_oracle(_c5_oracle(seed=1), policy, tol, "config 5 full depth, seed 1", grad_tol=gtol)
