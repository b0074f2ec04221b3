"""CPU: the oracle restatement reproduces the golden vectors that oracle/validate_vs_reference.py generated from the
reference's own utils/TAVFormer.py classes + the HF modules it calls (closed-form weights/inputs, tests/closed_form.py).
Also pins that the product modules expose exactly the reference-side state_dict keys (the closed-form fill is by key)."""
import os

import numpy as np
import pytest
import torch

import closed_form as cf
import tav_amd  # noqa: F401
from oracle import tav_oracle as O
from tav_amd import config as C
from tav_amd.models.tav import PreFormer, TAVForMAE
from tav_amd.utils.TAVFormer import VideoMAEEncoder

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "tav_golden.npz"))


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _masks(S):
    m = torch.zeros(2, 1, 1, S)
    m[..., : S // 4] = O.FP16_MIN
    m[..., S // 4: S // 2] = 65505.0
    m[0, ..., S // 2 - 1] = 1.0
    return {"none": None, "zeros": torch.zeros(2, 1, 1, S), "refstyle": m}


@pytest.mark.parametrize("S", [8, 37])
@pytest.mark.parametrize("mname", ["none", "zeros", "refstyle"])
def test_fusion_encoder_matches_reference_golden(S, mname):
    enc = cf.fill_module_(VideoMAEEncoder(dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12), 2))
    sd = {"f." + k: v.clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    x = (cf.tensor_for(f"fusion_x{S}", (2, S, 768), kind="bias") * 20).requires_grad_(True)
    y = O.fusion_encoder(sd, "f", x, _masks(S)[mname], dict(layers=2, heads=12, eps=1e-12))
    y.square().mean().backward()
    assert rel(y.detach(), GOLD[f"fusion_S{S}_{mname}_y"]) < 1e-5
    assert rel(x.grad, GOLD[f"fusion_S{S}_{mname}_dx"]) < 1e-4
    assert rel(sd["f.layer.0.intermediate.dense.weight"].grad[:8, :8], GOLD[f"fusion_S{S}_{mname}_dW1_l0"]) < 1e-4


@pytest.mark.parametrize("early", [0, 1])
def test_transformer_encoder_golden(early):
    # key names of reference utils/TAVFormer.py TransformerEncoder(768, num_layers=1)
    shapes = {}
    for n in ("query_matrix", "key_matrix", "value_matrix"):
        shapes[f"t.layers.0.attention.{n}.weight"] = (768, 768)
    shapes.update({"t.layers.0.attention.out.weight": (768, 768), "t.layers.0.attention.out.bias": (768,), "t.layers.0.norm1.weight": (768,),
                   "t.layers.0.norm1.bias": (768,), "t.layers.0.feed_forward.1.weight": (3072, 768), "t.layers.0.feed_forward.1.bias": (3072,),
                   "t.layers.0.feed_forward.3.weight": (768, 3072), "t.layers.0.feed_forward.3.bias": (768,), "t.layers.0.norm2.weight": (768,),
                   "t.layers.0.norm2.bias": (768,)})
    sd = {k: cf.tensor_for(k[2:], s) for k, s in shapes.items()}     # generator filled the bare module (no "t." prefix)
    x = cf.tensor_for("te_x", (2, 11, 768), kind="bias") * 20
    m = torch.zeros(2, 1, 1, 11)
    m[..., 8:] = O.FP16_MIN
    y = O.transformer_encoder(sd, "t", x, m, 1, 12, bool(early))
    assert rel(y, GOLD[f"transformer_encoder_early{early}_y"]) < 1e-5


@pytest.fixture(scope="module", params=["A", "B"])
def composed(request):
    name = request.param
    cfg = C.preset(name + "-tiny")
    pre = cf.fill_module_(PreFormer(cfg))
    model = cf.fill_module_(TAVForMAE(dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12), cfg))
    batch, labels = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=16, image=32, vocab=cfg["text"]["vocab"], pad_id=cfg["text"]["pad_id"], nkeep_fusion=4)
    return name, cfg, pre, model, batch, labels


def test_presets_agree_with_oracle(composed):
    name, cfg, *_ = composed
    ocfg = O.preset(name + "-tiny")
    ocfg["video"]["image"] = 32
    assert ocfg == cfg
    assert O.preset(name) == C.preset(name)


def test_encoders_golden(composed):
    name, cfg, pre, model, batch, _ = composed
    sd = model.state_dict()
    with torch.no_grad():
        _, pooled = O.text_encoder(sd, "bert", cfg["text"], batch["input_ids"], batch["text_mask"])
        aud = O.w2v2_model(sd, "wav2vec2", cfg["audio"], batch["audio_features"])
        vid = O.videomae_model(sd, "videomae", cfg["video"], batch["video_embeds"], batch["visual_mask"])
    assert rel(pooled, GOLD[f"{name}_text_pooled"]) < 1e-4
    assert rel(aud, GOLD[f"{name}_audio_last"]) < 1e-4
    assert rel(vid.mean(1), GOLD[f"{name}_video_mean"]) < 1e-4


def test_preformer_and_step_golden(composed):
    name, cfg, pre, model, batch, labels = composed
    sdp = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
    sdm = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    tav, emb, mask = O.preformer_forward(sdp, cfg, batch["input_ids"], batch["audio_features"], batch["video_embeds"], batch["text_mask"], batch["audio_mask"],
                                         batch["visual_mask"])
    assert rel(tav.detach(), GOLD[f"{name}_pre_tav"]) < 1e-5
    assert (emb.numpy() == GOLD[f"{name}_pre_tav_embed"]).all()
    assert rel(mask, GOLD[f"{name}_pre_attention_mask"]) == 0.0
    # mask value sets the reference really produces (models/tav.py:383-397)
    assert set(np.unique(mask.numpy()).tolist()) <= {0.0, -65504.0, 65505.0, 1.0}
    logits, loss = O.tav_step(sdm, sdp, cfg, batch, labels)
    loss.backward()
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in list(sdp.values()) + list(sdm.values()) if v.requires_grad and v.grad is not None)).item()
    assert rel(logits.detach(), GOLD[f"{name}_logits"]) < 1e-4
    assert abs(loss.item() - GOLD[f"{name}_loss"][0]) < 1e-5
    assert abs(gn - GOLD[f"{name}_gradnorm"][0]) / GOLD[f"{name}_gradnorm"][0] < 1e-4
    assert rel(sdm["linear1.weight"].grad[:, :16], GOLD[f"{name}_grad_linear1"]) < 1e-3
    assert rel(sdm["random_mae_encoder.layer.0.attention.attention.query.weight"].grad[:8, :8], GOLD[f"{name}_grad_fusion_q0"]) < 2e-2


# ---- text-only classifier, text+audio dual classifier ---------------------------------------------
ARGS_F = dict(output_dim=7, dropout=0.5)


@pytest.mark.parametrize("name", ["A", "B"])
@pytest.mark.parametrize("tag", ["textcls", "textaudio"])
def test_single_and_dual_models_golden(name, tag):
    from tav_amd.DoubleModels.models.text_audio import BertAudioClassifier
    from tav_amd.SingleModels.models.text import BertClassifier
    cfg = C.preset(name + "-tiny")
    batch, labels = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=16, image=32, vocab=cfg["text"]["vocab"], pad_id=cfg["text"]["pad_id"], nkeep_fusion=4)
    model = cf.fill_module_(BertClassifier(ARGS_F, config=cfg) if tag == "textcls" else BertAudioClassifier(ARGS_F, config=cfg))
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    if tag == "textcls":
        assert {"linear.weight", "linear.bias"} <= set(sd) and all(k.startswith(("bert.", "linear.")) for k in sd)     # reference text.py:48-52
        logits = O.text_classifier_forward(sd, cfg, batch["input_ids"], batch["text_mask"])
    else:
        assert {"bert_norm.weight", "aud_norm.weight", "wav_2_768_2.weight", "linear1.weight"} <= set(sd)            # names of models/tav.py:435-458
        assert sd["linear1.weight"].shape == (7, 1536)
        logits = O.text_audio_forward(sd, cfg, batch["input_ids"], batch["text_mask"], batch["audio_features"])
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.requires_grad and v.grad is not None)).item()
    assert rel(logits.detach(), GOLD[f"{name}_{tag}_logits"]) < 1e-4
    assert abs(loss.item() - GOLD[f"{name}_{tag}_loss"][0]) < 1e-5
    assert abs(gn - GOLD[f"{name}_{tag}_gradnorm"][0]) / GOLD[f"{name}_{tag}_gradnorm"][0] < 1e-4


def test_videomae_large_geometry_golden():
    """BASELINE config 5 geometry (VideoMAE-large widths 1024 / 16 heads / 4096, 32 frames; tiny depth and image): the oracle against the vectors
    made from the Hugging Face VideoMAEModel the reference calls at models/tav.py:456,480 (oracle/validate_vs_reference.py section 5)."""
    vl = dict(layers=2, hidden=1024, heads=16, inter=4096, frames=32, image=32, patch=16, tubelet=2, eps=1e-12)
    shapes = {}
    H, Fd = vl["hidden"], vl["inter"]
    shapes["embeddings.patch_embeddings.projection.weight"] = (H, 3, 2, 16, 16)
    shapes["embeddings.patch_embeddings.projection.bias"] = (H,)
    for i in range(vl["layers"]):
        lp = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            shapes[lp + f"attention.attention.{n}.weight"] = (H, H)
            shapes[lp + f"attention.attention.{n}.bias"] = (H,)
        shapes[lp + "attention.output.dense.weight"], shapes[lp + "attention.output.dense.bias"] = (H, H), (H,)
        shapes[lp + "intermediate.dense.weight"], shapes[lp + "intermediate.dense.bias"] = (Fd, H), (Fd,)
        shapes[lp + "output.dense.weight"], shapes[lp + "output.dense.bias"] = (H, Fd), (H,)
        for n in ("layernorm_before", "layernorm_after"):
            shapes[lp + n + ".weight"], shapes[lp + n + ".bias"] = (H,), (H,)
    sd = {"v." + k: v for k, v in cf.state_dict_for(shapes).items()}
    batch, _ = cf.batch_for(B=2, S_text=12, T_audio=3200, frames=32, image=32, vocab=1000, pad_id=0, nkeep_fusion=8)
    with torch.no_grad():
        out = O.videomae_model(sd, "v", vl, batch["video_embeds"], batch["visual_mask"])
        emb = O.videomae_embeddings(sd, "v.embeddings", vl, batch["video_embeds"], ~batch["visual_mask"])
    assert out.shape == (2, 56, 1024) and emb.shape == (2, 8, 1024)
    assert rel(out.mean(1), GOLD["L_video_mean"]) < 1e-5
    assert rel(out[:, 0], GOLD["L_video_tok0"]) < 1e-5
    assert rel(emb, GOLD["L_video_embed_fusion"]) < 1e-5
