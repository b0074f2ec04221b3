"""ORACLE — test infrastructure, not product code.

CPU (fp32, plain PyTorch tensor ops) restatement of the reference's TAV hot path, written from the reference text:
  * models/tav.py:344-417   PreFormer.forward          -> preformer_forward()
  * models/tav.py:473-504   TAVForMAE.forward          -> tavformae_forward()
  * utils/TAVFormer.py:171-439  VideoMAEEncoder (fusion stack, mask added AFTER softmax) -> fusion_encoder()
  * utils/TAVFormer.py:10-166   TransformerEncoder (post-LN alternative fusion)          -> transformer_encoder()
  * utils/global_functions.py:51-83  NewCrossEntropyLoss                                 -> new_cross_entropy()
and of the third-party arithmetic those call into (transformers, not vendored by the reference; pinned
`transformers==4.18.0` in README_and_Requirements/requirements.txt:108 which predates VideoMAE, so the effective
version is unknown; restated against the installed 5.15.0 sources):
  * HF roberta/modeling_roberta.py:75-155,211-250,336-398,530-536 (+ bert position ids) -> text_*()
  * HF wav2vec2/modeling_wav2vec2.py:256-434,466-802,1319-1375                          -> w2v2_*()
  * HF videomae/modeling_videomae.py:80-177,209-376,420-466                             -> videomae_*()

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinning: the reference has no tests/goldens for this path ("parity unpinned" by the reference itself).
oracle/validate_vs_reference.py (build container only) checks every function here against the reference's own
utils/TAVFormer.py classes and the HF model classes instantiated from local configs, and writes tests/golden/*.npz.

All functions are pure: (state_dict-like mapping of tensors keyed with the reference/HF parameter names, config dict,
inputs) -> outputs.  Gradients come from torch.autograd over these same ops.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FP16_MIN = float(torch.finfo(torch.float16).min)       # -65504.0, models/tav.py:383,390
F32_MIN = float(torch.finfo(torch.float32).min)


# ------------------------------------------------------------------------------------------------ presets
def preset(name="B", **over):
    """Model geometry. 'A' = what the reference's from_pretrained names resolve to (SURVEY.md §8 presets),
    'B' = BASELINE.json trio (bert-base + wav2vec2-base + videomae-base). 'tiny' variants keep true widths
    but cut depth for tests."""
    conv = dict(conv_dim=[512] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2])
    video = dict(layers=12, hidden=768, heads=12, inter=3072, frames=16, image=224, patch=16, tubelet=2, eps=1e-12)
    fusion = dict(layers=12, hidden=768, heads=12, inter=3072, eps=1e-12)
    if name.startswith("A"):
        cfg = dict(
            text=dict(kind="roberta", layers=6, hidden=768, heads=12, inter=3072, vocab=50265, max_pos=514, type_vocab=1, pad_id=1, eps=1e-5),
            audio=dict(layers=24, hidden=1024, heads=16, inter=4096, feat_norm="layer", stable_ln=True, conv_bias=True, pos_k=128,
                       pos_groups=16, eps=1e-5, **conv),
            video=video, fusion=fusion, output_dim=7)
    else:
        cfg = dict(
            text=dict(kind="bert", layers=12, hidden=768, heads=12, inter=3072, vocab=30522, max_pos=512, type_vocab=2, pad_id=0, eps=1e-12),
            audio=dict(layers=12, hidden=768, heads=12, inter=3072, feat_norm="group", stable_ln=False, conv_bias=False, pos_k=128,
                       pos_groups=16, eps=1e-5, **conv),
            video=video, fusion=fusion, output_dim=7)
    if name.endswith("tiny"):
        cfg["text"]["layers"] = 2
        cfg["audio"]["layers"] = 2
        cfg["video"] = dict(video, layers=2)
        cfg["fusion"] = dict(fusion, layers=2)
        cfg["text"]["vocab"] = 1000
    for k, v in over.items():
        cfg[k] = v
    return cfg


# ------------------------------------------------------------------------------------------------ primitives
def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x, bias=True):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias and (p + ".bias") in sd else None)


def _split_heads(x, nh):
    B, S, H = x.shape
    return x.view(B, S, nh, H // nh).permute(0, 2, 1, 3)


def _merge_heads(x):
    B, nh, S, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, S, nh * d)


# ------------------------------------------------------------------------------------------------ fusion encoder
def fusion_layer(sd, p, x, attention_mask, nh, eps, head_mask=None, probs_out=None):
    """One VideoMAELayer of the reference's own copy (utils/TAVFormer.py:243-271 -> :343-391).
    head_mask: the layer's entry of VideoMAEEncoder.forward(head_mask=) (:190), multiplied into the probabilities BEFORE the mask is
    added (:368-370); probs_out: list that receives the returned attention_probs (:389, output_attentions=True)."""
    h = _ln(sd, p + ".layernorm_before", x, eps)                                     # :254
    a = p + ".attention.attention"
    q_bias, v_bias = sd.get(a + ".q_bias"), sd.get(a + ".v_bias")
    k = F.linear(h, sd[a + ".key.weight"], torch.zeros_like(v_bias) if q_bias is not None else None)   # :347-348
    v = F.linear(h, sd[a + ".value.weight"], v_bias)                                 # :349
    q = F.linear(h, sd[a + ".query.weight"], q_bias)                                 # :350
    q, k, v = _split_heads(q, nh), _split_heads(k, nh), _split_heads(v, nh)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])          # :357-359
    probs = torch.softmax(scores, dim=-1)                                            # :362 (dropout p=0 :366)
    if head_mask is not None:                                                        # :368-370
        probs = probs * head_mask
    if attention_mask is not None:                                                   # :372-375  mask added AFTER softmax
        probs = probs + attention_mask.expand(-1, 1, attention_mask.shape[-1], -1)
    if probs_out is not None:
        probs_out.append(probs)                                                      # :389
    ctx = _merge_heads(torch.matmul(probs, v))                                       # :383-387
    attn_out = _lin(sd, p + ".attention.output.dense", ctx)                          # :419 (dropout p=0)
    x = attn_out + x                                                                 # :260
    h2 = _ln(sd, p + ".layernorm_after", x, eps)                                     # :263
    h2 = F.gelu(_lin(sd, p + ".intermediate.dense", h2))                             # :401-403, nn.GELU() exact
    return _lin(sd, p + ".output.dense", h2) + x                                     # :432-437


def fusion_encoder(sd, p, x, attention_mask, cfg, head_mask=None, probs_out=None):
    for i in range(cfg["layers"]):
        x = fusion_layer(sd, f"{p}.layer.{i}", x, attention_mask, cfg["heads"], cfg["eps"],
                         head_mask[i] if head_mask is not None else None, probs_out)  # :190
    return x                                                                         # plain tensor, :223


def transformer_encoder(sd, p, x, attention_mask, num_layers, nh, early_div=False):
    """utils/TAVFormer.py:144-166 with every nn.Dropout as identity (.eval()); post-LN, bias-free q/k/v."""
    for i in range(num_layers):
        lp = f"{p}.layers.{i}"
        sq = math.sqrt(x.shape[-1] // nh)
        q = F.linear(x, sd[lp + ".attention.query_matrix.weight"])
        if early_div:
            q = q / sq                                                               # :45-46
        k = F.linear(x, sd[lp + ".attention.key_matrix.weight"])
        v = F.linear(x, sd[lp + ".attention.value_matrix.weight"])
        q, k, v = _split_heads(q, nh), _split_heads(k, nh), _split_heads(v, nh)
        w = torch.matmul(q, k.transpose(-1, -2))
        if not early_div:
            w = w / sq                                                               # :62-63
        if attention_mask is not None:
            w = w + attention_mask.expand(-1, -1, attention_mask.shape[-1], -1)      # :68-72 (before softmax)
        pv = torch.matmul(torch.softmax(w, dim=-1), v)                               # [B, nh, S, d]
        B_, _, S_, d_ = pv.shape
        # :84 quirk: the reference holds scores as [B*nh, S, d], does .transpose(1,2).contiguous().view(B, S, nh*d):
        # per batch the memory is [nh][d][S] re-read as [S][nh*d] -- NOT a head merge.  Restated as-is.
        ctx = pv.reshape(B_ * nh, S_, d_).transpose(1, 2).contiguous().view(B_, S_, nh * d_)
        attn = _lin(sd, lp + ".attention.out", ctx)
        n1 = _ln(sd, lp + ".norm1", x + attn, 1e-5)                                  # :130-133
        ff = _lin(sd, lp + ".feed_forward.3", F.gelu(_lin(sd, lp + ".feed_forward.1", n1)))
        x = _ln(sd, lp + ".norm2", ff + n1, 1e-5)                                    # :135-139
    return x


# ------------------------------------------------------------------------------------------------ text
def text_position_ids(input_ids, cfg):
    if cfg["kind"] == "roberta":                                                     # HF roberta:142-155
        m = input_ids.ne(cfg["pad_id"]).int()
        return (torch.cumsum(m, dim=1).type_as(m) * m).long() + cfg["pad_id"]
    return torch.arange(input_ids.shape[1])[None].expand_as(input_ids)               # HF bert absolute positions


def text_embeddings(sd, p, cfg, input_ids):
    e = sd[p + ".word_embeddings.weight"][input_ids] + sd[p + ".token_type_embeddings.weight"][0]
    e = e + sd[p + ".position_embeddings.weight"][text_position_ids(input_ids, cfg)]
    return _ln(sd, p + ".LayerNorm", e, cfg["eps"])                                  # dropout off (eval)


def text_encoder(sd, p, cfg, input_ids, attention_mask):
    """HF RobertaModel/BertModel forward, eager attention, eval. Returns (sequence_output, pooled_output)."""
    x = text_embeddings(sd, p + ".embeddings", cfg, input_ids)
    add = None
    if attention_mask is not None:
        add = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * F32_MIN         # additive key mask
    nh = cfg["heads"]
    for i in range(cfg["layers"]):
        lp = f"{p}.encoder.layer.{i}"
        q = _split_heads(_lin(sd, lp + ".attention.self.query", x), nh)
        k = _split_heads(_lin(sd, lp + ".attention.self.key", x), nh)
        v = _split_heads(_lin(sd, lp + ".attention.self.value", x), nh)
        s = torch.matmul(q, k.transpose(2, 3)) * (q.shape[-1] ** -0.5)
        if add is not None:
            s = s + add
        ctx = _merge_heads(torch.matmul(torch.softmax(s, dim=-1), v))
        x = _ln(sd, lp + ".attention.output.LayerNorm", _lin(sd, lp + ".attention.output.dense", ctx) + x, cfg["eps"])
        h = _lin(sd, lp + ".output.dense", F.gelu(_lin(sd, lp + ".intermediate.dense", x)))
        x = _ln(sd, lp + ".output.LayerNorm", h + x, cfg["eps"])
    pooled = torch.tanh(_lin(sd, p + ".pooler.dense", x[:, 0]))                      # HF roberta:530-536
    return x, pooled


# ------------------------------------------------------------------------------------------------ audio
def w2v2_conv_out_lengths(lengths, cfg):
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):                         # models/tav.py:308-324
        lengths = torch.div(lengths - k, s, rounding_mode="floor") + 1
    return lengths


def w2v2_feature_extractor(sd, p, cfg, wave):
    """[B,T] -> [B,512,T'] (HF wav2vec2:382-419 with :256-323)."""
    h = wave[:, None]
    for i, s in enumerate(cfg["conv_stride"]):
        lp = f"{p}.conv_layers.{i}"
        h = F.conv1d(h, sd[lp + ".conv.weight"], sd.get(lp + ".conv.bias"), stride=s)
        if cfg["feat_norm"] == "layer":
            h = F.layer_norm(h.transpose(-2, -1), (h.shape[1],), sd[lp + ".layer_norm.weight"], sd[lp + ".layer_norm.bias"], 1e-5).transpose(-2, -1)
        elif i == 0:
            h = F.group_norm(h, h.shape[1], sd[lp + ".layer_norm.weight"], sd[lp + ".layer_norm.bias"], 1e-5)
        h = F.gelu(h)
    return h


def w2v2_feature_projection(sd, p, cfg, feats):
    normed = _ln(sd, p + ".layer_norm", feats, cfg["eps"])                           # HF wav2vec2:422-434
    return _lin(sd, p + ".projection", normed), normed


def w2v2_pos_conv(sd, p, cfg, hidden):
    """HF wav2vec2:326-379: weight-normed grouped conv (dim=2), drop last step, GELU."""
    g, v = sd[p + ".conv.parametrizations.weight.original0"], sd[p + ".conv.parametrizations.weight.original1"]
    w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    k = cfg["pos_k"]
    h = F.conv1d(hidden.transpose(1, 2), w, sd[p + ".conv.bias"], padding=k // 2, groups=cfg["pos_groups"])
    if k % 2 == 0:
        h = h[:, :, :-1]
    return F.gelu(h).transpose(1, 2)


def _w2v2_attention(sd, p, cfg, x):
    nh = cfg["heads"]
    q = _split_heads(_lin(sd, p + ".q_proj", x), nh)
    k = _split_heads(_lin(sd, p + ".k_proj", x), nh)
    v = _split_heads(_lin(sd, p + ".v_proj", x), nh)
    s = torch.matmul(q, k.transpose(2, 3)) * (q.shape[-1] ** -0.5)
    return _lin(sd, p + ".out_proj", _merge_heads(torch.matmul(torch.softmax(s, dim=-1), v)))


def _w2v2_ff(sd, p, x):
    return _lin(sd, p + ".output_dense", F.gelu(_lin(sd, p + ".intermediate_dense", x)))


def w2v2_encoder(sd, p, cfg, hidden):
    """HF wav2vec2:729-802 (stable LN) / :657-726 (post LN); no attention mask on the path, eval (no layerdrop)."""
    hidden = hidden + w2v2_pos_conv(sd, p + ".pos_conv_embed", cfg, hidden)
    if not cfg["stable_ln"]:
        hidden = _ln(sd, p + ".layer_norm", hidden, cfg["eps"])
    for i in range(cfg["layers"]):
        lp = f"{p}.layers.{i}"
        if cfg["stable_ln"]:                                                         # :611-654
            hidden = hidden + _w2v2_attention(sd, lp + ".attention", cfg, _ln(sd, lp + ".layer_norm", hidden, cfg["eps"]))
            hidden = hidden + _w2v2_ff(sd, lp + ".feed_forward", _ln(sd, lp + ".final_layer_norm", hidden, cfg["eps"]))
        else:                                                                        # :575-608
            hidden = _ln(sd, lp + ".layer_norm", hidden + _w2v2_attention(sd, lp + ".attention", cfg, hidden), cfg["eps"])
            hidden = _ln(sd, lp + ".final_layer_norm", hidden + _w2v2_ff(sd, lp + ".feed_forward", hidden), cfg["eps"])
    if cfg["stable_ln"]:
        hidden = _ln(sd, p + ".layer_norm", hidden, cfg["eps"])
    return hidden


def w2v2_model(sd, p, cfg, wave):
    """Wav2Vec2Model.forward(input_values) with no attention_mask (models/tav.py:476), eval mode."""
    feats = w2v2_feature_extractor(sd, p + ".feature_extractor", cfg, wave).transpose(1, 2)
    hidden, _ = w2v2_feature_projection(sd, p + ".feature_projection", cfg, feats)
    return w2v2_encoder(sd, p + ".encoder", cfg, hidden)


# ------------------------------------------------------------------------------------------------ video
def sinusoid_table(n_position, d_hid):
    """HF videomae:80-91 (numpy, float64 -> float32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table)


def videomae_num_patches(cfg):
    return (cfg["image"] // cfg["patch"]) ** 2 * (cfg["frames"] // cfg["tubelet"])


def videomae_embeddings(sd, p, cfg, video, bool_masked_pos):
    """video [B,F,3,H,W] -> visible tokens [B,Nvis,768] (HF videomae:94-177)."""
    x = video.permute(0, 2, 1, 3, 4)
    t, ps = cfg["tubelet"], cfg["patch"]
    e = F.conv3d(x, sd[p + ".patch_embeddings.projection.weight"], sd[p + ".patch_embeddings.projection.bias"], stride=(t, ps, ps))
    e = e.flatten(2).transpose(1, 2)
    e = e + sinusoid_table(videomae_num_patches(cfg), cfg["hidden"])[None]
    if bool_masked_pos is not None:
        B, _, C = e.shape
        e = e[~bool_masked_pos].reshape(B, -1, C)
    return e


def videomae_encoder(sd, p, cfg, x):
    nh = cfg["heads"]
    for i in range(cfg["layers"]):
        lp = f"{p}.layer.{i}"
        h = _ln(sd, lp + ".layernorm_before", x, cfg["eps"])
        a = lp + ".attention.attention"
        q, k, v = (_split_heads(_lin(sd, a + n, h), nh) for n in (".query", ".key", ".value"))
        s = torch.matmul(q, k.transpose(2, 3)) * (q.shape[-1] ** -0.5)
        ctx = _merge_heads(torch.matmul(torch.softmax(s, dim=-1), v))
        x = _lin(sd, lp + ".attention.output.dense", ctx) + x
        h2 = F.gelu(_lin(sd, lp + ".intermediate.dense", _ln(sd, lp + ".layernorm_after", x, cfg["eps"])))
        x = _lin(sd, lp + ".output.dense", h2) + x
    return x


def videomae_model(sd, p, cfg, video, bool_masked_pos):
    """VideoMAEModel.forward; use_mean_pooling=True -> no final LayerNorm (HF videomae:406-409,462-464)."""
    return videomae_encoder(sd, p + ".encoder", cfg, videomae_embeddings(sd, p + ".embeddings", cfg, video, bool_masked_pos))


# ------------------------------------------------------------------------------------------------ PreFormer
def feature_vector_attention_mask(feature_len, attention_mask, cfg):
    """models/tav.py:326-342."""
    lengths = w2v2_conv_out_lengths(attention_mask.cumsum(dim=-1)[:, -1], cfg).to(torch.long)
    B = attention_mask.shape[0]
    m = torch.zeros((B, feature_len), dtype=attention_mask.dtype)
    m[(torch.arange(B), lengths - 1)] = 1
    return m.flip([-1]).cumsum(-1).flip([-1]).bool()


def preformer_forward(sd, cfg, input_ids, audio_features, video_embeds, text_mask, audio_mask, visual_mask):
    """models/tav.py:344-417 with train=False (no SpecAugment). Returns (tav, tav_embed, attention_mask)."""
    emb_text = text_embeddings(sd, "bert.embeddings", cfg["text"], input_ids)                         # :349
    feats = w2v2_feature_extractor(sd, "wav2vec2.feature_extractor", cfg["audio"], audio_features)    # :352
    amask = feature_vector_attention_mask(feats.shape[2], audio_mask, cfg["audio"])                   # :355
    emb_audio, _ = w2v2_feature_projection(sd, "wav2vec2.feature_projection", cfg["audio"], feats.transpose(1, 2))   # :356
    emb_audio = emb_audio + w2v2_pos_conv(sd, "wav2vec2.encoder.pos_conv_embed", cfg["audio"], emb_audio)           # :360
    emb_audio = _ln(sd, "wav2vec2.encoder.layer_norm", emb_audio, cfg["audio"]["eps"])                # :361
    emb_audio = _lin(sd, "wav_2_768", emb_audio)                                                      # :363
    emb_video = videomae_embeddings(sd, "videomae.embeddings", cfg["video"], video_embeds, ~visual_mask)            # :368
    if cfg["video"]["hidden"] != 768:        # BASELINE config 5 (videomae-large): Linear(1024, 768), the analogue of wav_2_768 at :363 (no reference line)
        emb_video = _lin(sd, "vid_2_768", emb_video)
    tav = torch.concat((emb_text, emb_audio, emb_video), dim=1)                                       # :372
    B = tav.shape[0]
    St, Sa, Sv = emb_text.shape[1], emb_audio.shape[1], emb_video.shape[1]
    tmask = (1.0 - text_mask[:, None, None, :]) * FP16_MIN                                            # :383
    amask4 = 1.0 - amask[:, None, None, :] * FP16_MIN                                                 # :390 (precedence quirk)
    vmask = torch.zeros((B, 1, 1, Sv))                                                                # :397
    tav_embed = torch.concat((torch.zeros(B, St), torch.ones(B, Sa), torch.ones(B, Sv) + 1), dim=1).long()   # :381-405
    attention_mask = torch.concat((tmask, amask4, vmask), dim=-1)                                     # :409
    return tav, tav_embed, attention_mask


# ------------------------------------------------------------------------------------------------ TAVForMAE
def tavformae_forward(sd, cfg, input_ids, text_attention_mask, audio_features, video_embeds, visual_mask, hidden_states,
                      pos_embed, attention_mask, check="val", dropout_mask=None, dropout_p=0.0):
    """models/tav.py:473-504.  check == 'train' applies dropout on the [B,3072] concat; pass the keep-mask to pin it."""
    av = hidden_states + sd["embedding.weight"][pos_embed]                                            # :474
    aud = w2v2_model(sd, "wav2vec2", cfg["audio"], audio_features)                                    # :476
    aud = torch.mean(_lin(sd, "wav_2_768_2", aud), dim=1)                                             # :478
    vid = videomae_model(sd, "videomae", cfg["video"], video_embeds, visual_mask)                    # :480
    if cfg["video"]["hidden"] != 768:        # config 5: Linear(1024, 768), the analogue of wav_2_768_2 at :478 (no reference line)
        vid = _lin(sd, "vid_2_768_2", vid)
    vid = torch.mean(vid, dim=1)                                                                      # :481
    _, t = text_encoder(sd, "bert", cfg["text"], input_ids, text_attention_mask)                      # :485
    t = _ln(sd, "bert_norm", t, 1e-5)                                                                 # :486
    av = fusion_encoder(sd, "random_mae_encoder", av, attention_mask, cfg["fusion"])                  # :487
    av = _ln(sd, "rand_norm", torch.mean(av, dim=1), 1e-5)                                            # :488
    aud = _ln(sd, "aud_norm", aud, 1e-5)                                                              # :489
    vid = _ln(sd, "vid_norm", vid, 1e-5)                                                              # :490
    tav = torch.cat([av, t, aud, vid], dim=1)                                                         # :495
    if check == "train" and dropout_mask is not None:                                                 # :497-498
        tav = tav * dropout_mask / (1.0 - dropout_p)
    return _lin(sd, "linear1", tav)                                                                   # :499


def randomize_model_(model):
    """M1, models/tav.py:442 + :461-471: `VideoMAEEncoder(...).apply(self.randomize_model)`.  `apply` visits every sub-module (children
    first) and the rule itself walks `named_modules()` of what it is handed, so a Linear nested d levels deep is redrawn d + 1 times; the
    LAST draw (the call on the root) is the one that stays.  Restated with the same traversal so that, from one RNG state, the resulting
    weights equal the reference's bit for bit (pinned by oracle/validate_vs_reference.py against the reference's own function body):
    xavier_uniform on every Linear / Embedding weight, Linear bias 0, LayerNorm weight := ones(768) (a fresh Parameter, :467), bias 0;
    q_bias / v_bias are not touched (they stay 0 from utils/TAVFormer.py:330-331)."""
    def rule(sub):
        for _, m in sub.named_modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Embedding)):
                torch.nn.init.xavier_uniform_(m.weight)
            elif isinstance(m, torch.nn.LayerNorm):
                m.bias.data.zero_()
                m.weight = torch.nn.Parameter(torch.ones(768))
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
        return sub
    return model.apply(rule)


def new_cross_entropy(logits, target, epoch, epoch_switch, class_weights):
    """utils/global_functions.py:69-83."""
    if epoch % epoch_switch == 0:
        return F.cross_entropy(logits, target)
    return F.cross_entropy(logits, target, weight=class_weights)


def tav_step(sd_model, sd_pre, cfg, batch, labels, class_weights=None, epoch=0, epoch_switch=2):
    """train_model/tav_train.py:15-48 (get_statistics) with check='val', train=False: PreFormer -> model -> loss."""
    tav, tav_embed, amask = preformer_forward(sd_pre, cfg, batch["input_ids"], batch["audio_features"], batch["video_embeds"],
                                              batch["text_mask"], batch["audio_mask"], batch["visual_mask"])
    logits = tavformae_forward(sd_model, cfg, batch["input_ids"], batch["text_mask"], batch["audio_features"], batch["video_embeds"],
                               batch["visual_mask"], tav, tav_embed, amask, check="val")
    if class_weights is None:
        loss = F.cross_entropy(logits, labels)
    else:
        loss = new_cross_entropy(logits, labels, epoch, epoch_switch, class_weights)
    return logits, loss


# ------------------------------------------------------------------------------------------------ SURVEY.md §8(f) rows 1-2
def text_classifier_forward(sd, cfg, input_id, mask, check="val", dropout_mask=None, dropout_p=0.0):
    """SingleModels/models/text.py:41-69 (BertClassifier.forward): pooled output -> dropout iff check == 'train' -> Linear(768, out)."""
    _, x = text_encoder(sd, "bert", cfg["text"], input_id, mask)                                      # :58
    if check == "train" and dropout_mask is not None:                                                 # :61-62
        x = x * dropout_mask / (1.0 - dropout_p)
    return _lin(sd, "linear", x)                                                                      # :65


def text_audio_forward(sd, cfg, input_ids, text_attention_mask, audio_features, check="val", dropout_mask=None, dropout_p=0.0):
    """Dual text+audio classifier (BASELINE config 4).  The reference file DoubleModels/models/text_audio.py does not parse
    (SURVEY.md §8f.2), so the path is DEFINED as models/tav.py:473-499 minus the video and fusion branches:
    logits = Linear(1536, out)(dropout(cat[LN(pooled_text), LN(mean_t(Linear(H_a,768)(wav2vec2(audio))))]))."""
    aud = w2v2_model(sd, "wav2vec2", cfg["audio"], audio_features)                                    # tav.py:476
    aud = torch.mean(_lin(sd, "wav_2_768_2", aud), dim=1)                                             # :478
    _, t = text_encoder(sd, "bert", cfg["text"], input_ids, text_attention_mask)                      # :485
    t = _ln(sd, "bert_norm", t, 1e-5)                                                                 # :486
    aud = _ln(sd, "aud_norm", aud, 1e-5)                                                              # :489
    ta = torch.cat([t, aud], dim=1)
    if check == "train" and dropout_mask is not None:
        ta = ta * dropout_mask / (1.0 - dropout_p)
    return _lin(sd, "linear1", ta)
