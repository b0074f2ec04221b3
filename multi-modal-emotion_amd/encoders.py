"""Parameter containers with the reference's / Hugging Face's state_dict key names, and the forward orchestration of the
three encoders on the libtavhip engine.  The torch.nn modules below (nn.Linear, nn.LayerNorm, nn.Embedding ...) are
used ONLY as named parameter holders so that `state_dict()` matches what a reference `best.pt` contains
(SURVEY.md §8b); their own forward() is never called -- all arithmetic goes through engine.* -> libtavhip.

  TextEncoder   <- HF RobertaModel / BertModel      (keys: embeddings.*, encoder.layer.N.*, pooler.dense.*)
  AudioEncoder  <- HF Wav2Vec2Model                 (keys: feature_extractor.*, feature_projection.*, encoder.*)
  VideoEncoder  <- HF VideoMAEModel                 (keys: embeddings.patch_embeddings.projection.*, encoder.layer.N.*)
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import engine as E
from . import ops, runtime


_DEBUG_CHECKS = os.environ.get("TAV_DEBUG_CHECKS", "0") == "1"      # host-synchronising input checks on the sync-free paths


def _holder(**children):
    m = nn.Module()
    for k, v in children.items():
        setattr(m, k, v)
    return m


def _init_linear(m, std=0.02):
    nn.init.normal_(m.weight, 0.0, std)
    if m.bias is not None:
        nn.init.zeros_(m.bias)
    return m


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: libtavhip runs on the GPU only (tensor is on {t.device}); there is no CPU fallback. "
                           "Move the module and its inputs to 'cuda'.")


# ------------------------------------------------------------------------------------------------ text
class TextEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = dict(cfg)
        H, F = cfg["hidden"], cfg["inter"]
        self.embeddings = _holder(
            word_embeddings=nn.Embedding(cfg["vocab"], H), position_embeddings=nn.Embedding(cfg["max_pos"], H),
            token_type_embeddings=nn.Embedding(cfg["type_vocab"], H), LayerNorm=nn.LayerNorm(H, eps=cfg["eps"]))
        layers = []
        for _ in range(cfg["layers"]):
            att = _holder(self=_holder(query=nn.Linear(H, H), key=nn.Linear(H, H), value=nn.Linear(H, H)),
                          output=_holder(dense=nn.Linear(H, H), LayerNorm=nn.LayerNorm(H, eps=cfg["eps"])))
            layers.append(_holder(attention=att, intermediate=_holder(dense=nn.Linear(H, F)),
                                  output=_holder(dense=nn.Linear(F, H), LayerNorm=nn.LayerNorm(H, eps=cfg["eps"]))))
        self.encoder = _holder(layer=nn.ModuleList(layers))
        self.pooler = _holder(dense=nn.Linear(H, H))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _init_linear(m)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, 0.0, 0.02)

    def _layer_params(self, L):
        a, o = L.attention.self, L.attention.output
        return (o.LayerNorm.weight, o.LayerNorm.bias, a.query.weight, a.query.bias, a.key.weight, a.key.bias, a.value.weight, a.value.bias,
                o.dense.weight, o.dense.bias, L.output.LayerNorm.weight, L.output.LayerNorm.bias, L.intermediate.dense.weight,
                L.intermediate.dense.bias, L.output.dense.weight, L.output.dense.bias)

    def embed(self, input_ids):
        """HF embeddings (roberta:75-121).  Returns (x f32 [B*S,H], x_lp)."""
        ectx = runtime.ctx()
        _need_cuda(input_ids, "TextEncoder.embed")
        e, c = self.embeddings, self.cfg
        pad = c["pad_id"] if c["kind"] == "roberta" else -1
        return E.text_embed(ectx, input_ids.contiguous(), e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                            e.LayerNorm.weight, e.LayerNorm.bias, c["eps"], pad)

    def forward(self, input_ids, attention_mask=None):
        """Returns (sequence_output f32 [B,S,H], pooled_output f32 [B,H]) like HF `return_dict=False`."""
        ectx = runtime.ctx()
        B, S = input_ids.shape
        c = self.cfg
        x, x_lp = self.embed(input_ids)
        key_mask, mode = None, 0
        if attention_mask is not None:
            # HF: (1 - mask) * finfo.min added to the scores before softmax (float 0/1 masks accepted)
            key_mask = ((1.0 - attention_mask.to(torch.float32)) * torch.finfo(torch.float32).min).contiguous()
            mode = 1
        spec = E.LayerSpec(B, S, c["heads"], c["eps"], pre_ln=False, mask_mode=mode, branch="text")
        n = len(self.encoder.layer)
        for i, L in enumerate(self.encoder.layer):
            x = runtime.cut_point("text", i, n, x)
            x, x_lp = E.encoder_layer(ectx, spec, x, x_lp, key_mask, self._layer_params(L))
        seq = x.view(B, S, -1)
        first = seq[:, 0]                                                   # HF roberta:530-536
        pre = E.LinearFn.apply(first, None, self.pooler.dense.weight, self.pooler.dense.bias, None, ectx, True)
        return seq, TanhFn.apply(pre)


class TanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.tanh_fwd(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.tanh_bwd(y, g.contiguous())


# ------------------------------------------------------------------------------------------------ audio
def _wn_conv_holder(H, Cg, K):
    """nn.utils.parametrizations.weight_norm(conv, dim=2) key names: conv.bias, conv.parametrizations.weight.original{0,1}"""
    p = nn.Module()
    p.original0 = nn.Parameter(torch.ones(1, 1, K))
    p.original1 = nn.Parameter(torch.empty(H, Cg, K))
    conv = nn.Module()
    conv.bias = nn.Parameter(torch.zeros(H))
    conv.parametrizations = _holder(weight=p)
    return conv


class AudioEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = dict(cfg)
        H, F = cfg["hidden"], cfg["inter"]
        self.masked_spec_embed = nn.Parameter(torch.empty(H).uniform_())
        convs, cin = [], 1
        for i, (cd, k, s) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"], cfg["conv_stride"])):
            layer = nn.Module()
            layer.conv = nn.Conv1d(cin, cd, k, stride=s, bias=cfg["conv_bias"])
            nn.init.kaiming_normal_(layer.conv.weight)
            if cfg["feat_norm"] == "layer" or i == 0:
                layer.layer_norm = nn.LayerNorm(cd)      # GroupNorm(cd, cd) has the same [cd] weight/bias keys
            convs.append(layer)
            cin = cd
        self.feature_extractor = _holder(conv_layers=nn.ModuleList(convs))
        self.feature_projection = _holder(layer_norm=nn.LayerNorm(cfg["conv_dim"][-1], eps=cfg["eps"]), projection=_init_linear(nn.Linear(cfg["conv_dim"][-1], H)))
        Cg, K = H // cfg["pos_groups"], cfg["pos_k"]
        pos = _wn_conv_holder(H, Cg, K)
        nn.init.normal_(pos.parametrizations.weight.original1, 0.0, 2 * math.sqrt(1 / (K * H)))
        with torch.no_grad():
            v = pos.parametrizations.weight.original1
            pos.parametrizations.weight.original0.copy_(v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        layers = []
        for _ in range(cfg["layers"]):
            att = _holder(k_proj=nn.Linear(H, H), v_proj=nn.Linear(H, H), q_proj=nn.Linear(H, H), out_proj=nn.Linear(H, H))
            layers.append(_holder(attention=att, layer_norm=nn.LayerNorm(H, eps=cfg["eps"]),
                                  feed_forward=_holder(intermediate_dense=nn.Linear(H, F), output_dense=nn.Linear(F, H)),
                                  final_layer_norm=nn.LayerNorm(H, eps=cfg["eps"])))
        self.encoder = _holder(pos_conv_embed=_holder(conv=pos), layer_norm=nn.LayerNorm(H, eps=cfg["eps"]), layers=nn.ModuleList(layers))
        for L in layers:
            for m in L.modules():
                if isinstance(m, nn.Linear):
                    _init_linear(m)

    def conv_out_len(self, T):
        for k, s in zip(self.cfg["conv_kernel"], self.cfg["conv_stride"]):
            T = (T - k) // s + 1
        return T

    def _layer_params(self, L):
        a, f = L.attention, L.feed_forward
        return (L.layer_norm.weight, L.layer_norm.bias, a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias, a.v_proj.weight, a.v_proj.bias,
                a.out_proj.weight, a.out_proj.bias, L.final_layer_norm.weight, L.final_layer_norm.bias, f.intermediate_dense.weight,
                f.intermediate_dense.bias, f.output_dense.weight, f.output_dense.bias)

    # ---- pieces (the reference's PreFormer calls them one by one, models/tav.py:352-362) ----
    def feature_extractor_fwd(self, wave):
        """raw waveform f32 [B,T] -> channels-last features [B, T', 512] in the operand dtype (HF wav2vec2:382-419)."""
        ectx, c = runtime.ctx(), self.cfg
        _need_cuda(wave, "AudioEncoder")
        layers = self.feature_extractor.conv_layers
        L0 = layers[0]
        h = E.Conv0Fn.apply(wave, L0.conv.weight, L0.conv.bias, c["conv_stride"][0], ectx)
        if c["feat_norm"] == "group":
            h = E.GroupNormGeluFn.apply(h, L0.layer_norm.weight, L0.layer_norm.bias, 1e-5)
        else:
            B, T, C = h.shape
            h = E.layer_norm_lp(ectx, h.view(B * T, C), L0.layer_norm.weight, L0.layer_norm.bias, 1e-5, act=1).view(B, T, C)
        u = None                                              # pre-activation of the producing conv layer (chains the GELU backward into col2im)
        for i in range(1, len(layers)):
            L = layers[i]
            fuse_gelu = c["feat_norm"] == "group"
            h, u = E.ConvGemmFn.apply(h, u, L.conv.weight, L.conv.bias, c["conv_stride"][i], fuse_gelu, ectx)
            if not fuse_gelu:
                B, T, C = h.shape
                h = E.layer_norm_lp(ectx, h.view(B * T, C), L.layer_norm.weight, L.layer_norm.bias, 1e-5, act=1).view(B, T, C)
        return h

    def feature_projection_fwd(self, feats):
        """[B,T',512] -> hidden f32 [B*T', H]  (LayerNorm -> Linear; HF wav2vec2:422-434, dropout off)."""
        ectx, fp = runtime.ctx(), self.feature_projection
        B, T, C = feats.shape
        n = E.layer_norm_lp(ectx, feats.reshape(B * T, C), fp.layer_norm.weight, fp.layer_norm.bias, self.cfg["eps"])
        return E.LinearFn.apply(None, n, fp.projection.weight, fp.projection.bias, None, ectx, True)

    def pos_conv_fwd(self, hidden, B, T):
        """hidden + pos_conv_embed(hidden), f32 [B*T, H]."""
        pc = self.encoder.pos_conv_embed.conv
        w = pc.parametrizations.weight
        return E.PosConvFn.apply(hidden, w.original1, w.original0, pc.bias, B, T, self.cfg["pos_groups"], runtime.ctx())

    def encoder_fwd(self, hidden, B, T):
        """HF Wav2Vec2Encoder / EncoderStableLayerNorm without attention mask, eval (no layerdrop/dropout)."""
        ectx, c = runtime.ctx(), self.cfg
        x = self.pos_conv_fwd(hidden, B, T)
        x_lp = None
        if not c["stable_ln"]:
            x, x_lp = E.layer_norm_f32(ectx, x, self.encoder.layer_norm.weight, self.encoder.layer_norm.bias, c["eps"])
        spec = E.LayerSpec(B, T, c["heads"], c["eps"], pre_ln=c["stable_ln"], mask_mode=0, branch="audio")
        n = len(self.encoder.layers)
        for i, L in enumerate(self.encoder.layers):
            x = runtime.cut_point("audio", i, n, x)
            x, x_lp = E.encoder_layer(ectx, spec, x, x_lp, None, self._layer_params(L))
        if c["stable_ln"]:
            x, x_lp = E.layer_norm_f32(ectx, x, self.encoder.layer_norm.weight, self.encoder.layer_norm.bias, c["eps"])
        return x, x_lp

    def forward(self, wave):
        """Wav2Vec2Model(input_values)[0] (no attention_mask, eval): returns (last_hidden f32 [B*T', H], lp copy, T')."""
        feats = self.feature_extractor_fwd(wave)
        B, T, _ = feats.shape
        hidden = self.feature_projection_fwd(feats)
        x, x_lp = self.encoder_fwd(hidden, B, T)
        return x, x_lp, T


# ------------------------------------------------------------------------------------------------ video
def sinusoid_table(n_position, d_hid):
    """Fixed sin-cos table of HF videomae:80-91 (float64 numpy -> float32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table)


class VideoEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = dict(cfg)
        H, F = cfg["hidden"], cfg["inter"]
        proj = nn.Conv3d(3, H, kernel_size=(cfg["tubelet"], cfg["patch"], cfg["patch"]), stride=(cfg["tubelet"], cfg["patch"], cfg["patch"]))
        self.embeddings = _holder(patch_embeddings=_holder(projection=proj))
        self.num_patches = (cfg["image"] // cfg["patch"]) ** 2 * (cfg["frames"] // cfg["tubelet"])
        self.register_buffer("_pos_table", sinusoid_table(self.num_patches, H), persistent=False)
        layers = []
        for _ in range(cfg["layers"]):
            att = _holder(attention=_holder(query=nn.Linear(H, H), key=nn.Linear(H, H), value=nn.Linear(H, H)), output=_holder(dense=nn.Linear(H, H)))
            layers.append(_holder(attention=att, intermediate=_holder(dense=nn.Linear(H, F)), output=_holder(dense=nn.Linear(F, H)),
                                  layernorm_before=nn.LayerNorm(H, eps=cfg["eps"]), layernorm_after=nn.LayerNorm(H, eps=cfg["eps"])))
        self.encoder = _holder(layer=nn.ModuleList(layers))
        for L in layers:
            for m in L.modules():
                if isinstance(m, nn.Linear):
                    _init_linear(m)
        if cfg["tubelet"] != 2 or cfg["patch"] != 16:
            raise ValueError("libtavhip's patch gather is built for tubelet 2 / patch 16 (videomae-base/large)")

    def _layer_params(self, L):
        a = L.attention.attention
        return (L.layernorm_before.weight, L.layernorm_before.bias, a.query.weight, a.query.bias, a.key.weight, a.key.bias, a.value.weight, a.value.bias,
                L.attention.output.dense.weight, L.attention.output.dense.bias, L.layernorm_after.weight, L.layernorm_after.bias,
                L.intermediate.dense.weight, L.intermediate.dense.bias, L.output.dense.weight, L.output.dense.bias)

    def embed(self, video, bool_masked_pos, nkeep=None):
        """VideoMAEEmbeddings(pixel_values, bool_masked_pos): tokens where bool_masked_pos is False, in order.
        Returns (x f32 [B*nkeep, H], nkeep).  Every row must keep the same number of tokens (HF's reshape(B,-1,C))."""
        ectx = runtime.ctx()
        _need_cuda(video, "VideoEncoder")
        B = video.shape[0]
        host_checked = nkeep is None
        if nkeep is None:
            total = int((~bool_masked_pos).sum().item())        # host sync; pass nkeep to avoid it
            if total % B:
                raise ValueError("visible token count not divisible by the batch size")
            nkeep = total // B
        if not 0 < nkeep <= bool_masked_pos.shape[1]:
            raise ValueError(f"VideoEncoder.embed: {nkeep} kept tokens per row out of {bool_masked_pos.shape[1]}")
        idx, counts = ops.mask_to_index(bool_masked_pos.contiguous(), False, nkeep)
        # HF's `embeddings[~mask].reshape(B, -1, C)` mixes utterances when rows keep different numbers of tokens (the reference's
        # collate only equalises the TOTAL, models/tav.py:211-217); here that is an error, not a silent shuffle.  The kernels are
        # safe either way (mask_to_index pads short rows with a valid index); the check costs a host read, so it runs when the
        # host already synchronised to count, or on request (TAV_DEBUG_CHECKS=1) when the caller passed nkeep / n_visual_true.
        if host_checked or _DEBUG_CHECKS:
            c = counts.cpu()
            if not bool((c == nkeep).all()):
                raise ValueError(f"visual mask keeps {c.tolist()} tokens per row; every row must keep the same number ({nkeep})")
        p = self.embeddings.patch_embeddings.projection
        x = E.PatchEmbedFn.apply(video, idx, p.weight, p.bias, self._pos_table, ectx)
        return x, nkeep

    def forward(self, video, bool_masked_pos, nkeep=None):
        """VideoMAEModel(pixel_values, bool_masked_pos)[0] with use_mean_pooling=True (no final LayerNorm)."""
        ectx, c = runtime.ctx(), self.cfg
        x, nkeep = self.embed(video, bool_masked_pos, nkeep)
        spec = E.LayerSpec(video.shape[0], nkeep, c["heads"], c["eps"], pre_ln=True, mask_mode=0, branch="video")
        n = len(self.encoder.layer)
        for i, L in enumerate(self.encoder.layer):
            x = runtime.cut_point("video", i, n, x)
            x, _ = E.encoder_layer(ectx, spec, x, None, None, self._layer_params(L))
        return x, nkeep
