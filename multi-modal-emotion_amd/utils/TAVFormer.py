"""Drop-in for the reference's utils/TAVFormer.py: the two fusion encoder stacks, same constructor arguments, forward
signatures and state_dict keys, executed by libtavhip (HIP/gfx950) instead of eager ATen.

  VideoMAEEncoder(config, num_layers)   reference utils/TAVFormer.py:171-439  (the fusion encoder TAVForMAE uses;
        pre-LN blocks; the additive attention_mask [B,1,1,S] is applied AFTER the softmax, :372-375)
  TransformerEncoder(embed_dim, ...)    reference utils/TAVFormer.py:10-166   (post-LN alternative stack; mask BEFORE softmax)
"""
import torch
from torch import nn

from .. import engine as E
from .. import runtime


def _cfg_get(config, name, default=None):
    if isinstance(config, dict):
        return config.get(name, default)
    return getattr(config, name, default)


def _holder(**children):
    m = nn.Module()
    for k, v in children.items():
        setattr(m, k, v)
    return m


class VideoMAESelfAttention(nn.Module):
    """Parameter holder with the reference's names (:312-341): bias-free query/key/value + q_bias / v_bias."""

    def __init__(self, hidden, qkv_bias=True):
        super().__init__()
        self.query = nn.Linear(hidden, hidden, bias=False)
        self.key = nn.Linear(hidden, hidden, bias=False)
        self.value = nn.Linear(hidden, hidden, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(hidden))
            self.v_bias = nn.Parameter(torch.zeros(hidden))
        else:
            self.q_bias = None
            self.v_bias = None


class VideoMAELayer(nn.Module):
    def __init__(self, hidden, inter, eps, qkv_bias=True):
        super().__init__()
        self.layernorm_before = nn.LayerNorm(hidden, eps=eps)
        self.attention = _holder(attention=VideoMAESelfAttention(hidden, qkv_bias), output=_holder(dense=nn.Linear(hidden, hidden)))
        self.layernorm_after = nn.LayerNorm(hidden, eps=eps)
        self.intermediate = _holder(dense=nn.Linear(hidden, inter))
        self.output = _holder(dense=nn.Linear(inter, hidden))

    def params(self):
        a = self.attention.attention
        return (self.layernorm_before.weight, self.layernorm_before.bias, a.query.weight, a.q_bias, a.key.weight, None, a.value.weight, a.v_bias,
                self.attention.output.dense.weight, self.attention.output.dense.bias, self.layernorm_after.weight, self.layernorm_after.bias,
                self.intermediate.dense.weight, self.intermediate.dense.bias, self.output.dense.weight, self.output.dense.bias)


class VideoMAEEncoder(nn.Module):
    def __init__(self, config, num_layers: int) -> None:
        super().__init__()
        self.config = config
        hidden = _cfg_get(config, "hidden_size", _cfg_get(config, "hidden", 768))
        self.num_heads = _cfg_get(config, "num_attention_heads", _cfg_get(config, "heads", 12))
        inter = _cfg_get(config, "intermediate_size", _cfg_get(config, "inter", 3072))
        self.eps = _cfg_get(config, "layer_norm_eps", _cfg_get(config, "eps", 1e-12))
        qkv_bias = _cfg_get(config, "qkv_bias", True)
        if hidden != self.num_heads * 64:
            raise ValueError("libtavhip attention is built for head_dim 64")
        self.layer = nn.ModuleList([VideoMAELayer(hidden, inter, self.eps, qkv_bias) for _ in range(num_layers)])
        self.gradient_checkpointing = False

    def forward(self, hidden_states, attention_mask=None, head_mask=None, output_attentions: bool = False, output_hidden_states: bool = False,
                return_dict: bool = True):
        if not hidden_states.is_cuda:
            raise RuntimeError("VideoMAEEncoder runs on libtavhip (GPU) only; there is no CPU fallback")
        ectx = runtime.ctx()
        B, S, H = hidden_states.shape
        key_mask, mode = None, 0
        if attention_mask is not None:
            key_mask = attention_mask.reshape(B, S).to(torch.float32).contiguous()      # [B,1,1,S] broadcast over heads and queries
            mode = 2
        spec = E.LayerSpec(B, S, self.num_heads, self.eps, pre_ln=True, mask_mode=mode, branch="fusion")
        x = hidden_states.reshape(B * S, H)
        all_hidden = () if output_hidden_states else None
        all_attn = [] if output_attentions else None
        for i, layer in enumerate(self.layer):
            if output_hidden_states:
                all_hidden = all_hidden + (x.view(B, S, H),)
            x = runtime.cut_point("fusion", i, len(self.layer), x)
            lspec = spec
            if head_mask is not None or output_attentions:
                # slow path (reference utils/TAVFormer.py:190, :368-370, :389): a per-head factor on the probabilities before the mask is
                # added, and the probabilities themselves as a (detached) f32 tensor.  The training loop never asks for either.
                lspec = E.LayerSpec(B, S, self.num_heads, self.eps, pre_ln=True, mask_mode=mode, branch="fusion",
                                    head_scale=self._head_scale(head_mask[i], B) if head_mask is not None and head_mask[i] is not None else None,
                                    probs_out=all_attn)
            x, _ = E.encoder_layer(ectx, lspec, x, None, key_mask, layer.params())
        out = x.view(B, S, H)
        if output_hidden_states:
            all_hidden = all_hidden + (out,)
        if not return_dict:
            return tuple(v for v in [out, all_hidden, tuple(all_attn) if all_attn is not None else None] if v is not None)
        return out

    def _head_scale(self, layer_head_mask, B):
        """The layer's head_mask entry as per-head factors: anything that broadcasts over queries and keys -- a scalar, [heads],
        [1, heads, 1, 1] or [B, heads, 1, 1] (what HF's get_head_mask produces) -> f32 [heads] or [B, heads]."""
        m = torch.as_tensor(layer_head_mask, dtype=torch.float32, device=next(self.parameters()).device)
        nh = self.num_heads
        if m.numel() == 1:
            return m.reshape(1).expand(nh).contiguous()
        if m.numel() == nh and (m.dim() == 1 or tuple(m.shape[-3:]) == (nh, 1, 1)):
            return m.reshape(nh).contiguous()
        if m.numel() == B * nh and tuple(m.shape[-3:]) == (nh, 1, 1):
            return m.reshape(B, nh).contiguous()
        raise NotImplementedError(f"head_mask of shape {tuple(m.shape)}: only per-head (optionally per-batch-entry) factors are built")


class MultiHeadAttention(nn.Module):
    """Parameter holder, reference names (utils/TAVFormer.py:10-28): bias-free query/key/value matrices + `out` with bias."""

    def __init__(self, embed_dim=768, n_heads=12):
        super().__init__()
        self.embed_dim, self.n_heads = embed_dim, n_heads
        self.single_head_dim = int(embed_dim / n_heads)
        self.query_matrix = nn.Linear(embed_dim, embed_dim, bias=False)
        self.key_matrix = nn.Linear(embed_dim, embed_dim, bias=False)
        self.value_matrix = nn.Linear(embed_dim, embed_dim, bias=False)
        self.out = nn.Linear(embed_dim, embed_dim)


class TransformerBlock(nn.Module):
    def __init__(self, embed_dim, expansion_factor=4, n_heads=12, dropout=0.2):
        super().__init__()
        self.dropout = dropout
        self.attention = MultiHeadAttention(embed_dim, n_heads)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.feed_forward = nn.Sequential(nn.Dropout(dropout), nn.Linear(embed_dim, expansion_factor * embed_dim), nn.GELU(),
                                          nn.Linear(expansion_factor * embed_dim, embed_dim))
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(embed_dim)

    def params(self):
        a, f = self.attention, self.feed_forward
        return (a.query_matrix.weight, a.key_matrix.weight, a.value_matrix.weight, a.out.weight, a.out.bias, self.norm1.weight, self.norm1.bias,
                f[1].weight, f[1].bias, f[3].weight, f[3].bias, self.norm2.weight, self.norm2.bias)


class TransformerEncoder(nn.Module):
    """Drop-in for reference utils/TAVFormer.py:144-166 (post-LN stack, additive mask [B,1,1,S] BEFORE softmax).  The reference
    never calls .eval() on it, so its three dropouts per block stay active in training; here they follow `self.training`."""

    def __init__(self, embed_dim, num_layers=2, expansion_factor=4, n_heads=12, dropout=0.2, early_div=False):
        super().__init__()
        if embed_dim != n_heads * 64:
            raise ValueError("libtavhip attention is built for head_dim 64")
        self.early_div = early_div            # /8 before or after q.k^T: bit-identical (exact power of two), one code path
        self.n_heads, self.p = n_heads, dropout
        self.layers = nn.ModuleList([TransformerBlock(embed_dim, expansion_factor, n_heads, dropout) for _ in range(num_layers)])
        self._calls = 0

    def forward(self, x, attention_mask=None):
        if not x.is_cuda:
            raise RuntimeError("TransformerEncoder runs on libtavhip (GPU) only; there is no CPU fallback")
        ectx = runtime.ctx()
        B, S, H = x.shape
        key_mask = attention_mask.reshape(B, S).to(torch.float32).contiguous() if attention_mask is not None else None
        p = self.p if self.training else 0.0
        h = x.reshape(B * S, H)
        for i, layer in enumerate(self.layers):
            self._calls += 1
            seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._calls) & 0xFFFFFFFFFFFFFFFF
            h = E.TransformerBlockFn.apply(h, key_mask, ectx, B, S, self.n_heads, p, seed, *layer.params())
        return h.view(B, S, H)
