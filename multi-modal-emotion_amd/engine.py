"""Execution engine of the TAV path: precision policy, per-step weight-operand cache and the autograd Functions that
sequence libtavhip kernels for one transformer layer / front-end / tail.

Design (MI355X-first):
  * residual streams, LayerNorm/softmax statistics and every parameter gradient are f32; GEMM/attention operands are
    bf16 (Policy("bf16")) or f32 (Policy("fp32"), exact-f32 MFMA, used for the 1e-3 parity runs);
  * activations stay token-major [tokens, features]; q/k/v are column slices of one fused [tokens, 3H] buffer, so the
    attention kernels and the QKV projection's backward share buffers without any head-split copies;
  * each layer is ONE autograd node that launches ~12 kernels forward / ~20 backward on the current HIP stream and saves
    exactly the tensors its backward needs; nothing is allocated or synchronised inside the kernels, so a whole step can
    be captured into a hipGraph.
"""
import os
import weakref

import torch

from . import ops

_EPOCH = [0]


def bump_weight_epoch():
    """Call after parameters were updated through raw pointers (our fused AdamW) so cached operand copies refresh."""
    _EPOCH[0] += 1


class Policy:
    def __init__(self, name="bf16"):
        name = {"bfloat16": "bf16", "float32": "fp32", "f32": "fp32", "float8": "fp8", "e4m3": "fp8"}.get(name, name)
        if name not in ("bf16", "fp32", "fp8", "fp8-all", "fp8-wgrad8"):
            raise ValueError(f"unknown precision policy {name!r}")
        self.name = name
        # "fp8" (BASELINE config 5): the four linear layers of every VIDEO-encoder block (forward, dgrad, wgrad) run on e4m3 operands with
        # per-tensor scales -- at videomae-large / 32 frames that stack is 91 % of the step's FLOPs.  Everything else (attention,
        # normalisation, front-ends, residual streams, gradients, and the text / audio / fusion stacks) is the bf16 policy: the text
        # branch feeds ONE un-pooled token to the head, so e4m3 rounding there goes straight into the logits (measured 2.2e-2 with
        # every stack in fp8, above the 1e-2 budget).  "fp8-all" puts all four stacks on e4m3 anyway (throughput experiments).
        self.lp = torch.float32 if name == "fp32" else torch.bfloat16
        self.f32 = name == "fp32"
        self.fp8 = name.startswith("fp8")
        self.fp8_stacks = () if not self.fp8 else (("video", "text", "audio", "fusion") if name == "fp8-all" else ("video",))
        # Weight gradients of the fp8 stacks: bf16 (the grouped 256-wide TN kernel, bias sums fused, no transposed e4m3 copies to produce) unless
        # "fp8-wgrad8"/"fp8-all": e4m3 NT GEMMs of transposed quantised copies split over the token axis.  At config 5 / batch 16 the bf16 form wins
        # since the 256-wide weight-gradient tile exists (profiles/r02_experiments.md).
        self.fp8_wgrad8 = name in ("fp8-wgrad8", "fp8-all")
        # which of a layer's GEMMs take e4m3 operands: bit 0 qkv, 1 out-proj, 2 FFN1, 3 FFN2 -- forward (fwd) and their dgrads (bwd).
        # The others run the bf16 kernels.  (TAV_FP8_FWD_MASK / TAV_FP8_BWD_MASK: error-attribution experiments, tools/gpu_fp8_attrib.py.)
        # Default (round 4, measured at config 5's REAL depth -- 24 video layers -- over THREE seeds, profiles/r04_fp8_seeds.txt): NO forward GEMM on
        # e4m3.  Round 3 kept the QKV projection (logits 6.1e-3 on seed 0), but that was one lucky draw: the same policy gives 1.13e-2 and 1.36e-2 on
        # seeds 1 and 2 -- over the 1e-2 budget -- with the pre-scaled or the plain q alike (bf16: 5.9e-3 / 7.2e-3 / 4.2e-3); e4m3 out-proj / FFN1 /
        # FFN2 inputs cost 1.5e-2 .. 2.2e-2 each (profiles/r03_fp8_attribution.txt), also under MX block scales (profiles/r04_fp8_mx_emulation.txt).
        # All four dgrads stay on e4m3: they cannot touch the logits or the loss, and the gradient norm moves by < 2.5e-3.  TAV_FP8_FWD_MASK=1
        # restores round 3's policy; "fp8-all" / "fp8-wgrad8" remain the all-e4m3 throughput experiments.
        self.fp8_fwd = int(os.environ.get("TAV_FP8_FWD_MASK", "15" if name == "fp8-all" else "0")) if self.fp8 else 0
        self.fp8_bwd = int(os.environ.get("TAV_FP8_BWD_MASK", "15")) if self.fp8 else 0
        if self.fp8_wgrad8:
            self.fp8_fwd = self.fp8_bwd = 15             # (the e4m3 weight gradients read the transposed quantised copies of every operand)


class WeightCache:
    """Low-precision (and transposed) operand copies of f32 parameters, rebuilt only when a parameter changed."""

    def __init__(self, policy):
        self.pol = policy
        self.d = {}
        self._fp8_states = None

    @staticmethod
    def _same(refs, params):
        """Entries are keyed by id(parameter); a dead parameter's id can be handed to a new one, so every entry also holds weak
        references and is only valid while they still resolve to the very same objects."""
        live = [p for p in params if p is not None]
        return len(refs) == len(live) and all(r() is p for r, p in zip(refs, live))

    @staticmethod
    def _refs(params):
        return tuple(weakref.ref(p) for p in params if p is not None)

    def _get(self, key, params, build):
        ver = (_EPOCH[0],) + tuple(p._version for p in params if p is not None)
        e = self.d.get(key)
        if e is not None and e[0] == ver and self._same(e[2], params):
            return e[1]
        with torch.no_grad():
            val = build()
        self.d[key] = (ver, val, self._refs(params))
        return val

    def linear(self, w):
        """-> (W [N,K], W^T [K,N]) in the operand dtype."""
        def build():
            if self.pol.f32:
                return w.detach(), ops.cast_weight(w.detach(), torch.float32, want_n=False)[1]
            return ops.cast_weight(w.detach(), self.pol.lp)
        return self._get(("lin", id(w)), (w,), build)

    def qkv(self, wq, wk, wv, bq, bk, bv):
        """-> (Wqkv [3H,K], Wqkv^T [K,3H], bias [3H] f32) fused operands (biases that are None read as zero)."""
        def build():
            H, K = wq.shape
            dev = wq.device
            n = torch.empty(3 * H, K, dtype=self.pol.lp, device=dev)
            t = torch.empty(K, 3 * H, dtype=self.pol.lp, device=dev)
            bias = ops.zeros_f32((3 * H,), dev)
            for j, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
                ops.cast_weight(w.detach(), self.pol.lp, out_n=n[j * H:(j + 1) * H], out_t=t[:, j * H:(j + 1) * H])
                if b is not None:
                    ops.cast2d(b.detach().view(1, H), torch.float32, out=bias[j * H:(j + 1) * H].view(1, H))
            return n, t, bias
        return self._get(("qkv", id(wq)), (wq, wk, wv, bq, bk, bv), build)

    def layer(self, wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=None):
        """All GEMM operands of one transformer layer, refreshed by ONE multi-tensor launch when any of them changed:
        -> (Wqkv [3H,K], Wqkv^T [K,3H], bias_qkv [3H] f32, Wo, Wo^T, W1, W1^T, W2, W2^T).  Buffers and the device descriptor
        table are created once (parameter storage is stable), so the refresh is graph-capturable and allocation free.
        q_scale: factor folded into the q rows of the FORWARD operand Wqkv and of bias_qkv (ops.ATTN_Q_PRESCALE: the attention kernels
        then get q * scale * log2(e) straight from the projection, rounded once).  Wqkv^T -- the dgrad operand -- stays unscaled: the
        attention backward returns the gradient w.r.t. the unscaled q."""
        key = ("layer", id(wq), q_scale)
        params = (wq, wk, wv, bq, bk, bv, wo, w1, w2)
        ver = (_EPOCH[0],) + tuple(p._version for p in params if p is not None)
        e = self.d.get(key)
        if e is not None and not self._same(e[3], params):
            e = None                                              # another (dead) layer's entry under a recycled id: rebuild everything
        if e is not None and e[0] == ver:
            return e[1]
        with torch.no_grad():
            if e is None:
                H, K = wq.shape
                dev, lp = wq.device, self.pol.lp
                n = torch.empty(3 * H, K, dtype=lp, device=dev)
                t = torch.empty(K, 3 * H, dtype=lp, device=dev)
                bias = ops.zeros_f32((3 * H,), dev)
                ents = []
                for j, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
                    sc = q_scale if j == 0 else None
                    ents.append((w.detach(), n[j * H:(j + 1) * H], t[:, j * H:(j + 1) * H], sc))
                    if b is not None:
                        ents.append((b.detach().view(1, H), bias[j * H:(j + 1) * H].view(1, H), None, sc))
                outs = [n, t, bias]
                for w in (wo, w1, w2):
                    wn_ = torch.empty(w.shape, dtype=lp, device=dev)
                    wt_ = torch.empty(w.shape[1], w.shape[0], dtype=lp, device=dev)
                    ents.append((w.detach(), wn_, wt_))
                    outs += [wn_, wt_]
                descs, tiles = ops.make_cast_descs(ents, dev)
                val, aux = tuple(outs), (descs, len(ents), tiles)
            else:
                val, aux = e[1], e[2]
            ops.cast_weights_multi(aux[0], aux[1], aux[2])       # one workgroup per 32x32 tile of the largest tensor
        self.d[key] = (ver, val, aux, self._refs(params))
        return val

    def fp8_state(self, param, tag):
        """(Fp8States, key) of one quantisation site: the tensor `tag` produced next to `param` (a layer's wq identifies the layer)."""
        if self._fp8_states is None:
            self._fp8_states = ops.Fp8States(param.device)
        return self._fp8_states, (id(param), tag)

    def layer_fp8(self, wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=None):
        """e4m3 operands of one transformer layer (per-tensor scales): (Wqkv forward, Wqkv dgrad, Wo, W1, W2) as ops.Fp8.  The forward Wqkv is
        the [3H, K] copy of the q-SCALED bf16 operand (q_scale as in layer(): the attention kernels get q * scale * log2(e) from the
        projection), the dgrad Wqkv the transposed [K, 3H_pad] copy of the UNSCALED one (the attention backward returns the gradient
        w.r.t. the unscaled q); Wo / W1 / W2 carry both copies.  Weights move little from step to step: delayed scaling (ops.fp8_quantize)."""
        def build():
            wqkv, wqkv_t, _, wo_n, _, w1_n, _, w2_n, _ = self.layer(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=q_scale)
            f = ops.fp8_quantize(wqkv, want_t=False, state=self.fp8_state(wq, "Wqkv.f")) if (self.pol.fp8_fwd & 1) else None
            # the dgrad operand is the transposed copy [K, 3H] of the UNSCALED weight: exactly the row-major e4m3 copy of layer()'s Wqkv^T
            bq8 = ops.fp8_quantize(wqkv_t, want_t=False, state=self.fp8_state(wq, "Wqkv.b"))
            b = ops.Fp8(None, bq8.q, bq8.scales, wqkv_t.shape[1], wqkv_t.shape[0])
            rest = tuple(ops.fp8_quantize(w, want_t=True, state=self.fp8_state(wq, t)) for w, t in ((wo_n, "Wo"), (w1_n, "W1"), (w2_n, "W2")))
            return (f, b) + rest
        return self._get(("layer_fp8", id(wq), q_scale), (wq, wk, wv, wo, w1, w2), build)

    def conv(self, w, stride=0):
        """nn.Conv1d weight [co,ci,k] -> (operand [co, k*ci], column-buffer dgrad operand [k*ci, co], per-phase dgrad operands or None)."""
        return self._get(("conv", id(w), int(stride)), (w,), lambda: ops.cast_conv_weight(w.detach(), self.pol.lp, stride))

    def posconv(self, v, g):
        """weight-normed grouped conv -> (w [G,Cg,K*Cg], flipped dgrad form, norms [K])."""
        return self._get(("posconv", id(v)), (v, g), lambda: ops.weight_norm_fwd(v.detach(), g.detach(), self.pol.lp))


class Ctx:
    """What every Function needs besides tensors."""

    def __init__(self, policy, cache=None):
        self.pol = policy if isinstance(policy, Policy) else Policy(policy)
        self.cache = cache or WeightCache(self.pol)


# ---------------------------------------------------------------------------------------------- helpers
def _ln_fwd(pol, x, w, b, eps, need_f32, act=0):
    if pol.f32:
        y32, _, mean, rstd = ops.ln_fwd(x, w, b, eps, want_f32=True, lp_dtype=None, act=act)
        return y32, y32, mean, rstd
    y32, ylp, mean, rstd = ops.ln_fwd(x, w, b, eps, want_f32=need_f32, lp_dtype=pol.lp, act=act)
    return y32, ylp, mean, rstd


def _ln_bwd(pol, dy, x, w, b, mean, rstd, dx_add=None, need_lp=True, act=0, need_f32=True):
    if pol.f32:
        dx32, _, dg, db = ops.ln_bwd(dy, x, w, b, mean, rstd, dx_add=dx_add, want_f32=True, lp_dtype=None, act=act)
        return dx32, dx32, dg, db
    return ops.ln_bwd(dy, x, w, b, mean, rstd, dx_add=dx_add, want_f32=need_f32, lp_dtype=pol.lp if need_lp else None, act=act)


def _to_lp(pol, x32):
    return x32 if pol.f32 else ops.cast2d(x32, pol.lp)


# A pre-LN layer's backward starts by casting the incoming f32 gradient to the operand dtype.  When that gradient is the very tensor
# the next layer's backward just produced (LN backward can emit the low-precision copy for free), the copy travels beside it:
# one entry per stream, matched by tensor identity (a summed / re-materialised gradient never matches and is cast as before).
_LP_HINT = {}


_LP_HINT_ON = os.environ.get("TAV_LP_HINT", "1") == "1"


def _hint_set(g32, glp):
    if _LP_HINT_ON and glp is not None and g32 is not None and g32.is_cuda:
        _LP_HINT[torch.cuda.current_stream().cuda_stream] = (g32, glp)


def _hint_take(g):
    if not g.is_cuda:
        return None
    e = _LP_HINT.pop(torch.cuda.current_stream().cuda_stream, None)
    return e[1] if e is not None and e[0] is g else None


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


_BWD_LP_OUT = os.environ.get("TAV_BWD_LP_OUT", "1") == "1"


def _DGRAD_OUT(pol):
    return torch.float32 if (pol.f32 or not _BWD_LP_OUT) else pol.lp


# ---------------------------------------------------------------------------------------------- encoder layer
class LayerSpec:
    def __init__(self, B, S, nheads, eps, pre_ln, mask_mode=0, branch=None, head_scale=None, probs_out=None):
        self.B, self.S, self.nheads, self.eps, self.pre_ln, self.mask_mode, self.branch = B, S, nheads, eps, pre_ln, mask_mode, branch
        # slow-path options of the fusion encoder (reference utils/TAVFormer.py:190, :368-370, :389; never set by the training loop):
        # head_scale: f32 [nheads] / [B, nheads] factor on the softmax part of the context (head_mask); probs_out: list that receives the
        # layer's attention probabilities [B, nheads, S, S] f32 (output_attentions; detached)
        self.head_scale, self.probs_out = head_scale, probs_out


GROUPED_WGRAD = [os.environ.get("TAV_GROUPED_WGRAD", "1") == "1"]
# attention pre-scale of the q rows (see WeightCache.layer); TAV_ATTN_PRESCALE=0 keeps q unscaled (the round-2 kernels' convention, A/B)
QSC = ops.ATTN_Q_PRESCALE if os.environ.get("TAV_ATTN_PRESCALE", "1") == "1" else None


def _layer_wgrads(pairs, into=None):
    """Weight and bias gradients of one layer's linears, [(dY_lp, X_lp), ...] -> [(dW, db), ...].  One grouped launch in which
    every tile sums over all tokens (no split slabs, no reduce kernels) when the layer's tiles fill the chip; else one TN GEMM each.
    into: destinations inside a data-parallel bucket (see _arena_into), honoured by the grouped launch."""
    tiles = sum(((a.shape[1] + 127) // 128) * ((b.shape[1] + 127) // 128) for a, b in pairs)
    if GROUPED_WGRAD[0] and len(pairs) <= 4 and tiles >= 256:
        return ops.gemm_tn_grouped(pairs, want_bias=True, into=into)
    return [ops.gemm_tn(a, b, want_bias=True) for a, b in pairs]


def _arena_into(wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2):
    """Gradient-arena destinations of one layer's grouped weight-gradient launch, in its pair order (W2, W1, Wo, Wqkv), or None when no
    data-parallel capture has registered slots (runtime.grad_slots).  Wq | Wk | Wv need adjacent slots: their gradient is one fused tensor."""
    from . import runtime as _rt
    if not _rt.grad_slots:
        return None
    # Every slot is handed out ONCE (popped): a layer differentiated a second time inside the same segment (shared / re-applied layer) gets no
    # slot, its launch writes a fresh buffer and autograd ADDS that to the gradient already sitting in the arena view -- a second launch into
    # the same slot would overwrite the first gradient and then be added to itself (2 x the second gradient, silently).
    g = lambda p: _rt.grad_slots.pop(p.data_ptr(), None) if p is not None else None       # noqa: E731

    def fused(ps):
        if any(p is None for p in ps):
            return None
        v = _rt.fused_slot(ps)
        for p in ps:                                 # (also when they are not adjacent: a partly used triple must not be half-taken later)
            _rt.grad_slots.pop(p.data_ptr(), None)
        return v
    return [(g(w2), g(b2)), (g(w1), g(b1)), (g(wo), g(bo)), (fused((wq, wk, wv)), fused((bq, bk, bv)))]


class EncoderLayerFn(torch.autograd.Function):
    """One transformer layer.  params = (ln1_w, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2).

    pre_ln  (VideoMAE / fusion / wav2vec2 stable-LN; reference utils/TAVFormer.py:243-271, HF videomae:326-357, wav2vec2:611-654):
        x1 = x + Wo.attn(LN1(x));  x2 = x1 + W2.gelu(W1.LN2(x1))
    post_ln (BERT / RoBERTa / wav2vec2-base; HF roberta:211-398, wav2vec2:575-608):
        x1 = LN1(x + Wo.attn(x));  x2 = LN2(x1 + W2.gelu(W1.x1))
    Returns (x2 f32, x2_lp): the low-precision copy is a by-product for the next layer's GEMMs (not differentiable).
    """

    @staticmethod
    def forward(ctx, x, x_lp, key_mask, ectx, spec, *params):
        pol, cache = ectx.pol, ectx.cache
        (ln1_w, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2) = params
        B, S, nh = spec.B, spec.S, spec.nheads
        H = nh * 64
        from . import runtime as _rt
        _rt.note_layer_group(wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2)          # (data-parallel gradient arena: which gradients one grouped launch produces)
        # the q third of qkv comes out of the projection already multiplied by scale * log2(e) (folded into the weight copy)
        wqkv, _, bqkv, wo_n, _, w1_n, _, w2_n, _ = cache.layer(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=QSC)
        x = _c(x)
        if spec.pre_ln:
            _, a, mean1, rstd1 = _ln_fwd(pol, x, ln1_w, ln1_b, spec.eps, need_f32=False)
        else:
            a = x_lp if x_lp is not None else _to_lp(pol, x)
            mean1 = rstd1 = None
        qkv = ops.gemm_nt(a, wqkv, bias=bqkv)
        o, lse, aux = ops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nh, key_mask=key_mask, mask_mode=spec.mask_mode, q_prescaled=QSC is not None)
        hs, o_ctx = spec.head_scale, o
        if hs is not None:                                      # head_mask: context = m_h * softmax(s) v (+ the rank-1 mask term, unscaled)
            if spec.mask_mode == 2:
                o_ctx = ops.head_scale(o, aux[1], hs, -1.0, B, S, nh)
            elif spec.mask_mode == 0:
                o_ctx = ops.head_scale(None, o, hs, 0.0, B, S, nh)
            else:
                raise NotImplementedError("head_mask with a pre-softmax key mask is not on any path of the reference")
        if spec.probs_out is not None:
            spec.probs_out.append(ops.attn_probs(qkv[:, :H], qkv[:, H:2 * H], lse, B, S, nh, key_mask=key_mask, mask_mode=spec.mask_mode,
                                                 q_prescaled=QSC is not None, head_scale=hs))
        y1 = ops.gemm_nt(o_ctx, wo_n, bias=bo, resid=x, out_dtype=torch.float32)
        if spec.pre_ln:
            x1 = y1
            _, c, mean2, rstd2 = _ln_fwd(pol, x1, ln2_w, ln2_b, spec.eps, need_f32=False)
        else:
            x1, c, mean1, rstd1 = _ln_fwd(pol, y1, ln1_w, ln1_b, spec.eps, need_f32=True)
        h, u = ops.gemm_nt(c, w1_n, bias=b1, act=3, want_pre=True)          # u = gelu'(W1 c + b1): what the backward multiplies by
        y2 = ops.gemm_nt(h, w2_n, bias=b2, resid=x1, out_dtype=torch.float32)
        if spec.pre_ln:
            x2, x2_lp = y2, None
            mean_o = rstd_o = None
        else:
            x2, x2_lp, mean_o, rstd_o = _ln_fwd(pol, y2, ln2_w, ln2_b, spec.eps, need_f32=True)
            mean2, rstd2 = mean_o, rstd_o
        ctx.ectx, ctx.spec = ectx, spec
        ctx.set_materialize_grads(False)          # no zero-filled gradient tensor for the non-differentiable lp by-product
        ctx.has = [p is not None for p in params]
        corr, o_soft = aux
        ctx.save_for_backward(x if spec.pre_ln else None, a, qkv, o, lse, corr, o_soft, y1, c, u, h, y2 if not spec.pre_ln else None,
                              mean1, rstd1, mean2, rstd2, key_mask, o_ctx if hs is not None else None, hs, *params)
        if x2_lp is None or pol.f32:
            x2_lp = x2.new_empty(0)
        ctx.mark_non_differentiable(x2_lp)
        return x2, x2_lp

    @staticmethod
    def backward(ctx, g2, _g_lp):
        pol, cache, spec = ctx.ectx.pol, ctx.ectx.cache, ctx.spec
        sv = ctx.saved_tensors
        x, a, qkv, o, lse, corr, o_soft, y1, c, u, h, y2, mean1, rstd1, mean2, rstd2, key_mask, o_ctx, hs = sv[:19]
        (ln1_w, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2) = sv[19:]
        B, S, nh = spec.B, spec.S, spec.nheads
        H = nh * 64
        _, wqkv_t, _, _, wo_t, _, w1_t, _, w2_t = cache.layer(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=QSC)
        g2 = _c(g2)
        if spec.pre_ln:
            hint = None if pol.f32 else _hint_take(g2)
            dy2, dy2_lp = g2, (hint if hint is not None else _to_lp(pol, g2))
            dg2 = db2 = None
        else:
            dy2, dy2_lp, dg2, db2 = _ln_bwd(pol, g2, y2, ln2_w, ln2_b, mean2, rstd2)
        # FFN
        du = ops.gemm_nt(dy2_lp, w2_t, gelu_in=u, act=4)
        if spec.pre_ln:
            # dgrad outputs that ONLY feed a LayerNorm backward (pre-LN layers) leave the GEMM in the operand dtype: LN backward takes the
            # rounded values straight into its f32 row arithmetic and adds the f32 residual gradient there, so the residual stream itself
            # stays f32 -- and the GEMM epilogue writes, and LN backward reads, half the bytes (TAV_BWD_LP_OUT=0: f32 as in round 2)
            dc = ops.gemm_nt(du, w1_t, out_dtype=_DGRAD_OUT(pol))
            g1, g1_lp, dg2, db2 = _ln_bwd(pol, dc, y1, ln2_w, ln2_b, mean2, rstd2, dx_add=dy2)
            dy1, dy1_lp = g1, g1_lp
        else:
            g1 = ops.gemm_nt(du, w1_t, resid=dy2, out_dtype=torch.float32)
            dy1, dy1_lp, dg1, db1 = _ln_bwd(pol, g1, y1, ln1_w, ln1_b, mean1, rstd1)
        # attention
        do = ops.gemm_nt(dy1_lp, wo_t)
        qs, ks, vs, pre = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], QSC is not None
        if hs is None:
            dqkv = ops.attn_bwd(qs, ks, vs, o, do, lse, (corr, o_soft) if spec.mask_mode == 2 else None,
                                B, S, nh, key_mask=key_mask, mask_mode=spec.mask_mode, q_prescaled=pre)
        elif spec.mask_mode == 2:
            # context = o + (m_h - 1) o_soft: the whole of dO goes through the usual backward, (m_h - 1) dO once more through the softmax part alone
            dqkv = ops.attn_bwd(qs, ks, vs, o, do, lse, (corr, o_soft), B, S, nh, key_mask=key_mask, mask_mode=2, q_prescaled=pre)
            dqkv2 = ops.attn_bwd(qs, ks, vs, o_soft, ops.head_scale(None, do, hs, -1.0, B, S, nh), lse, None, B, S, nh, mask_mode=0, q_prescaled=pre)
            ops.head_scale(dqkv, dqkv2, None, 1.0, B, S, 3 * nh, out=dqkv)
        else:                                                   # context = m_h o
            dqkv = ops.attn_bwd(qs, ks, vs, o, ops.head_scale(None, do, hs, 0.0, B, S, nh), lse, None, B, S, nh, mask_mode=0, q_prescaled=pre)
        if hs is not None:
            o = o_ctx                                           # what the out-projection multiplied (its weight gradient below)
        if spec.pre_ln:
            da = ops.gemm_nt(dqkv, wqkv_t, out_dtype=_DGRAD_OUT(pol))
            g0, g0_lp, dg1, db1 = _ln_bwd(pol, da, x, ln1_w, ln1_b, mean1, rstd1, dx_add=dy1, need_lp=_LP_HINT_ON and not pol.f32)
            if not pol.f32:
                _hint_set(g0, g0_lp)
        else:
            g0 = ops.gemm_nt(dqkv, wqkv_t, resid=dy1, out_dtype=torch.float32)
        # the four weight (+bias) gradients of the layer: leaves nobody reads before the optimizer, issued last as ONE grouped launch
        (dW2, dB2), (dW1, dB1), (dWo, dBo), (dWqkv, dBqkv) = _layer_wgrads([(dy2_lp, h), (du, c), (dy1_lp, o), (dqkv, a)],
                                                                           _arena_into(wq, wk, wv, bq, bk, bv, wo, bo, w1, b1, w2, b2))
        grads = [dg1, db1, dWqkv[:H], dBqkv[:H], dWqkv[H:2 * H], dBqkv[H:2 * H], dWqkv[2 * H:], dBqkv[2 * H:], dWo, dBo, dg2, db2, dW1, dB1, dW2, dB2]
        grads = [g if has else None for g, has in zip(grads, ctx.has)]
        return (g0, None, None, None, None, *grads)


# Experiment hook (tools/gpu_fp8_mx_probe.py; never set in the product): TAV_FP8_MX_EMULATE=1 runs the forward GEMMs selected by Policy.fp8_fwd bits
# 1-3 (out-proj / FFN1 / FFN2) on operands rounded to MX-e4m3 -- e4m3 elements with a power-of-two scale per 32 k-values, the format the
# block-scaled MFMA takes natively -- by quantise / dequantise in torch, to measure what block scales would buy BEFORE building that path.
_MX_EMULATE = os.environ.get("TAV_FP8_MX_EMULATE", "0") == "1"


def _mx_roundtrip(x, block=32):
    M, K = x.shape
    xb = x.float().reshape(M, K // block, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    sc = torch.exp2(torch.floor(torch.log2(448.0 / amax)))           # E8M0: the largest power of two that keeps the block inside +-448
    return ((xb * sc).to(torch.float8_e4m3fn).float() / sc).reshape(M, K).to(x.dtype)


class EncoderLayerFp8Fn(torch.autograd.Function):
    """EncoderLayerFn with the four linear layers on e4m3 operands (Policy("fp8"), BASELINE config 5).  Each GEMM input (LayerNorm output,
    attention output, GELU output, and in the backward the four incoming gradients) is quantised once per use with its own per-tensor
    scale (ops.fp8_quantize: amax -> scale on the device -> e4m3 copy + transposed copy); forward and dgrad read the row-major copies, the
    weight gradients are NT GEMMs of the transposed copies (ops.wgrad_fp8).  Attention, LayerNorm, residual adds and all statistics are
    exactly those of the bf16 policy.  What is saved for the backward are the fp8 transposed activations, not their bf16 originals."""

    @staticmethod
    def forward(ctx, x, x_lp, key_mask, ectx, spec, *params):
        pol, cache = ectx.pol, ectx.cache
        (ln1_w, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2) = params
        B, S, nh = spec.B, spec.S, spec.nheads
        H = nh * 64
        wqkv, _, bqkv, wo_n, _, w1_n, _, w2_n, _ = cache.layer(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=QSC)
        wqkv8, _, wo8, w18, w28 = cache.layer_fp8(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=QSC)
        fm = pol.fp8_fwd
        a8 = o8 = c8 = h8 = None
        st = lambda tag: cache.fp8_state(wq, tag)           # noqa: E731  (one delayed-scaling state per quantisation site of this layer)
        x = _c(x)
        if spec.pre_ln:
            _, a, mean1, rstd1 = _ln_fwd(pol, x, ln1_w, ln1_b, spec.eps, need_f32=False)
        else:
            a = x_lp if x_lp is not None else _to_lp(pol, x)
            mean1 = rstd1 = None
        if fm & 1:
            a8 = ops.fp8_quantize(a, want_t=pol.fp8_wgrad8, state=st("a"))
            qkv = ops.gemm_nt_fp8(a8, wqkv8, bias=bqkv)
        else:
            qkv = ops.gemm_nt(a, wqkv, bias=bqkv)
        o, lse, aux = ops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nh, key_mask=key_mask, mask_mode=spec.mask_mode, q_prescaled=QSC is not None)
        if (fm & 2) and _MX_EMULATE:
            y1 = ops.gemm_nt(_mx_roundtrip(o), _mx_roundtrip(wo_n), bias=bo, resid=x, out_dtype=torch.float32)
        elif fm & 2:
            o8 = ops.fp8_quantize(o, want_t=pol.fp8_wgrad8, state=st("o"))
            y1 = ops.gemm_nt_fp8(o8, wo8, bias=bo, resid=x, out_dtype=torch.float32)
        else:
            y1 = ops.gemm_nt(o, wo_n, bias=bo, resid=x, out_dtype=torch.float32)
        if spec.pre_ln:
            x1 = y1
            _, c, mean2, rstd2 = _ln_fwd(pol, x1, ln2_w, ln2_b, spec.eps, need_f32=False)
        else:
            x1, c, mean1, rstd1 = _ln_fwd(pol, y1, ln1_w, ln1_b, spec.eps, need_f32=True)
        if (fm & 4) and _MX_EMULATE:
            h, u = ops.gemm_nt(_mx_roundtrip(c), _mx_roundtrip(w1_n), bias=b1, act=3, want_pre=True)
        elif fm & 4:
            c8 = ops.fp8_quantize(c, want_t=pol.fp8_wgrad8, state=st("c"))
            h, u = ops.gemm_nt_fp8(c8, w18, bias=b1, act=3, want_pre=True)
        else:
            h, u = ops.gemm_nt(c, w1_n, bias=b1, act=3, want_pre=True)
        if (fm & 8) and _MX_EMULATE:
            y2 = ops.gemm_nt(_mx_roundtrip(h), _mx_roundtrip(w2_n), bias=b2, resid=x1, out_dtype=torch.float32)
        elif fm & 8:
            h8 = ops.fp8_quantize(h, want_t=pol.fp8_wgrad8, state=st("h"))
            y2 = ops.gemm_nt_fp8(h8, w28, bias=b2, resid=x1, out_dtype=torch.float32)
        else:
            y2 = ops.gemm_nt(h, w2_n, bias=b2, resid=x1, out_dtype=torch.float32)
        if spec.pre_ln:
            x2, x2_lp = y2, None
        else:
            x2, x2_lp, mean2, rstd2 = _ln_fwd(pol, y2, ln2_w, ln2_b, spec.eps, need_f32=True)
        ctx.ectx, ctx.spec = ectx, spec
        ctx.set_materialize_grads(False)
        ctx.has = [p is not None for p in params]
        ctx.rows = a.shape[0]
        corr, o_soft = aux
        if pol.fp8_wgrad8:
            keep = (a8.qt, a8.scales, o8.qt, o8.scales, c8.qt, c8.scales, h8.qt, h8.scales)
        else:                                    # the bf16 operands of the weight-gradient GEMMs (o is saved for attention anyway)
            keep = (a, None, None, None, c, None, h, None)
        ctx.save_for_backward(x if spec.pre_ln else None, qkv, o, lse, corr, o_soft, y1, u, y2 if not spec.pre_ln else None, mean1, rstd1, mean2, rstd2, key_mask,
                              *keep, *params)
        if x2_lp is None:
            x2_lp = x2.new_empty(0)
        ctx.mark_non_differentiable(x2_lp)
        return x2, x2_lp

    @staticmethod
    def backward(ctx, g2, _g_lp):
        pol, cache, spec = ctx.ectx.pol, ctx.ectx.cache, ctx.spec
        sv = ctx.saved_tensors
        x, qkv, o, lse, corr, o_soft, y1, u, y2, mean1, rstd1, mean2, rstd2, key_mask = sv[:14]
        a_t, a_s, o_t, o_s, c_t, c_s, h_t, h_s = sv[14:22]
        (ln1_w, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2) = sv[22:]
        B, S, nh = spec.B, spec.S, spec.nheads
        H, F, M = nh * 64, w1.shape[0], ctx.rows
        _, wqkv8, wo8, w18, w28 = cache.layer_fp8(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=QSC)
        _, wqkv_t, _, _, wo_t, _, w1_t, _, w2_t = cache.layer(wq, wk, wv, bq, bk, bv, wo, w1, w2, q_scale=QSC)
        w8 = pol.fp8_wgrad8
        bm = pol.fp8_bwd
        pre = QSC is not None
        if w8:
            a8, o8, c8, h8 = (ops.Fp8(None, t, s, M, t.shape[0]) for t, s in ((a_t, a_s), (o_t, o_s), (c_t, c_s), (h_t, h_s)))

        def dgrad(dy_lp, bit, wq8, w_t, **kw):   # dY [M, N] x W [N, K] -> [M, K]: the NT GEMM against the transposed copy W^T [K, N(_pad)]
            if bm & bit:
                dy8 = ops.fp8_quantize(dy_lp, want_t=w8, state=cache.fp8_state(wq, f"dY{bit}"))
                return ops.gemm_nt(dy8.q, wq8.qt[:, :wq8.rows], a_dequant=dy8.dequant, b_dequant=wq8.dequant, **kw), dy8
            return ops.gemm_nt(dy_lp, w_t, **kw), None

        g2 = _c(g2)
        if spec.pre_ln:
            hint = _hint_take(g2)
            dy2, dy2_lp = g2, (hint if hint is not None else _to_lp(pol, g2))
            dg2 = db2 = None
        else:
            dy2, dy2_lp, dg2, db2 = _ln_bwd(pol, g2, y2, ln2_w, ln2_b, mean2, rstd2)
        du, dy28 = dgrad(dy2_lp, 8, w28, w2_t, gelu_in=u, act=4)
        if spec.pre_ln:
            dc, du8 = dgrad(du, 4, w18, w1_t, out_dtype=torch.float32)
            g1, g1_lp, dg2, db2 = _ln_bwd(pol, dc, y1, ln2_w, ln2_b, mean2, rstd2, dx_add=dy2)
            dy1, dy1_lp = g1, g1_lp
        else:
            g1, du8 = dgrad(du, 4, w18, w1_t, resid=dy2, out_dtype=torch.float32)
            dy1, dy1_lp, dg1, db1 = _ln_bwd(pol, g1, y1, ln1_w, ln1_b, mean1, rstd1)
        do, dy18 = dgrad(dy1_lp, 2, wo8, wo_t)
        dqkv = ops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, do, lse, (corr, o_soft) if spec.mask_mode == 2 else None,
                            B, S, nh, key_mask=key_mask, mask_mode=spec.mask_mode, q_prescaled=pre)
        if spec.pre_ln:
            da, dqkv8 = dgrad(dqkv, 1, wqkv8, wqkv_t, out_dtype=torch.float32)
            g0, g0_lp, dg1, db1 = _ln_bwd(pol, da, x, ln1_w, ln1_b, mean1, rstd1, dx_add=dy1, need_lp=_LP_HINT_ON)
            _hint_set(g0, g0_lp)
        else:
            g0, dqkv8 = dgrad(dqkv, 1, wqkv8, wqkv_t, resid=dy1, out_dtype=torch.float32)
        if w8:
            dW2, dW1, dWo, dWqkv = ops.wgrad_fp8(dy28, h8), ops.wgrad_fp8(du8, c8), ops.wgrad_fp8(dy18, o8), ops.wgrad_fp8(dqkv8, a8)
            dB2, dB1, dBo, dBqkv = ops.colsum(dy2_lp), ops.colsum(du), ops.colsum(dy1_lp), ops.colsum(dqkv)
        else:                                    # one grouped bf16 launch, bias sums fused (a_t / c_t / h_t hold the bf16 activations here)
            (dWqkv, dBqkv), (dWo, dBo), (dW1, dB1), (dW2, dB2) = ops.gemm_tn_grouped([(dqkv, a_t), (dy1_lp, o), (du, c_t), (dy2_lp, h_t)], want_bias=True)
        grads = [dg1, db1, dWqkv[:H], dBqkv[:H], dWqkv[H:2 * H], dBqkv[H:2 * H], dWqkv[2 * H:], dBqkv[2 * H:], dWo, dBo, dg2, db2, dW1, dB1, dW2, dB2]
        grads = [g if has else None for g, has in zip(grads, ctx.has)]
        return (g0, None, None, None, None, *grads)


_F32_BRANCHES = tuple(b for b in os.environ.get("TAV_F32_BRANCHES", "").split(",") if b)      # error attribution only (tools/gpu_bf16_attrib.py)
_f32_ctx = []


def encoder_layer(ectx, spec, x, x_lp, key_mask, params):
    if _F32_BRANCHES and spec.branch in _F32_BRANCHES and not ectx.pol.f32:
        # attribution experiment: this stack's transformer layers run under the fp32 policy inside an otherwise bf16 step
        if not _f32_ctx:
            _f32_ctx.append(Ctx("fp32"))
        x2, _ = EncoderLayerFn.apply(x, None, key_mask, _f32_ctx[0], spec, *params)
        return x2, None
    fn = EncoderLayerFp8Fn if (ectx.pol.fp8 and spec.branch in ectx.pol.fp8_stacks) else EncoderLayerFn
    if fn is EncoderLayerFp8Fn and (spec.head_scale is not None or spec.probs_out is not None):
        raise NotImplementedError("head_mask / output_attentions are built for the bf16 and fp32 policies only")
    x2, x2_lp = fn.apply(x, x_lp, key_mask, ectx, spec, *params)
    return x2, (x2_lp if x2_lp.numel() else None)


# ---------------------------------------------------------------------------------------------- generic pieces
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) (+ resid) with x given as its operand-dtype copy; returns f32 or lp output.
    Used for projections outside the layer body (feature projection, wav_2_768*, pooler, patch embedding)."""

    @staticmethod
    def forward(ctx, x32, x_lp, w, b, resid, ectx, out_f32):
        pol = ectx.pol
        if x_lp is None:
            x_lp = _to_lp(pol, _c(x32))
        w_n, _ = ectx.cache.linear(w)
        y = ops.gemm_nt(x_lp, w_n, bias=b, resid=resid, out_dtype=torch.float32 if out_f32 else pol.lp)
        ctx.ectx = ectx
        ctx.in_f32 = x32 is not None
        ctx.save_for_backward(x_lp, w, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        pol = ctx.ectx.pol
        x_lp, w, b = ctx.saved_tensors
        _, w_t = ctx.ectx.cache.linear(w)
        gy = _c(gy)
        gy_lp = gy if gy.dtype == pol.lp else _to_lp(pol, gy)
        dW = dB = None
        if ctx.needs_input_grad[2]:
            if b is not None and ctx.needs_input_grad[3]:
                dW, dB = ops.gemm_tn(gy_lp, x_lp, want_bias=True)
            else:
                dW = ops.gemm_tn(gy_lp, x_lp)
        elif b is not None and ctx.needs_input_grad[3]:
            dB = ops.colsum(gy_lp)
        dx = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = ops.gemm_nt(gy_lp, w_t, out_dtype=torch.float32 if ctx.in_f32 else pol.lp)
        d_res = gy if ctx.needs_input_grad[4] else None
        return (dx if ctx.in_f32 else None, None if ctx.in_f32 else dx, dW, dB, d_res, None, None)


class LayerNormF32Fn(torch.autograd.Function):
    """f32 residual stream [rows, W] -> (y f32, y_lp by-product for the next GEMM, not differentiable)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, ectx):
        x = _c(x)
        y32, ylp, mean, rstd = _ln_fwd(ectx.pol, x, w, b, eps, need_f32=True)
        ctx.ectx = ectx
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, w, b, mean, rstd)
        if ectx.pol.f32:
            ylp = y32.new_empty(0)
        ctx.mark_non_differentiable(ylp)
        return y32, ylp

    @staticmethod
    def backward(ctx, g32, _glp):
        x, w, b, mean, rstd = ctx.saved_tensors
        dx32, _, dg, db = _ln_bwd(ctx.ectx.pol, _c(g32), x, w, b, mean, rstd, need_lp=False)
        return dx32, dg, db, None, None


class LayerNormLpFn(torch.autograd.Function):
    """operand-dtype activation [rows, W] -> operand-dtype output; act=1 fuses the exact GELU
    (wav2vec2 'layer' conv blocks, HF wav2vec2:275-301; feature projection LN :422-434)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, ectx, act):
        x = _c(x)
        _, ylp, mean, rstd = _ln_fwd(ectx.pol, x, w, b, eps, need_f32=False, act=act)
        ctx.ectx, ctx.act = ectx, act
        ctx.save_for_backward(x, w, b, mean, rstd)
        return ylp

    @staticmethod
    def backward(ctx, g):
        x, w, b, mean, rstd = ctx.saved_tensors
        pol = ctx.ectx.pol
        _, dxlp, dg, db = _ln_bwd(pol, _c(g), x, w, b, mean, rstd, need_lp=True, act=ctx.act, need_f32=pol.f32)
        return dxlp, dg, db, None, None, None


def layer_norm_f32(ectx, x, w, b, eps):
    """Returns (y_f32, y_lp); in the fp32 policy both are the same tensor."""
    y32, ylp = LayerNormF32Fn.apply(x, w, b, eps, ectx)
    return y32, (y32 if ectx.pol.f32 else ylp)


def layer_norm_lp(ectx, x, w, b, eps, act=0):
    return LayerNormLpFn.apply(x, w, b, eps, ectx, act)


# ---------------------------------------------------------------------------------------------- embeddings
class EmbedAddFn(torch.autograd.Function):
    """out = x + table[ids]   (models/tav.py:474: hidden_states + nn.Embedding(3,768)(pos_embed))."""

    @staticmethod
    def forward(ctx, x, ids, table):
        x = _c(x)
        ctx.save_for_backward(ids)
        ctx.ntable = table.shape[0]
        return ops.embed_add_fwd(x, ids, table.detach())

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        g = _c(g)
        dt = ops.embed_add_bwd(g, ids, ctx.ntable) if ctx.needs_input_grad[2] else None
        return g, None, dt


class TextEmbedFn(torch.autograd.Function):
    """HF roberta/bert embeddings (roberta:75-121): word + position + token_type[0] -> LayerNorm. Returns (y f32, y_lp)."""

    @staticmethod
    def forward(ctx, ids, word, pos, typ, ln_w, ln_b, eps, pad_id, ectx):
        pol = ectx.pol
        y32, ylp, pre, pos_ids, mean, rstd = ops.text_embed_fwd(ids, word.detach(), pos.detach(), typ.detach(), ln_w.detach(), ln_b.detach(), eps,
                                                                pad_id, want_f32=True, lp_dtype=None if pol.f32 else pol.lp)
        ctx.ectx = ectx
        ctx.set_materialize_grads(False)
        ctx.shapes = (word.shape[0], pos.shape[0], typ.shape[0])
        ctx.save_for_backward(ids, pos_ids, pre, mean, rstd, ln_w, ln_b)
        if ylp is None:
            ylp = y32.new_empty(0)
        ctx.mark_non_differentiable(ylp)
        return y32, ylp

    @staticmethod
    def backward(ctx, g, _):
        ids, pos_ids, pre, mean, rstd, ln_w, ln_b = ctx.saved_tensors
        dpre, _, dg, db = _ln_bwd(ctx.ectx.pol, _c(g), pre, ln_w, ln_b, mean, rstd, need_lp=False)
        nv, npos, ntyp = ctx.shapes
        dword = ops.scatter_add_rows(dpre, ids.reshape(-1), nv) if ctx.needs_input_grad[1] else None
        dpos = ops.scatter_add_rows(dpre, pos_ids.reshape(-1), npos) if ctx.needs_input_grad[2] else None
        dtyp = None
        if ctx.needs_input_grad[3]:
            dtyp = ops.zeros_f32((ntyp, dpre.shape[1]), dpre.device)
            ops.colsum(dpre, out=dtyp[0])
        return None, dword, dpos, dtyp, dg, db, None, None, None


def text_embed(ectx, ids, word, pos, typ, ln_w, ln_b, eps, pad_id):
    y32, ylp = TextEmbedFn.apply(ids, word, pos, typ, ln_w, ln_b, eps, pad_id, ectx)
    return y32, (y32 if ectx.pol.f32 else ylp)


class PatchEmbedFn(torch.autograd.Function):
    """VideoMAE tubelet embedding of the KEPT tokens only (HF videomae:94-177): Conv3d(k=s=(2,16,16)) as a GEMM over
    gathered patches, + bias + fixed sin-cos position rows.  `embeddings[~mask]` keeps rows in ascending token order,
    which is the order of keep_idx.  Output f32 [B*nkeep, hidden]."""

    @staticmethod
    def forward(ctx, video, keep_idx, w, b, pos_table, ectx):
        pol = ectx.pol
        patches = ops.patchify(_c(video), keep_idx, pol.lp)
        w2d = w.detach().view(w.shape[0], -1)
        w_n, _ = ectx.cache._get(("patch", id(w)), (w,), lambda: (w2d, None) if pol.f32 else (ops.cast_weight(w2d, pol.lp, want_t=False)[0], None))
        resid = ops.gather_rows(pos_table, keep_idx.reshape(-1))
        y = ops.gemm_nt(patches, w_n, bias=b, resid=resid, out_dtype=torch.float32)
        ctx.ectx, ctx.wshape = ectx, tuple(w.shape)
        ctx.save_for_backward(patches)
        return y

    @staticmethod
    def backward(ctx, g):
        (patches,) = ctx.saved_tensors
        g_lp = _to_lp(ctx.ectx.pol, _c(g))
        dW = ops.gemm_tn(g_lp, patches).view(ctx.wshape) if ctx.needs_input_grad[2] else None
        dB = ops.colsum(g_lp) if ctx.needs_input_grad[3] else None
        return None, None, dW, dB, None, None


# ---------------------------------------------------------------------------------------------- wav2vec2 front-end
class Conv0Fn(torch.autograd.Function):
    """Conv1d(1, C, k, stride) on the raw waveform -> channels-last [B, T_out, C] in the operand dtype (HF wav2vec2:382-419)."""

    @staticmethod
    def forward(ctx, wave, w, b, stride, ectx):
        wave = _c(wave)
        K = w.shape[-1]
        T_out = (wave.shape[1] - K) // stride + 1
        y = ops.conv0_fwd(wave, w.detach(), b.detach() if b is not None else None, T_out, stride, ectx.pol.lp)
        ctx.stride, ctx.K, ctx.has_b = stride, K, b is not None
        ctx.save_for_backward(wave)
        return y

    @staticmethod
    def backward(ctx, g):
        (wave,) = ctx.saved_tensors
        dw, db = ops.conv0_bwd_w(wave, _c(g), ctx.K, ctx.stride, ctx.has_b)
        return None, dw, db, None, None


class GroupNormGeluFn(torch.autograd.Function):
    """GroupNorm(C groups) over time + GELU on channels-last activations (wav2vec2 'group' layer 0, HF wav2vec2:304-323)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, stats = ops.gn_gelu_fwd(_c(x), w.detach(), b.detach(), eps)
        ctx.save_for_backward(x, w, b, stats)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, b, stats = ctx.saved_tensors
        dx, dg, db = ops.gn_gelu_bwd(x, _c(g), w, b, stats)
        return dx, dg, db, None


# Input gradient of the strided convs without a column buffer (round 4; 0 = the NT GEMM into [T_out, k*Ci] + col2im of rounds 1-3).
CONV_PHASE_DGRAD = [os.environ.get("TAV_CONV_PHASE", "1") == "1"]
CONV_PAD = 2                  # zero rows in front of and behind every batch entry of a padded gradient buffer (covers k <= 3 * stride taps)
_CONV_PADDED = {}             # data_ptr of a padded buffer's interior view -> the buffer, from the layer that wrote it to the one that reads it


class ConvGemmFn(torch.autograd.Function):
    """Conv1d(C_in, C_out, k, stride) on channels-last [B, T_in, C_in] as one NT GEMM over overlapping rows
    (lda = stride*C_in, K = k*C_in), optional bias and fused exact GELU; gradients: TN GEMM (dW), NT GEMM + col2im (dx).

    Returns (y, pre): `pre` is the pre-activation when `gelu` (else None).  A following conv layer passes it back as `u_in`
    (then x must be gelu(u_in), the producer's y): its backward folds gelu'(u_in) into the col2im pass and hands the result to the
    producer through `pre`'s gradient, so the chain of the wav2vec2 feature extractor (HF wav2vec2:382-419) runs no separate
    GELU-backward pass between the layers."""

    @staticmethod
    def forward(ctx, x, u_in, w, b, stride, gelu, ectx):
        x = _c(x)
        B, T_in, Ci = x.shape
        Co, _, k = w.shape
        T_out = (T_in - k) // stride + 1
        w_n = ectx.cache.conv(w, stride)[0]
        if u_in is None:
            _CONV_PADDED.clear()                            # (first conv of a stack, forward: nothing of an earlier backward is still wanted)
        res = ops.gemm_nt(x, w_n, bias=b, act=1 if gelu else 0, want_pre=gelu, M=T_out, N=Co, K=k * Ci, lda=stride * Ci, ldb=k * Ci, ldc=Co,
                          nzb=B, a_zb=T_in * Ci, c_zb=T_out * Co, out_shape=(B, T_out, Co))
        y, pre = res if gelu else (res, None)
        ctx.ectx, ctx.geom, ctx.has_b = ectx, (B, T_in, Ci, Co, k, stride, T_out), b is not None
        ctx.chained = u_in is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, w, pre, u_in)
        return y, pre

    @staticmethod
    def backward(ctx, g, g_pre):
        x, w, pre, u_in = ctx.saved_tensors
        B, T_in, Ci, Co, k, stride, T_out = ctx.geom
        du = g_pre                                              # from a chained consumer: already times gelu'(pre); possibly a view of a padded buffer
        if g is not None:
            d2 = ops.gelu_bwd(pre, _c(g)) if pre is not None else _c(g)
            du = d2 if du is None else _c(du) + d2
        if du is None:
            return None, None, None, None, None, None, None
        _, w_t, w_ph = ctx.ectx.cache.conv(w, stride)
        P = CONV_PAD
        TP = T_out + 2 * P
        # `du` in PADDED form [B, P + T_out + P, Co] with zero frames (what the phase GEMMs below read past the ends): either this layer's
        # chained consumer produced it that way (its `dui`, recognised by its storage), or -- small tensors only -- it is copied into one here.
        du_pad = _CONV_PADDED.pop(du.data_ptr(), None) if g is None else None
        if du_pad is not None and (tuple(du.shape) != (B, T_out, Co) or tuple(du.stride()) != (TP * Co, Co, 1) or tuple(du_pad.shape) != (B, TP, Co)):
            du_pad = None
        want_dx = ctx.needs_input_grad[1] if ctx.chained else ctx.needs_input_grad[0]
        phase = (CONV_PHASE_DGRAD[0] and w_ph is not None and want_dx and (k + stride - 1) // stride - 1 <= P        # taps behind / rows past the end
                 and (T_in - 1) // stride + 1 <= T_out + P)                                                         # ... stay inside the zero frames
        if du_pad is None:
            du = _c(du)
            if phase and du.numel() <= (1 << 23):
                du_pad = ops.zero_pad_rows(torch.empty(B, TP, Co, dtype=du.dtype, device=du.device), B, T_out, Co, P)
                du_pad[:, P:P + T_out].copy_(du)
            else:
                phase = False
        du_v = du_pad[:, P:P + T_out] if du_pad is not None else du                 # [B, T_out, Co], batch stride za
        za = TP * Co if du_pad is not None else T_out * Co
        dw = db = dx = dui = None
        if ctx.needs_input_grad[2]:
            dw = ops.gemm_tn(du_v, x, N1=Co, N2=k * Ci, lda=Co, ldb=stride * Ci, rows_per_batch=T_out, nbatch=B, a_zb=za, b_zb=T_in * Ci,
                             perm_inner=Ci, perm_outer=k, out_shape=(Co, Ci, k))
        if ctx.has_b and ctx.needs_input_grad[3]:
            db = ops.colsum(du_pad.view(B * TP, Co) if du_pad is not None else du.view(B * T_out, Co))     # (the zero frames add nothing)
        if want_dx and phase:
            # dx[s*m + r] = sum_q dy[m - q] . W[:, :, r + q*s]: per residue r ONE NT GEMM over Q_r consecutive rows of the padded dy (overlapping
            # rows, lda = Co, K = Q_r * Co) that writes every s-th row of dx -- no [T_out, k*Ci] column buffer, no col2im pass, f32 accumulation
            # over the taps.  A chained layer multiplies by gelu'(u_in) in the epilogue and writes into the NEXT padded buffer.
            if ctx.chained:
                out = ops.zero_pad_rows(torch.empty(B, T_in + 2 * P, Ci, dtype=du_v.dtype, device=du_v.device), B, T_in, Ci, P)
                zc, obase = (T_in + 2 * P) * Ci, P * Ci
            else:
                out = torch.empty(B, T_in, Ci, dtype=du_v.dtype, device=du_v.device)
                zc, obase = T_in * Ci, 0
            oflat, aflat, uflat = out.view(-1), du_pad.view(-1), (u_in.reshape(-1) if ctx.chained else None)
            off = 0
            for r in range(stride):
                Q = (k - r + stride - 1) // stride
                Mr = (T_in - r + stride - 1) // stride
                ops.gemm_nt(aflat[(P - (Q - 1)) * Co:], w_ph[off:off + Ci * Q * Co], out=oflat[obase + r * Ci:], M=Mr, N=Ci, K=Q * Co, lda=Co, ldb=Q * Co,
                            ldc=stride * Ci, nzb=B, a_zb=TP * Co, c_zb=zc, gelu_in=uflat[r * Ci:] if ctx.chained else None,
                            ld_gelu=stride * Ci, gelu_zb=T_in * Ci)
                off += Ci * Q * Co
            if ctx.chained:
                dui = out[:, P:P + T_in]
                _CONV_PADDED[dui.data_ptr()] = out
            else:
                dx = out
        elif want_dx:
            dcol = ops.gemm_nt(_c(du_v).view(B * T_out, Co), w_t, out_shape=(B * T_out, k * Ci))
            d = ops.col2im_1d(dcol, B, T_in, T_out, Ci, k, stride, pre_act=u_in if ctx.chained else None)
            if ctx.chained:
                dui = d
            else:
                dx = d
        return dx, dui, dw, db, None, None, None


class PosConvFn(torch.autograd.Function):
    """hidden + gelu(grouped Conv1d(H, H, k=128, pad=64, groups=16)(hidden)[..., :-1]) with weight-norm (dim=2) parameters
    (HF wav2vec2:326-379, used at :758-759/:833-834 and reference models/tav.py:360).  f32 residual in, f32 out."""

    @staticmethod
    def forward(ctx, x, v, g, bias, B, T, G, ectx):
        pol = ectx.pol
        x = _c(x)
        H = x.shape[1]
        Cg, K = H // G, v.shape[2]
        w, _, norms = ectx.cache.posconv(v, g)
        pad = K // 2
        xg = ops.group_pad(x, B, T, H, G, pad, pad, pol.lp)
        TP = T + 2 * pad
        y, pre = ops.gemm_nt(xg, w, bias=bias, act=1, resid=x, want_pre=True, out_dtype=torch.float32, M=T, N=Cg, K=K * Cg, lda=Cg, ldb=K * Cg,
                             ldc=H, nzb=B, nzg=G, a_zb=G * TP * Cg, a_zg=TP * Cg, b_zg=Cg * K * Cg, c_zb=T * H, c_zg=Cg, bias_zg=Cg,
                             out_shape=(B * T, H))
        ctx.ectx, ctx.geom = ectx, (B, T, H, G, Cg, K)
        ctx.save_for_backward(xg, pre, v, g, norms)
        return y

    @staticmethod
    def backward(ctx, gout):
        xg, pre, v, g, norms = ctx.saved_tensors
        pol = ctx.ectx.pol
        B, T, H, G, Cg, K = ctx.geom
        gout = _c(gout)
        _, wf, _ = ctx.ectx.cache.posconv(v, g)
        du = ops.gelu_bwd(pre, gout)                                   # f32 [B*T, H]
        pad = K // 2
        dug = ops.group_pad(du, B, T, H, G, pad - 1, pad, pol.lp)      # flipped-kernel form of the input gradient
        TPd, TP = T + 2 * pad - 1, T + 2 * pad
        dx = ops.gemm_nt(dug, wf, resid=gout, out_dtype=torch.float32, M=T, N=Cg, K=K * Cg, lda=Cg, ldb=K * Cg, ldc=H, nzb=B, nzg=G,
                         a_zb=G * TPd * Cg, a_zg=TPd * Cg, b_zg=Cg * K * Cg, c_zb=T * H, c_zg=Cg, out_shape=(B * T, H))
        du_lp = _to_lp(pol, du)
        dw = torch.empty(G, Cg, K * Cg, dtype=torch.float32, device=du.device)
        for gi in range(G):
            ops.gemm_tn(du_lp[:, gi * Cg:], xg[:, gi], out=dw[gi], N1=Cg, N2=K * Cg, lda=H, ldb=Cg, rows_per_batch=T, nbatch=B, a_zb=T * H,
                        b_zb=G * TP * Cg)
        dv, dg = ops.weight_norm_bwd(v, g, norms, dw)
        dbias = ops.colsum(du)
        return dx, dv, dg.view_as(g), dbias, None, None, None, None


# ---------------------------------------------------------------------------------------------- sequence concat / pooling / tail
class ConcatSeqFn(torch.autograd.Function):
    """torch.concat(parts, dim=1) for [B, S_i, W] f32 parts (models/tav.py:372)."""

    @staticmethod
    def forward(ctx, B, *parts):
        W = parts[0].shape[-1]
        lens = [p.numel() // (B * W) for p in parts]
        Sf = sum(lens)
        out = torch.empty(B, Sf, W, dtype=torch.float32, device=parts[0].device)
        flat = out.view(B, Sf * W)
        off = 0
        for p, L in zip(parts, lens):
            ops.cast2d(_c(p).view(B, L * W), torch.float32, out=flat[:, off * W:(off + L) * W])
            off += L
        ctx.lens, ctx.W, ctx.B = lens, W, B
        return out

    @staticmethod
    def backward(ctx, g):
        B, W = ctx.B, ctx.W
        flat = _c(g).view(B, -1)
        outs, off = [], 0
        for L in ctx.lens:
            outs.append(ops.cast2d(flat[:, off * W:(off + L) * W], torch.float32).view(B * L, W))
            off += L
        return (None, *outs)


class TailFn(torch.autograd.Function):
    """models/tav.py:478-499: mean-pool the audio/video/fusion streams, four LayerNorms, concat [av, t, aud, vid] -> [B,3072],
    optional dropout (check == 'train'), Linear(3072, output_dim).  All f32 (tiny tensors)."""

    @staticmethod
    def forward(ctx, av_seq, t_pooled, aud_seq, vid_seq, B, S_av, S_aud, S_vid, p_drop, seed, rand_w, rand_b, bert_w, bert_b, aud_w, aud_b,
                vid_w, vid_b, lin_w, lin_b):
        W = 768
        pooled = [ops.mean_pool_fwd(_c(av_seq), B, S_av), _c(t_pooled), ops.mean_pool_fwd(_c(aud_seq), B, S_aud), ops.mean_pool_fwd(_c(vid_seq), B, S_vid)]
        norms = [(rand_w, rand_b), (bert_w, bert_b), (aud_w, aud_b), (vid_w, vid_b)]
        cat = torch.empty(B, 4 * W, dtype=torch.float32, device=av_seq.device)
        stats = []
        for j, (x, (w, b)) in enumerate(zip(pooled, norms)):
            y32, _, mean, rstd = ops.ln_fwd(x, w.detach(), b.detach(), 1e-5, want_f32=True)
            ops.cast2d(y32, torch.float32, out=cat[:, j * W:(j + 1) * W])
            stats += [mean, rstd]
        mask = None
        feat = cat
        if p_drop > 0.0:
            feat, mask = ops.dropout_fwd(cat, p_drop, seed, 0)
        logits = ops.head_fwd(feat, lin_w.detach(), lin_b.detach())
        ctx.geom = (B, S_av, S_aud, S_vid, p_drop)
        ctx.save_for_backward(feat, mask, lin_w, *pooled, *stats, rand_w, rand_b, bert_w, bert_b, aud_w, aud_b, vid_w, vid_b)
        return logits

    @staticmethod
    def backward(ctx, g):
        B, S_av, S_aud, S_vid, p_drop = ctx.geom
        sv = ctx.saved_tensors
        feat, mask, lin_w = sv[:3]
        pooled, stats, nw = sv[3:7], sv[7:15], sv[15:23]
        W = 768
        dfeat, dW, db = ops.head_bwd(feat, lin_w, _c(g))
        if mask is not None:
            dfeat = ops.dropout_bwd(dfeat, mask, p_drop)
        dpool, dparams = [], []
        for j in range(4):
            dy = ops.cast2d(dfeat[:, j * W:(j + 1) * W], torch.float32)
            dx, _, dg, dbt = ops.ln_bwd(dy, pooled[j], nw[2 * j], nw[2 * j + 1], stats[2 * j], stats[2 * j + 1], want_f32=True)
            dpool.append(dx)
            dparams += [dg, dbt]
        d_av, _ = ops.mean_pool_bwd(dpool[0], B, S_av)
        d_aud, _ = ops.mean_pool_bwd(dpool[2], B, S_aud)
        d_vid, _ = ops.mean_pool_bwd(dpool[3], B, S_vid)
        return (d_av, dpool[1], d_aud, d_vid, None, None, None, None, None, None, *dparams, dW, db)


class HeadFn(torch.autograd.Function):
    """dropout(p) -> Linear(K, N) on a [B, K] f32 feature matrix, N small (7): the classifier head of the single- and dual-modal
    models (reference SingleModels/models/text.py:61-65; models/tav.py:497-499 is the same pair inside TailFn)."""

    @staticmethod
    def forward(ctx, x, p_drop, seed, w, b):
        x = _c(x)
        mask = None
        if p_drop > 0.0:
            x, mask = ops.dropout_fwd(x, p_drop, seed, 0)
        ctx.p_drop = p_drop
        ctx.save_for_backward(x, mask, w)
        return ops.head_fwd(x, w.detach(), b.detach())

    @staticmethod
    def backward(ctx, g):
        x, mask, w = ctx.saved_tensors
        dx, dW, db = ops.head_bwd(x, w, _c(g))
        if mask is not None:
            dx = ops.dropout_bwd(dx, mask, ctx.p_drop)
        return dx, None, None, dW, db


class PoolNormCatFn(torch.autograd.Function):
    """cat_j LN_j(pool_j(x_j)) for a list of branches, each either already pooled [B, 768] or a sequence [B*S_j, 768] that is
    mean-pooled over its S_j tokens first (models/tav.py:478-495 without the fusion branch)."""

    @staticmethod
    def forward(ctx, B, seq_lens, *args):
        n = len(seq_lens)
        xs, params = args[:n], args[n:]
        pooled, stats, outs = [], [], []
        for j in range(n):
            xp = _c(xs[j]) if seq_lens[j] == 0 else ops.mean_pool_fwd(_c(xs[j]), B, seq_lens[j])
            y, _, mean, rstd = ops.ln_fwd(xp, params[2 * j], params[2 * j + 1], 1e-5, want_f32=True)
            pooled.append(xp)
            stats += [mean, rstd]
            outs.append(y)
        ctx.geom = (B, tuple(seq_lens))
        ctx.save_for_backward(*pooled, *stats, *params)
        return torch.cat(outs, dim=1)

    @staticmethod
    def backward(ctx, g):
        B, seq_lens = ctx.geom
        n = len(seq_lens)
        sv = ctx.saved_tensors
        pooled, stats, params = sv[:n], sv[n:3 * n], sv[3 * n:]
        W = pooled[0].shape[1]
        dxs, dparams = [], []
        for j in range(n):
            dy = ops.cast2d(g[:, j * W:(j + 1) * W], torch.float32)
            dx, _, dg, dbt = ops.ln_bwd(dy, pooled[j], params[2 * j], params[2 * j + 1], stats[2 * j], stats[2 * j + 1], want_f32=True)
            if seq_lens[j]:
                dx, _ = ops.mean_pool_bwd(dx, B, seq_lens[j])
            dxs.append(dx)
            dparams += [dg, dbt]
        return (None, None, *dxs, *dparams)


class CrossEntropyFn(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss (optionally class-weighted), mean reduction (utils/global_functions.py:63-64)."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        loss, dlog = ops.cross_entropy(_c(logits), target, weight)
        ctx.save_for_backward(dlog)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None, None


# ---------------------------------------------------------------------------------------------- TransformerBlock (alt. fusion stack)
class TransformerBlockFn(torch.autograd.Function):
    """One block of the reference's TransformerEncoder (utils/TAVFormer.py:93-142 with MultiHeadAttention :10-90):
        a  = Wo . scramble(attn(x Wq, x Wk, x Wv; additive key mask BEFORE softmax)) + bo          bias-free q/k/v (:24-26)
        n1 = LN1(dropout1(x + a));   f = W2 . gelu(W1 . dropout_f(n1) + b1) + b2;   out = LN2(dropout2(f + n1))
    `scramble` is the reference's `scores[B*h,S,d].transpose(1,2).contiguous().view(B,S,h*d)` (:84): per batch the [S,H] -> [H,S]
    transpose of the token-major attention output re-read as [S,H].  early_div (:45-46 vs :62-63) divides by sqrt(64) = 8, an exact
    power of two, so both orders give bit-identical results and share one code path.
    params = (wq, wk, wv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b); p_drop > 0 draws three masks from `seed`."""

    @staticmethod
    def forward(ctx, x, key_mask, ectx, B, S, nh, p_drop, seed, *params):
        pol, cache = ectx.pol, ectx.cache
        wq, wk, wv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b = params
        H = nh * 64
        x = _c(x)
        x_lp = _to_lp(pol, x)
        wqkv, _, _ = cache.qkv(wq, wk, wv, None, None, None)
        wo_n, _ = cache.linear(wo)
        w1_n, _ = cache.linear(w1)
        w2_n, _ = cache.linear(w2)
        qkv = ops.gemm_nt(x_lp, wqkv)
        mode = 1 if key_mask is not None else 0
        o, lse, _ = ops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nh, key_mask=key_mask, mask_mode=mode)
        o_scr = ops.transpose2d(o, S, H, B)
        y1 = ops.gemm_nt(o_scr, wo_n, bias=bo, resid=x, out_dtype=torch.float32)
        m1 = m2 = m3 = None
        if p_drop > 0:
            y1, m1 = ops.dropout_fwd(y1, p_drop, seed, 0)
        n1, n1_lp, mean1, rstd1 = _ln_fwd(pol, y1, ln1_w, ln1_b, 1e-5, need_f32=True)
        f_in = n1_lp
        if p_drop > 0:
            d, m2 = ops.dropout_fwd(n1, p_drop, seed, 1 << 40)
            f_in = _to_lp(pol, d)
        h, u = ops.gemm_nt(f_in, w1_n, bias=b1, act=3, want_pre=True)
        y2 = ops.gemm_nt(h, w2_n, bias=b2, resid=n1, out_dtype=torch.float32)
        if p_drop > 0:
            y2, m3 = ops.dropout_fwd(y2, p_drop, seed, 2 << 40)
        out, _, mean2, rstd2 = _ln_fwd(pol, y2, ln2_w, ln2_b, 1e-5, need_f32=True)
        ctx.ectx, ctx.geom = ectx, (B, S, nh, p_drop, mode)
        ctx.save_for_backward(x_lp, qkv, o, o_scr, lse, y1, f_in, u, h, y2, mean1, rstd1, mean2, rstd2, key_mask, m1, m2, m3, *params)
        return out

    @staticmethod
    def backward(ctx, g):
        pol, cache = ctx.ectx.pol, ctx.ectx.cache
        B, S, nh, p_drop, mode = ctx.geom
        sv = ctx.saved_tensors
        x_lp, qkv, o, o_scr, lse, y1, f_in, u, h, y2, mean1, rstd1, mean2, rstd2, key_mask, m1, m2, m3 = sv[:18]
        wq, wk, wv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b = sv[18:]
        H = nh * 64
        _, wqkv_t, _ = cache.qkv(wq, wk, wv, None, None, None)
        _, wo_t = cache.linear(wo)
        _, w1_t = cache.linear(w1)
        _, w2_t = cache.linear(w2)
        dy2, _, dg2, db2 = _ln_bwd(pol, _c(g), y2, ln2_w, ln2_b, mean2, rstd2, need_lp=False)
        if m3 is not None:
            dy2 = ops.dropout_bwd(dy2, m3, p_drop)
        dy2_lp = _to_lp(pol, dy2)
        dW2, dB2 = ops.gemm_tn(dy2_lp, h, want_bias=True)
        du = ops.gemm_nt(dy2_lp, w2_t, gelu_in=u, act=4)
        dW1, dB1 = ops.gemm_tn(du, f_in, want_bias=True)
        df = ops.gemm_nt(du, w1_t, out_dtype=torch.float32)               # grad wrt dropout_f(n1)
        if m2 is not None:
            df = ops.dropout_bwd(df, m2, p_drop)
        dn1, _ = ops.add_f32(df, dy2)                                      # + residual path (f + n1)
        dy1, _, dg1, db1 = _ln_bwd(pol, dn1, y1, ln1_w, ln1_b, mean1, rstd1, need_lp=False)
        if m1 is not None:
            dy1 = ops.dropout_bwd(dy1, m1, p_drop)
        dy1_lp = _to_lp(pol, dy1)
        dWo, dBo = ops.gemm_tn(dy1_lp, o_scr, want_bias=True)
        do_scr = ops.gemm_nt(dy1_lp, wo_t)
        do = ops.transpose2d(do_scr, H, S, B)                              # inverse of the scramble
        dqkv = ops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, do, lse, None, B, S, nh, key_mask=key_mask, mask_mode=mode)
        dWqkv = ops.gemm_tn(dqkv, x_lp)
        g0 = ops.gemm_nt(dqkv, wqkv_t, resid=dy1, out_dtype=torch.float32)
        return (g0, None, None, None, None, None, None, None, dWqkv[:H], dWqkv[H:2 * H], dWqkv[2 * H:], dWo, dBo, dg1, db1, dW1, dB1, dW2, dB2, dg2, db2)
