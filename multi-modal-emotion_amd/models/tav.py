"""Drop-in for the reference's models/tav.py model classes, executed by libtavhip on MI355X.

  PreFormer   reference models/tav.py:249-417   modality front-ends -> fused token sequence + masks
  TAVForMAE   reference models/tav.py:420-504   3 encoders + fusion encoder + 7-way head
  collate_batch  reference models/tav.py:174-246   (mask / pad logic only: file decoding is out of scope)

Same constructor arguments, forward() signatures and state_dict keys.  `from_pretrained` checkpoints are not
reachable offline, so sub-models are built from a geometry preset (config.py) and initialised randomly; load a
reference `best.pt` with `load_state_dict(remap_reference_keys(sd))`.
The reference keeps PreFormer on the CPU and ships activations back and forth (models/tav.py:352,359,363); here every
tensor stays resident in HBM: inputs given on the CPU are moved to the GPU once, outputs are returned on the GPU
(the reference's `.to(device)` calls in tav_train.py:39-40 then cost nothing).
"""
import numpy as np
import torch
from torch import nn

from .. import config as C
from .. import engine as E
from .. import runtime
from ..encoders import AudioEncoder, TextEncoder, VideoEncoder
from ..utils.TAVFormer import VideoMAEEncoder

FP16_MIN = float(torch.finfo(torch.float16).min)


def _dev(device):
    d = torch.device(device if device is not None else "cuda")
    if d.type != "cuda":
        d = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else d
    if d.type != "cuda":
        raise RuntimeError("the TAV hot path runs on libtavhip (HIP, MI355X) only: no GPU is visible and there is no CPU fallback")
    return d


class PreFormer(nn.Module):
    """Modality front-ends (reference models/tav.py:249-417)."""

    def __init__(self, config=None):
        super().__init__()
        cfg = config if config is not None else C.default_config()
        self.cfg = cfg
        Ha = cfg["audio"]["hidden"]
        self.bert = TextEncoder(cfg["text"])
        self.wav2vec2 = AudioEncoder(cfg["audio"])
        self.masked_spec_embed = nn.Parameter(torch.FloatTensor(Ha).uniform_())
        self.videomae = VideoEncoder(cfg["video"])
        self.wav_2_768 = nn.Linear(Ha, 768)
        nn.init.xavier_normal_(self.wav_2_768.weight)
        if cfg["video"]["hidden"] != 768:                   # BASELINE config 5 (videomae-large): the bridge the audio branch already has
            self.vid_2_768 = nn.Linear(cfg["video"]["hidden"], 768)
            nn.init.xavier_normal_(self.vid_2_768.weight)
        self.check_shapes = 1

    # -- reference helpers, same names (models/tav.py:308-342) --
    def _get_feat_extract_output_lengths(self, input_lengths, add_adapter=None):
        for k, s in zip(self.cfg["audio"]["conv_kernel"], self.cfg["audio"]["conv_stride"]):
            input_lengths = torch.div(input_lengths - k, s, rounding_mode="floor") + 1
        return input_lengths

    def _get_feature_vector_attention_mask(self, feature_vector_length, attention_mask, add_adapter=None):
        non_padded = attention_mask.sum(dim=-1)          # == cumsum(-1)[:, -1] of the reference (:329) without a scan kernel
        out_len = self._get_feat_extract_output_lengths(non_padded).to(torch.long)
        # same result as the reference's "set index out_len-1, flip, cumsum, flip" (:337-341) without an index_put, which
        # synchronises the host and cannot be captured into a hipGraph
        return torch.arange(feature_vector_length, device=attention_mask.device)[None, :] < out_len[:, None]

    @staticmethod
    def _span_mask(B, L, lens, prob, length, min_masks, dev):
        """bool [B, L]: per row n spans of `length` positions, n = max(int(prob * len / length + eps), min_masks) with ONE eps ~ U[0,1) per call,
        starts drawn without replacement from the row's valid range [0, len - length] -- HF `_compute_mask_indices` (wav2vec2:101), the
        reference's sampler (models/tav.py:283-289, :296-301), drawn with torch ops on the device (no host read: capturable).  The number of spans a
        row can take is bounded by the static L; surplus slots are switched off by comparison instead of by shape."""
        eps = torch.rand(1, device=dev).expand(B)
        n = torch.clamp((prob * lens / length + eps).floor(), min=float(min_masks))
        n = torch.minimum(n, torch.clamp((lens - (length - 1)).floor(), min=0.0))           # never more spans than start positions
        n = torch.minimum(n, torch.full_like(n, float(L // length)))
        max_spans = max(int(min_masks), int(prob * L / length + 1.0), 1)                     # static bound (eps < 1)
        pos = torch.arange(L, device=dev)[None, :]
        score = torch.rand(B, L, device=dev)
        score = torch.where(pos < (lens[:, None] - (length - 1)), score, torch.full_like(score, 2.0))   # starts outside the valid range sort last
        starts = score.argsort(dim=1)[:, :max_spans]                                          # without replacement
        live = torch.arange(max_spans, device=dev)[None, :] < n[:, None]
        cover = (pos[:, None, :] >= starts[:, :, None]) & (pos[:, None, :] < starts[:, :, None] + length) & live[:, :, None]
        return cover.any(dim=1)

    def _mask_hidden_states(self, hidden, B, T, attention_mask, training=False):
        """SpecAugment (reference models/tav.py:269-306).  Along TIME (:281-290): spans of mask_time_length frames inside each row's valid range are
        replaced by `masked_spec_embed`; along the FEATURE axis (:292-304, round 4): spans of mask_feature_length channels are zeroed for every
        frame of the row -- inactive under the Wav2Vec2Config default mask_feature_prob = 0, which is what the reference's checkpoint carries.
        The reference samples on the host with numpy; here the same distribution is drawn ON THE DEVICE (`_span_mask`), so a training step with
        train=True stays graph-capturable.  Only active when train=True, which is outside the parity / benchmark configuration."""
        ac = self.cfg["audio"] if hasattr(self, "cfg") else {}
        mask_prob, mask_len, min_masks = ac.get("mask_time_prob", 0.05), ac.get("mask_time_length", 10), ac.get("mask_time_min_masks", 2)   # Wav2Vec2Config defaults; min_masks as the reference passes it
        f_prob, f_len, f_min = ac.get("mask_feature_prob", 0.0), ac.get("mask_feature_length", 10), ac.get("mask_feature_min_masks", 0)
        if not training or T < mask_len:                                                       # (:277-278)
            return hidden
        dev = hidden.device
        if mask_prob > 0:
            if attention_mask is not None:
                lens = attention_mask.to(dev).sum(-1).to(torch.float32)                       # frames that are not padding, per row
            else:
                lens = torch.full((B,), float(T), device=dev)
            sel = self._span_mask(B, T, lens, mask_prob, mask_len, min_masks, dev).reshape(B * T, 1)
            hidden = torch.where(sel, self.masked_spec_embed.to(hidden.dtype)[None, :], hidden)
        if f_prob > 0:
            Hh = hidden.shape[1]
            fsel = self._span_mask(B, Hh, torch.full((B,), float(Hh), device=dev), f_prob, f_len, f_min, dev)      # [B, H]: no attention mask on this axis (:296-301)
            hidden = torch.where(fsel[:, None, :].expand(B, T, Hh).reshape(B * T, Hh), torch.zeros((), dtype=hidden.dtype, device=dev), hidden)
        return hidden

    def forward(self, input_ids=None, audio_features=None, video_embeds=None, text_mask=None, audio_mask=None, visual_mask=None,
                device="cpu", train=False, n_visual_true=None):
        """n_visual_true (optional): number of True entries per row of visual_mask; passing it avoids one host sync."""
        dev = _dev(device if str(device) != "cpu" else None)
        ectx = runtime.ctx()
        if audio_features.is_cuda:
            runtime.mark_inputs_ready()
        B = audio_features.shape[0]
        parts = []
        St = 0
        audio_features, video_embeds, visual_mask = audio_features.to(dev, torch.float32), video_embeds.to(dev, torch.float32), visual_mask.to(dev)
        Sa = self.wav2vec2.conv_out_len(audio_features.shape[1])
        if audio_mask is not None:
            audio_mask = self._get_feature_vector_attention_mask(Sa, audio_mask.to(dev))     # :355 bool [B, Sa]

        def video_frontend():
            xv, nv = self.videomae.embed(video_embeds, ~visual_mask, n_visual_true)    # :368
            if hasattr(self, "vid_2_768"):
                xv = E.LinearFn.apply(xv, None, self.vid_2_768.weight, self.vid_2_768.bias, None, ectx, True)
            return xv, nv

        def audio_frontend():
            feats = self.wav2vec2.feature_extractor_fwd(audio_features)                 # :352  [B, Sa, 512]
            hidden = self.wav2vec2.feature_projection_fwd(feats)                        # :356  f32 [B*Sa, Ha]
            hidden = self._mask_hidden_states(hidden, B, Sa, audio_mask, train)         # :359
            hidden = self.wav2vec2.pos_conv_fwd(hidden, B, Sa)                          # :360
            enc = self.wav2vec2.encoder
            hidden, hidden_lp = E.layer_norm_f32(ectx, hidden, enc.layer_norm.weight, enc.layer_norm.bias, self.cfg["audio"]["eps"])   # :361
            return E.LinearFn.apply(hidden, hidden_lp if not ectx.pol.f32 else None, self.wav_2_768.weight, self.wav_2_768.bias, None, ectx, True)   # :363

        main = torch.cuda.current_stream()
        side = runtime.multistream[0] and runtime.front_side[0] and audio_features.is_cuda
        if side:        # the three front-ends are independent (models/tav.py:349-368): audio and video on side streams, text here
            ev = torch.cuda.Event()
            ev.record(main)
            s_a, s_v = runtime.front_streams(2)
            runtime.share_with(s_a, audio_features, audio_mask)
            runtime.share_with(s_v, video_embeds, visual_mask)
            with torch.cuda.stream(s_a):
                runtime.stream_wait(s_a, main, ev)
                x_audio = audio_frontend()
            with torch.cuda.stream(s_v):
                runtime.stream_wait(s_v, main, ev)
                x_video, Nv = video_frontend()
        if input_ids is not None:
            input_ids = input_ids.to(dev)
            x_text, _ = self.bert.embed(input_ids)                                      # :349
            St = input_ids.shape[1]
            parts.append(x_text)
        if side:
            for st, ten in ((s_a, x_audio), (s_v, x_video)):
                runtime.stream_wait(main, st)
                ten.record_stream(main)
        else:
            x_audio = audio_frontend()
            x_video, Nv = video_frontend()
        parts += [x_audio, x_video]
        tav = E.ConcatSeqFn.apply(B, *parts)                                            # :372-375

        # static modality ids and masks (:378-409) -- tiny host-logic tensors
        pos = [torch.zeros((B, St), device=dev)] if input_ids is not None else []
        pos += [torch.ones((B, Sa), device=dev), torch.ones((B, Nv), device=dev) + 1]
        tav_embed = torch.concat(pos, dim=1).type(torch.long)
        masks = []
        if input_ids is not None and text_mask is not None:
            masks.append((1.0 - text_mask.to(dev)[:, None, None, :]) * FP16_MIN)        # :383
        if audio_mask is not None:
            masks.append(1.0 - audio_mask[:, None, None, :] * FP16_MIN)                 # :390 (precedence as written)
        masks.append(torch.zeros((B, 1, 1, Nv), device=dev, dtype=torch.float))         # :397
        attention_mask = torch.concat(masks, dim=-1)                                    # :409
        if self.check_shapes == 1:
            self.check_shapes += 1
            print(f"Text shape is {(B, St, 768)}\nAudio shape is {(B, Sa, 768)}\nVideo shape is {(B, Nv, 768)}\n", flush=True)
        return tav, tav_embed, attention_mask


class TAVForMAE(nn.Module):
    """Tri-modal classifier (reference models/tav.py:420-504)."""

    def __init__(self, args, config=None):
        super().__init__()
        cfg = config if config is not None else C.default_config()
        self.cfg = cfg
        self.output_dim = args["output_dim"]
        self.dropout_p = float(args["dropout"])
        self.learn_PosEmbeddings = args["learn_PosEmbeddings"]
        self.num_layers = args["num_layers"]            # stored, unused: the fusion depth is fixed (reference :430,442)
        Ha = cfg["audio"]["hidden"]
        self.test_ctr = 1
        self.train_ctr = 1
        self.embedding = nn.Embedding(3, 768)
        self.embedding.weight.requires_grad = bool(self.learn_PosEmbeddings)
        self.bert = TextEncoder(cfg["text"])
        self.bert_norm = nn.LayerNorm(768)
        self.random_mae_config = dict(cfg["fusion"])
        self.random_mae_encoder = VideoMAEEncoder(self.random_mae_config, cfg["fusion"]["layers"]).apply(self.randomize_model)
        self.rand_norm = nn.LayerNorm(768)
        self.vid_norm = nn.LayerNorm(768)
        self.aud_norm = nn.LayerNorm(768)
        self.linear1 = nn.Linear(768 * 4, self.output_dim)
        self.wav2vec2 = AudioEncoder(cfg["audio"])
        self.videomae = VideoEncoder(cfg["video"])
        self.wav_2_768_2 = nn.Linear(Ha, 768)
        nn.init.xavier_normal_(self.wav_2_768_2.weight)
        if cfg["video"]["hidden"] != 768:
            self.vid_2_768_2 = nn.Linear(cfg["video"]["hidden"], 768)
            nn.init.xavier_normal_(self.vid_2_768_2.weight)
        self._drop_calls = 0

    def randomize_model(self, model):
        """reference :461-471: xavier_uniform Linear/Embedding weights, zero biases, LayerNorm weight = 1 / bias = 0."""
        for _, m in model.named_modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.xavier_uniform_(m.weight)
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
        return model

    def forward(self, input_ids, text_attention_mask, audio_features, video_embeds, visual_mask, hidden_states, pos_embed, attention_mask,
                batch_size=2, check="train", n_visual_true=None):
        dev = _dev(hidden_states.device if hidden_states.is_cuda else None)
        B, Sf, _ = hidden_states.shape
        ectx = runtime.ctx()
        nkeep = None if n_visual_true is None else visual_mask.shape[1] - n_visual_true
        audio_features, video_embeds = audio_features.to(dev, torch.float32), video_embeds.to(dev, torch.float32)
        visual_mask, input_ids, text_attention_mask = visual_mask.to(dev), input_ids.to(dev), text_attention_mask.to(dev)

        def video_branch():
            v, sv = self.videomae(video_embeds, visual_mask, nkeep)                      # :480
            if hasattr(self, "vid_2_768_2"):
                v = E.LinearFn.apply(v, None, self.vid_2_768_2.weight, self.vid_2_768_2.bias, None, ectx, True)
            return v, sv

        def audio_branch():
            a, a_lp, sa = self.wav2vec2(audio_features)                                  # :476
            return E.LinearFn.apply(a, a_lp if not ectx.pol.f32 else None, self.wav_2_768_2.weight, self.wav_2_768_2.bias, None, ectx, True), sa   # :478

        main = torch.cuda.current_stream()
        if runtime.multistream[0]:
            ev = runtime.take_inputs_event()
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(main)
            s_aud, s_vid, s_txt = runtime.branch_streams(3)
            runtime.share_with(s_aud, audio_features)
            runtime.share_with(s_vid, video_embeds, visual_mask)
            runtime.share_with(s_txt, input_ids, text_attention_mask)
            with torch.cuda.stream(s_vid):
                runtime.stream_wait(s_vid, main, ev)           # (ev was recorded on the caller's stream: here, or by PreFormer.forward)
                vid, Sv = video_branch()
            with torch.cuda.stream(s_aud):
                runtime.stream_wait(s_aud, main, ev)
                aud, Sa = audio_branch()
            with torch.cuda.stream(s_txt):
                runtime.stream_wait(s_txt, main, ev)
                _, t = self.bert(input_ids, text_attention_mask)                         # :485
        else:
            aud, Sa = audio_branch()
            vid, Sv = video_branch()
            _, t = self.bert(input_ids, text_attention_mask)
        av = E.EmbedAddFn.apply(hidden_states.to(dev).reshape(B * Sf, 768), pos_embed.to(dev).reshape(-1).contiguous(), self.embedding.weight)   # :474
        av = self.random_mae_encoder(av.view(B, Sf, 768), attention_mask.to(dev))       # :487 (fusion branch stays on the caller's stream)
        if runtime.multistream[0]:
            for st, ten in ((s_aud, aud), (s_vid, vid), (s_txt, t)):
                runtime.stream_wait(main, st)
                ten.record_stream(main)
        p_drop = self.dropout_p if check == "train" else 0.0
        self._drop_calls += 1
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._drop_calls) & 0xFFFFFFFFFFFFFFFF
        return E.TailFn.apply(av.reshape(B * Sf, 768), t, aud, vid, B, Sf, Sa, Sv, p_drop, seed,
                              self.rand_norm.weight, self.rand_norm.bias, self.bert_norm.weight, self.bert_norm.bias,
                              self.aud_norm.weight, self.aud_norm.bias, self.vid_norm.weight, self.vid_norm.bias,
                              self.linear1.weight, self.linear1.bias)                   # :486-499


def remap_reference_keys(state_dict):
    """Key names of a checkpoint saved with transformers 4.2x -> this package (transformers >= 5 naming), SURVEY.md §8b:
    VideoMAE `attention.attention.q_bias / v_bias` -> `query.bias / value.bias` (+ zero key.bias) inside `videomae.*`;
    weight-norm `weight_g / weight_v` -> `parametrizations.weight.original0 / original1`."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("videomae.") and k.endswith(".attention.attention.q_bias"):
            base = k[: -len("q_bias")]
            out[base + "query.bias"] = v
            out[base + "key.bias"] = torch.zeros_like(v)
        elif k.startswith("videomae.") and k.endswith(".attention.attention.v_bias"):
            out[k[: -len("v_bias")] + "value.bias"] = v
        elif k.endswith("pos_conv_embed.conv.weight_g"):
            out[k[: -len("weight_g")] + "parametrizations.weight.original0"] = v
        elif k.endswith("pos_conv_embed.conv.weight_v"):
            out[k[: -len("weight_v")] + "parametrizations.weight.original1"] = v
        elif k.endswith("embeddings.position_ids"):
            continue
        else:
            out[k] = v
    return out


def collate_batch(batch, check):
    """Batch assembly contract of reference models/tav.py:174-246 for ALREADY DECODED items
    ([{'input_ids','attention_mask'}, waveform 1-D tensor, video [16,3,224,224] (or [3,16,H,W])], label).
    Reproduces: random video token mask True w.p. 1/15 (:207-209) then flips so every row keeps the same number of False
    (the reference only equalises the total, which breaks batch>1 -- SURVEY.md 'Hard parts'); zero padding of audio with a
    0/1 mask (:225-228); labels as float tensor."""
    texts, masks, speech, vids, labels = [], [], [], [], []
    for (inp, label) in batch:
        texts.append(torch.as_tensor(inp[0]["input_ids"]).reshape(-1))
        masks.append(torch.as_tensor(inp[0]["attention_mask"]).reshape(-1).float())
        speech.append(torch.as_tensor(inp[1]).float().reshape(-1))
        v = torch.as_tensor(inp[2]).float()
        vids.append(v if v.shape[1] == 3 else v.permute(1, 0, 2, 3))
        labels.append(label)
    B = len(labels)
    ntok = (vids[0].shape[0] // 2) * (vids[0].shape[2] // 16) * (vids[0].shape[3] // 16)
    vid_mask = torch.randint(-13, 2, (B, ntok))
    vid_mask[vid_mask < 0] = 0
    vid_mask = vid_mask.bool()
    target = int(vid_mask.sum(1).max().item())
    for b in range(B):                                   # equal per-row counts (needed for reshape(B,-1,C) semantics)
        short = target - int(vid_mask[b].sum().item())
        if short > 0:
            idx = torch.where(~vid_mask[b])[0]
            vid_mask[b, idx[torch.randperm(len(idx))[:short]]] = True
    T = max(len(s) for s in speech)
    audio = torch.zeros(B, T)
    amask = torch.zeros(B, T)
    for b, s in enumerate(speech):
        audio[b, : len(s)] = s
        amask[b, : len(s)] = 1
    text = {"input_ids": torch.stack(texts).long(), "attention_mask": torch.stack(masks)}
    audio_features = {"audio_features": audio, "attention_mask": amask}
    visual = {"visual_embeds": torch.stack(vids), "attention_mask": vid_mask}
    return [text, audio_features, visual], torch.Tensor(np.array(labels))


def sample_video_mask(B, ntok, n_true=None, device="cpu", generator=None):
    """Video token mask of collate_batch (reference :206-217) built ON THE DEVICE without a host round trip: True marks the tokens the
    fusion stack sees and the video encoder drops.  The reference draws True w.p. 1/15 per token (a Binomial(ntok, 1/15) count per row)
    and then patches the TOTAL to a multiple of the batch size; rows of unequal count make `reshape(b, -1, 768)` mix utterances, so
    -- like collate_batch above -- every row gets the same count here, n_true (default round(ntok / 15): the reference's mean),
    chosen uniformly at random: the n_true smallest of ntok i.i.d. uniforms per row."""
    k = int(round(ntok / 15)) if n_true is None else int(n_true)
    r = torch.rand(B, ntok, device=device, generator=generator)
    idx = r.topk(k, dim=1, largest=False).indices
    return torch.zeros(B, ntok, dtype=torch.bool, device=device).scatter_(1, idx, True)


def collate_batch_device(batch, check, device="cuda", n_visual_true=None, generator=None):
    """collate_batch with the tensor work on `device`: items are decoded utterances as for collate_batch; every
    tensor is shipped once (non-blocking) and padding, the audio length mask (reference :225-228) and the video token mask are built
    there, sync-free: nothing in the step reads them back (PreFormer / TAVForMAE take `n_visual_true` instead of counting)."""
    texts, masks, speech, vids, labels = [], [], [], [], []
    for (inp, label) in batch:
        texts.append(torch.as_tensor(inp[0]["input_ids"]).reshape(-1))
        masks.append(torch.as_tensor(inp[0]["attention_mask"]).reshape(-1).float())
        speech.append(torch.as_tensor(inp[1]).float().reshape(-1))
        v = torch.as_tensor(inp[2]).float()
        vids.append(v if v.shape[1] == 3 else v.permute(1, 0, 2, 3))
        labels.append(float(label))
    B = len(labels)
    dev = torch.device(device)
    lens = torch.tensor([len(s) for s in speech])
    T = int(lens.max())
    audio = torch.nn.utils.rnn.pad_sequence([s.to(dev, non_blocking=True) for s in speech], batch_first=True)          # zero padding, reference :228
    amask = (torch.arange(T, device=dev)[None, :] < lens.to(dev, non_blocking=True)[:, None]).float()
    video = torch.stack([v.to(dev, non_blocking=True) for v in vids])
    ntok = (video.shape[1] // 2) * (video.shape[3] // 16) * (video.shape[4] // 16)
    vid_mask = sample_video_mask(B, ntok, n_visual_true, dev, generator)
    text = {"input_ids": torch.stack(texts).long().to(dev, non_blocking=True), "attention_mask": torch.stack(masks).to(dev, non_blocking=True)}
    return [text, {"audio_features": audio, "attention_mask": amask}, {"visual_embeds": video, "attention_mask": vid_mask}], \
        torch.tensor(labels, dtype=torch.float32).to(dev, non_blocking=True)

