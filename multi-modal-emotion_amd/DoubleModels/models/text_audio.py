"""Text + audio dual-modal classifier (BASELINE.json configs[3]: 10 s audio, long-sequence audio attention).
The reference's DoubleModels/models/text_audio.py does not parse (two `__init__`s on a class without a base, use-before-assignment in
`forward`, :70-112), so the path is defined as `TAVForMAE.forward` (reference models/tav.py:473-499) minus the video and fusion
branches, with the same sub-module names and state_dict keys:
    logits = linear1(dropout(cat[bert_norm(pooled_text), aud_norm(mean_t(wav_2_768_2(wav2vec2(audio))))])),   linear1: 1536 -> output_dim
The text and audio encoders run on two streams like the tri-modal model's branches."""
import torch
from torch import nn

from ... import config as C
from ... import engine as E
from ... import runtime
from ...encoders import AudioEncoder, TextEncoder


class BertAudioClassifier(nn.Module):
    def __init__(self, args, dropout=0.5, config=None):
        super().__init__()
        cfg = config if config is not None else C.default_config()
        self.cfg = cfg
        self.output_dim = args["output_dim"]
        self.dropout_p = float(args.get("dropout", dropout))
        self.bert = TextEncoder(cfg["text"])
        self.bert_norm = nn.LayerNorm(768)
        self.wav2vec2 = AudioEncoder(cfg["audio"])
        self.wav_2_768_2 = nn.Linear(cfg["audio"]["hidden"], 768)
        nn.init.xavier_normal_(self.wav_2_768_2.weight)
        self.aud_norm = nn.LayerNorm(768)
        self.linear1 = nn.Linear(768 * 2, self.output_dim)
        self._drop_calls = 0

    def forward(self, input_ids, text_attention_mask, audio_features, check="train"):
        dev = self.linear1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("BertAudioClassifier runs on libtavhip (GPU) only; there is no CPU fallback")
        ectx = runtime.ctx()
        B = input_ids.shape[0]
        input_ids, text_attention_mask = input_ids.to(dev), text_attention_mask.to(dev)
        audio_features = audio_features.to(dev, torch.float32)

        def audio_branch():
            a, a_lp, sa = self.wav2vec2(audio_features)
            return E.LinearFn.apply(a, a_lp if not ectx.pol.f32 else None, self.wav_2_768_2.weight, self.wav_2_768_2.bias, None, ectx, True), sa

        main = torch.cuda.current_stream()
        if runtime.multistream[0]:
            ev = torch.cuda.Event()
            ev.record(main)
            s_aud = runtime.branch_streams(3)[0]
            runtime.share_with(s_aud, audio_features)
            with torch.cuda.stream(s_aud):
                s_aud.wait_event(ev)
                aud, Sa = audio_branch()
            _, t = self.bert(input_ids, text_attention_mask)
            main.wait_stream(s_aud)
            aud.record_stream(main)
        else:
            aud, Sa = audio_branch()
            _, t = self.bert(input_ids, text_attention_mask)
        feat = E.PoolNormCatFn.apply(B, (0, Sa), t, aud, self.bert_norm.weight, self.bert_norm.bias, self.aud_norm.weight, self.aud_norm.bias)
        p = self.dropout_p if check == "train" else 0.0
        self._drop_calls += 1
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._drop_calls) & 0xFFFFFFFFFFFFFFFF
        return E.HeadFn.apply(feat, p, seed, self.linear1.weight, self.linear1.bias)
