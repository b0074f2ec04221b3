"""Drop-in for the reference's SingleModels/models/text.py `BertClassifier` (:41-69): text encoder -> pooler -> dropout (only when
check == "train") -> Linear(768, output_dim), executed by libtavhip (BASELINE.json configs[0]).
The reference builds the encoder with `BertModel.from_pretrained('j-hartmann/emotion-english-distilroberta-base')` (:48), a network
fetch; here the geometry comes from the preset's "text" block (config.py) and weights are loaded by the caller."""
import torch
from torch import nn

from ... import config as C
from ... import engine as E
from ... import runtime
from ...encoders import TextEncoder


class BertClassifier(nn.Module):
    def __init__(self, args, dropout=0.5, config=None):
        super().__init__()
        cfg = config if config is not None else C.default_config()
        self.cfg = cfg
        self.dropout_p = float(args["dropout"])          # reference :44 (the `dropout` keyword is shadowed there as well)
        self.output_dim = args["output_dim"]
        self.bert = TextEncoder(cfg["text"])
        self.linear = nn.Linear(768, self.output_dim)
        self._drop_calls = 0

    def forward(self, input_id, mask, check):
        dev = self.linear.weight.device
        if dev.type != "cuda":
            raise RuntimeError("BertClassifier runs on libtavhip (GPU) only; there is no CPU fallback")
        _, x = self.bert(input_id.to(dev), mask.to(dev))                                    # :58 (pooled output)
        p = self.dropout_p if check == "train" else 0.0                                      # :61-62
        self._drop_calls += 1
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._drop_calls) & 0xFFFFFFFFFFFFFFFF
        return E.HeadFn.apply(x, p, seed, self.linear.weight, self.linear.bias)              # :65
