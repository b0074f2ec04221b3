"""Drop-in for the reference's SingleModels/train_model/text_training.py: `get_statistics` (:14-29) and the per-batch step of
`not_grad_accum` (:31-56: backward, clip_grad_norm_, AdamW step, cosine warm restarts), on the fused clip+AdamW of this package."""
import torch

from ...optim import FusedAdamW
from ...utils.global_functions import CrossEntropyLoss  # noqa: F401  (re-exported for callers that build the criterion here)


def get_statistics(input, label, model, criterion, Metric, check="train", epoch=None):
    dev = next(model.parameters()).device
    label = label.to(dev)
    mask = input["attention_mask"].to(dev)
    input_id = input["input_ids"].squeeze(1).to(dev)                      # reference :20
    output = model(input_id, mask, check)
    if Metric is not None:
        Metric.update_metrics(torch.argmax(output, dim=1), label.long())
    if criterion is None:
        return None
    return criterion(output, label, epoch=epoch if epoch is not None else 1)


class TextTrainStep:
    """forward + loss + backward + clip_grad_norm_ + AdamW.step for the text-only classifier (reference :36-43)."""

    def __init__(self, model, criterion, lr=1e-6, weight_decay=1e-4, clip=1.0):
        self.model, self.criterion, self.clip = model, criterion, clip
        self.opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=weight_decay)

    def __call__(self, input, label, check="train", epoch=0):
        loss = get_statistics(input, label, self.model, self.criterion, None, check=check, epoch=epoch)
        loss.backward()
        norm = self.opt.clip_and_step(self.clip)
        self.opt.zero_grad()
        return loss, norm
