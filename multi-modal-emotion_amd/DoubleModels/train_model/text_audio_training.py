"""Step of the text+audio dual path: forward + loss + backward + clip_grad_norm_ + AdamW (same shape as
train_model/tav_train.py:56-65 of the reference, which is the loop every *_nn.py entrypoint repeats)."""
import torch

from ...optim import FusedAdamW


def get_statistics(input, label, model, criterion, Metric, check="train", epoch=None):
    text, audio = input[0], input[1]
    dev = next(model.parameters()).device
    label = label.to(dev)
    output = model(text["input_ids"], text["attention_mask"], audio["audio_features"], check=check)
    if Metric is not None:
        Metric.update_metrics(torch.argmax(output, dim=1), label.long())
    if criterion is None:
        return None
    return criterion(output, label, epoch=epoch if epoch is not None else 1)


class TextAudioTrainStep:
    def __init__(self, model, criterion, lr=1e-6, weight_decay=1e-4, clip=1.0):
        self.model, self.criterion, self.clip = model, criterion, clip
        self.opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=weight_decay)

    def __call__(self, input, label, check="train", epoch=0):
        loss = get_statistics(input, label, self.model, self.criterion, None, check=check, epoch=epoch)
        loss.backward()
        norm = self.opt.clip_and_step(self.clip)
        self.opt.zero_grad()
        return loss, norm
