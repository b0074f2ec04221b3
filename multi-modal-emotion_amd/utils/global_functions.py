"""Drop-in for the parts of the reference's utils/global_functions.py that the TAV path touches:
NewCrossEntropyLoss (:51-83), MySampler (:21-49), arg_parse (:260-297, same flag names) and a dependency-free Metrics
(the reference wraps torchmetrics, :114-188, which is outside the hot path and not installed here)."""
import os
from argparse import ArgumentParser

import torch
from torch import nn
from torch.utils.data.sampler import Sampler

from .. import engine as E


class MySampler(Sampler):
    """reference :21-49: multinomial sampling on epochs where epoch % epoch_switch == 0, else sequential."""

    def __init__(self, weights, num_samples, replacement=True, epoch=0, epoch_switch=2):
        if not isinstance(num_samples, int) or isinstance(num_samples, bool) or num_samples <= 0:
            raise ValueError("num_samples should be a positive integer value, but got num_samples={}".format(num_samples))
        if not isinstance(replacement, bool):
            raise ValueError("replacement should be a boolean value, but got replacement={}".format(replacement))
        self.weights = torch.as_tensor(weights, dtype=torch.double)
        self.num_samples, self.replacement, self.epoch, self.epoch_switch = num_samples, replacement, epoch, epoch_switch

    def __iter__(self):
        if self.epoch % self.epoch_switch == 0:
            self.epoch += 1
            yield from iter(torch.multinomial(self.weights, self.num_samples, self.replacement).tolist())
        else:
            self.epoch += 1
            yield from iter(range(self.num_samples))

    def __len__(self):
        return self.num_samples


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss(weight=None|w), mean reduction, on libtavhip (tav_nn.py:85)."""

    def __init__(self, weight=None):
        super().__init__()
        self.weight = weight

    def forward(self, logits, target, epoch=None):
        w = self.weight.to(logits.device, torch.float32) if self.weight is not None else None
        return E.CrossEntropyFn.apply(logits, target.to(logits.device).long(), w)


class NewCrossEntropyLoss(nn.Module):
    """reference :51-83: unweighted CE when epoch % epoch_switch == 0, class-weighted CE otherwise."""

    def __init__(self, class_weights, epoch_switch=2):
        super().__init__()
        self.class_weights = class_weights
        self.epoch_switch = epoch_switch
        self.weightedCEL = CrossEntropyLoss(weight=class_weights)
        self.normalCEL = CrossEntropyLoss()

    def forward(self, logits, target, epoch):
        if epoch % self.epoch_switch == 0:
            return self.normalCEL(logits, target)
        return self.weightedCEL(logits, target)


def _run_names():
    """(project, sweep_id, run name) of the active wandb run, as the reference's checkpoint path uses them (:203-226); fixed names offline."""
    try:
        import wandb
        run = getattr(wandb, "run", None)
        if run is not None:
            return str(run.project), str(run.sweep_id), str(run.name)
    except Exception:
        pass
    return "TAV_Train", "local", "run"


def checkpoint_file(path):
    return os.path.join(path, *_run_names(), "best.pt")


def save_model(model, PREFormer, optimizer, criterion, scheduler, epoch, step, path, log_val):
    """reference :199-236: best.pt = {'epoch', 'step', 'model_state_dict', 'optimizer_state_dict', 'loss', 'scheduler', ['PREFormer']}
    under <path>/<project>/<sweep>/<run>/ (path: no trailing slash)."""
    f = checkpoint_file(path)
    os.makedirs(os.path.dirname(f), exist_ok=True)
    torch.save({"epoch": epoch, "step": step, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "loss": criterion.state_dict() if criterion is not None else {}, "scheduler": scheduler.state_dict() if scheduler is not None else {},
                **({"PREFormer": PREFormer.state_dict()} if PREFormer is not None else {})}, f)
    return f


def load_model(model, PREFormer, optimizer, criterion, path, remap=True):
    """reference :238-258.  `optimizer` is refreshed in place (the reference builds a new AdamW over the same parameters and loads the saved state
    into it).  remap=True passes the state dicts through models.tav.remap_reference_keys, so a checkpoint written by the reference itself
    (transformers 4.2x key names) loads as well."""
    from ..models.tav import remap_reference_keys
    f = checkpoint_file(path)
    checkpoint = torch.load(f, map_location="cpu", weights_only=False)
    print(f"Current best model is on epoch {checkpoint['epoch']}, and step {checkpoint['step']} on path {f}", flush=True)
    fix = remap_reference_keys if remap else (lambda d: d)
    model.load_state_dict(fix(checkpoint["model_state_dict"]))
    if PREFormer is not None:
        PREFormer.load_state_dict(fix(checkpoint["PREFormer"]))
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    if criterion is not None and checkpoint.get("loss") is not None:
        criterion.load_state_dict(checkpoint["loss"])
    from .. import engine
    engine.bump_weight_epoch()               # parameters changed: cached low-precision operand copies must refresh
    return model, PREFormer, optimizer, criterion


class Metrics:
    """Confusion-matrix metrics (accuracy, macro/weighted F1, recall, precision) without torchmetrics."""

    def __init__(self, num_classes, id2label=None, rank="cuda", **_):
        self.num_classes = num_classes
        self.id2label = id2label or {i: str(i) for i in range(num_classes)}
        self.cm = torch.zeros(num_classes, num_classes, dtype=torch.long)

    def update_metrics(self, preds, target):
        idx = (target.reshape(-1).long().cpu() * self.num_classes + preds.reshape(-1).long().cpu())
        self.cm += torch.bincount(idx, minlength=self.num_classes ** 2).view(self.num_classes, self.num_classes)

    def reset_metrics(self):
        self.cm.zero_()

    def compute_scores(self, name):
        cm = self.cm.double()
        tp, sup, pred = cm.diag(), cm.sum(1), cm.sum(0)
        rec = tp / sup.clamp(min=1)
        prec = tp / pred.clamp(min=1)
        f1 = 2 * prec * rec / (prec + rec).clamp(min=1e-12)
        acc = rec
        mk = lambda tag, v: {f"{name}/{tag}/{self.id2label[i]}": v[i].item() for i in range(self.num_classes)}   # noqa: E731
        w = sup / sup.sum().clamp(min=1)
        return (mk("multiAcc", acc), mk("multiF1", f1), mk("multiRec", rec), mk("multiPrec", prec), acc.mean().item(), f1.mean().item(),
                (f1 * w).sum().item(), rec.mean().item(), prec.mean().item(), self.cm.clone())


def arg_parse(description, argv=None):
    """Same flag names / defaults as the reference (:260-297); extra flags select the MI355X runtime options."""
    parser = ArgumentParser(description=f" Run experiments on {description} ")
    parser.add_argument("--learning_rate", "-l", default=0.000001, type=float)
    parser.add_argument("--epoch", "-e", default=3, type=int)
    parser.add_argument("--batch_size", "-b", default=1, type=int)
    parser.add_argument("--weight_decay", "-w", default=0.0001, type=float)
    parser.add_argument("--clip", "-c", default=1.0, type=float)
    parser.add_argument("--epoch_switch", "-es", default=2, type=int)
    parser.add_argument("--patience", "-p", default=10.0, type=float)
    parser.add_argument("--T_max", "-t", default=2, type=int)
    parser.add_argument("--mask", "-ma", default=False, type=bool)
    parser.add_argument("--loss", "-ls", default="NewCrossEntropy", type=str)
    parser.add_argument("--beta", "-beta", default=1, type=float)
    parser.add_argument("--seed", "-s", default=32, type=int)
    parser.add_argument("--dataset", "-d", default="../data/text_audio_video_emotion_data")
    parser.add_argument("--model", "-m", default="MAE_encoder")
    parser.add_argument("--label_task", "-lt", default="emotion")
    parser.add_argument("--input_dim", "-z", default=2, type=int)
    parser.add_argument("--output_dim", "-y", default=7, type=int)
    parser.add_argument("--lstm_layers", "-ll", default=1, type=int)
    parser.add_argument("--hidden_layers", "-o", default="32,32", type=str)
    parser.add_argument("--early_div", "-ed", default=False, type=bool)
    parser.add_argument("--dropout", "-dr", default=0.5, type=float)
    parser.add_argument("--num_layers", "-nl", default=12, type=int)
    parser.add_argument("--learn_PosEmbeddings", "-lpe", default=True, type=bool)
    # MI355X build additions
    parser.add_argument("--preset", default="A", help="model geometry: A (reference names) | B (BASELINE trio) | *-tiny")
    parser.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--synthetic", default=64, type=int, help="number of synthetic utterances per split (no dataset files are read)")
    return parser.parse_args(argv)
