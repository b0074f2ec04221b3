"""Drop-in for the reference's train_model/tav_train.py: same function names and argument order
(get_statistics :15-48, not_grad_accum :52-83, validate :121-130, one_epoch :133-144, train_tav_network :147-164,
evaluate_tav :166-167), running on libtavhip.  Differences, all forced by the MI355X design:
  * no `assert video.shape == (1,16,3,224,224)` (reference :32 pins batch 1);
  * AdamW + clip_grad_norm_ are the fused multi-tensor kernels of optim.py (identical update rule);
  * wandb / checkpoint paths are optional (no network, no cluster paths);
  * with torch.distributed initialised, gradients are averaged over ranks by ddp.BucketedAllReduce overlapped with backward.
"""
import math
import os

import torch

from .. import ddp as tav_ddp
from ..optim import FusedAdamW
from ..utils.global_functions import checkpoint_file, load_model, save_model

try:                                    # optional, as in the reference's environment
    import wandb
except Exception:                       # pragma: no cover
    wandb = None

PATIENCE_ITER = 0


class CosineWarmRestarts:
    """torch CosineAnnealingWarmRestarts(T_0=T_max, T_mult=1, eta_min=0).step(epoch_float) for FusedAdamW."""

    def __init__(self, optimizer, T_0):
        self.opt, self.T_0, self.base_lr = optimizer, T_0, optimizer.lr

    def step(self, epoch):
        t_cur = epoch % self.T_0
        self._t_cur, self._last_epoch = t_cur, epoch
        self.opt.lr = self.base_lr * (1 + math.cos(math.pi * t_cur / self.T_0)) / 2

    def get_last_lr(self):
        return [self.opt.lr]

    def state_dict(self):
        """Keys of torch.optim.lr_scheduler.CosineAnnealingWarmRestarts.state_dict() that define the schedule."""
        return {"T_0": self.T_0, "T_i": self.T_0, "T_mult": 1, "eta_min": 0, "T_cur": getattr(self, "_t_cur", 0), "base_lrs": [self.base_lr],
                "last_epoch": getattr(self, "_last_epoch", 0), "_last_lr": [self.opt.lr]}

    def load_state_dict(self, sd):
        self.T_0, self.base_lr = sd["T_0"], sd["base_lrs"][0]
        self._t_cur, self._last_epoch = sd.get("T_cur", 0), sd.get("last_epoch", 0)
        if sd.get("_last_lr"):
            self.opt.lr = sd["_last_lr"][0]


def get_statistics(input, label, model, PREFormer, criterion, Metric, check="train", epoch=None, n_visual_true=None):
    device = "cuda"
    batch_size = len(label)
    text, audio_features, video_embeds = input[0], input[1], input[2]
    text_input_ids, text_attention_mask = text["input_ids"], text["attention_mask"]
    audio_input_ids, audio_attention_mask = audio_features["audio_features"], audio_features["attention_mask"]
    video_input_ids, video_attention_mask = video_embeds["visual_embeds"], video_embeds["attention_mask"]
    tav, tav_embed, attention_mask = PREFormer(input_ids=text_input_ids, audio_features=audio_input_ids, video_embeds=video_input_ids,
                                               text_mask=text_attention_mask, audio_mask=audio_attention_mask, visual_mask=video_attention_mask,
                                               device=device, train=True if check == "train" else False, n_visual_true=n_visual_true)
    output = model(input_ids=text_input_ids.to(device), text_attention_mask=text_attention_mask.to(device), audio_features=audio_input_ids.to(device),
                   video_embeds=video_input_ids.to(device), visual_mask=video_attention_mask.to(device), hidden_states=tav.to(device),
                   pos_embed=tav_embed.to(device), attention_mask=attention_mask.to(device), batch_size=batch_size, check=check,
                   n_visual_true=n_visual_true)
    label = label.to(device).long()          # (the reference's .type(torch.LongTensor) would bounce through the host)
    if Metric is not None:
        Metric.update_metrics(torch.argmax(output, dim=1), label.long())
    batch_loss = None
    if criterion is not None:
        batch_loss = criterion(output, label, epoch=epoch if epoch is not None else 1)
    return batch_loss


class TrainStep:
    """One optimisation step of reference :56-65: get_statistics -> backward -> (all-reduce) -> clip_grad_norm_ -> AdamW."""

    def __init__(self, model, PREFormer, criterion, lr=1e-6, weight_decay=1e-4, clip=1.0, bucket_mb=48.0, reduce_dtype=None, zero_grad_like_torch_1_10=False):
        self.model, self.pre, self.criterion, self.clip = model, PREFormer, criterion, clip
        # what `model.zero_grad()` (reference :64-65, :98-99, :104-105) does to .grad: torch 1.10 -- the version the reference pins,
        # requirements.txt:100 -- zero-FILLS; torch >= 2.0 sets None.  Matters only for grad_accum's second step (see there).
        self.zero_to_none = not zero_grad_like_torch_1_10
        self.params = [p for p in model.parameters() if p.requires_grad] + [p for p in PREFormer.parameters() if p.requires_grad]
        self.opt = FusedAdamW(self.params, lr=lr, weight_decay=weight_decay)
        self.reducer = None
        alone_ok = os.environ.get("TAV_DDP_SINGLE_RANK", "0") == "1"         # exercise the RCCL path with one rank (tests)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and (torch.distributed.get_world_size() > 1 or alone_ok):
            self.reducer = tav_ddp.BucketedAllReduce(self.params, bucket_mb=bucket_mb, reduce_dtype=reduce_dtype, single_rank_ok=alone_ok)

    def forward_loss(self, input, label, check="train", epoch=0, n_visual_true=None):
        return get_statistics(input, label, self.model, self.pre, self.criterion, None, check=check, epoch=epoch, n_visual_true=n_visual_true)

    def forward_backward(self, input, label, check="train", epoch=0, n_visual_true=None):
        loss = self.forward_loss(input, label, check, epoch, n_visual_true)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        return loss

    def update(self, clip=True):
        norm = self.opt.clip_and_step(self.clip if clip else None)
        self.opt.zero_grad(set_to_none=self.zero_to_none)
        return norm

    def __call__(self, input, label, check="train", epoch=0, n_visual_true=None):
        loss = self.forward_backward(input, label, check, epoch, n_visual_true)
        return loss, self.update()


def validate(val_dataloader, model, PREFormer, criterion, Metric, name="val"):
    total = 0.0
    with torch.no_grad():
        for val_input, val_label in val_dataloader:
            loss = get_statistics(val_input, val_label, model, PREFormer, criterion, Metric, name, epoch=None)
            if criterion is not None:
                total += loss.item()
        log(Metric, total / len(val_dataloader) if criterion is not None else 0, name)
    return total / len(val_dataloader)


def _save_if_better(val_loss, prev_val_loss, model, PREFormer, stepper, criterion, scheduler, epoch, batch_idx, path, log_val, patience):
    """reference :72-82 / :108-118: keep the best validation loss, save best.pt on improvement, count patience otherwise."""
    global PATIENCE_ITER
    if val_loss < prev_val_loss:
        PATIENCE_ITER = 0
        print(f"we have seen loss decrease the previous best and we are updating our best loss val to {val_loss}")
        if path is not None:
            save_model(model, PREFormer, stepper.opt, criterion, scheduler, epoch, batch_idx, path, log_val)
        return val_loss, False
    PATIENCE_ITER += 1
    print(f"we have seen loss increase for {PATIENCE_ITER} steps and validation loss is {val_loss}, and previous best validtion loss is {prev_val_loss}")
    return prev_val_loss, PATIENCE_ITER == patience


def not_grad_accum(epoch, train_dataloader, val_dataloader, model, PREFormer, criterion, stepper, scheduler, patience, Metric, prev_val_loss, log_val, path=None):
    """reference :52-83: one optimisation step per batch."""
    iters = len(train_dataloader)
    total_loss_train = 0.0
    for batch_idx, (train_input, train_label) in enumerate(train_dataloader):
        loss = get_statistics(train_input, train_label, model, PREFormer, criterion, Metric, check="train", epoch=epoch)
        total_loss_train += loss.item()
        loss.backward()
        if stepper.reducer is not None:
            stepper.reducer.finish()
        stepper.update()
        scheduler.step(epoch + batch_idx / iters)
        if ((batch_idx + 1) % log_val == 0) or (batch_idx + 1 == iters):
            log(Metric, total_loss_train / iters, "train")
            val_loss = validate(val_dataloader, model, PREFormer, criterion, Metric, name="val")
            prev_val_loss, stop = _save_if_better(val_loss, prev_val_loss, model, PREFormer, stepper, criterion, scheduler, epoch, batch_idx, path, log_val, patience)
            if stop:
                break
    return prev_val_loss


def grad_accum(epoch, train_dataloader, val_dataloader, model, PREFormer, criterion, stepper, scheduler, patience, Metric, prev_val_loss, log_val, path=None):
    """reference :87-119, the dialogue-level variant used on epochs with epoch % epoch_switch != 0.  Kept with its quirk: the loss is
    divided by the dialogue length (`dataset.retGradAccum(i)` -> (accum_iter, accum_sum)) but the optimizer still steps -- and the
    gradients are zeroed -- after EVERY batch (:96-100), so the extra, unclipped `optimizer.step()` at a dialogue end (:102-106) runs on zeroed
    gradients.  What that step does depends on the torch version behind `model.zero_grad()`:
      * torch >= 2.0 (`set_to_none=True`): every .grad is None, AdamW skips every parameter -- nothing changes but the scheduler call
        (the default here: `TrainStep(zero_grad_like_torch_1_10=False)`);
      * torch 1.10, the version the reference pins (README_and_Requirements/requirements.txt:100; `set_to_none=False`): the gradients are
        zero TENSORS, so AdamW still runs -- weights decay by (1 - lr * wd), both moments shrink by their betas, the step counter advances and
        the parameters move along the remaining momentum (`TrainStep(zero_grad_like_torch_1_10=True)` /
        `train_tav_network(..., zero_grad_like_torch_1_10=True)`).
    Both readings are tested (tests/test_abi_and_host.py on the call sequence, tests/test_model_gpu.py against torch.optim.AdamW)."""
    iters = len(train_dataloader)
    total_loss_train = 0.0
    for batch_idx, (train_input, train_label) in enumerate(train_dataloader):
        accum_iter, accum_sum = train_dataloader.dataset.retGradAccum(i=batch_idx)
        loss = get_statistics(train_input, train_label, model, PREFormer, criterion, Metric, check="train", epoch=epoch) / accum_iter
        total_loss_train += loss.item()
        loss.backward()
        if stepper.reducer is not None:
            stepper.reducer.finish()
        stepper.update()
        scheduler.step(epoch + batch_idx / iters)
        if ((batch_idx + 1) % accum_sum == 0) or (batch_idx + 1 == iters):
            stepper.update(clip=False)           # reference :103: optimizer.step() without clip_grad_norm_; no-op or a momentum / decay step (docstring)
            scheduler.step(epoch + batch_idx / iters)
        if ((batch_idx + 1) % log_val == 0) or (batch_idx + 1 == iters):
            log(Metric, total_loss_train / iters, "train")
            val_loss = validate(val_dataloader, model, PREFormer, criterion, Metric, name="val")
            prev_val_loss, stop = _save_if_better(val_loss, prev_val_loss, model, PREFormer, stepper, criterion, scheduler, epoch, batch_idx, path, log_val, patience)
            if stop:
                break
    return prev_val_loss


def one_epoch(epoch, train_dataloader, val_dataloader, model, PREFormer, criterion, stepper, scheduler, epoch_switch, patience, Metric, prev_val_loss,
              path=None, log_val=2400):
    """reference :133-144: alternate the two loops by epoch parity, then reload the best checkpoint of the run (:143)."""
    loop = not_grad_accum if (epoch % epoch_switch == 0 or not hasattr(train_dataloader.dataset, "retGradAccum")) else grad_accum
    prev_val_loss = loop(epoch, train_dataloader, val_dataloader, model, PREFormer, criterion, stepper, scheduler, patience, Metric, prev_val_loss, log_val, path)
    if path is not None and os.path.exists(checkpoint_file(path)):
        load_model(model, PREFormer, stepper.opt, criterion, path)
    return prev_val_loss


def train_tav_network(model, PREFormer, train_dataloader, val_dataloader, criterion, learning_rate, epochs, weight_decay, T_max, Metric, patience, clip,
                      epoch_switch, checkpoint=None, path=None, log_val=2400, zero_grad_like_torch_1_10=False):
    """reference :147-164.  `path` (None = keep nothing on disk) replaces the cluster path hard-coded at :137; `checkpoint` is a loaded
    best.pt dict whose optimizer / scheduler state resumes the run (:152-155)."""
    stepper = TrainStep(model, PREFormer, criterion, lr=learning_rate, weight_decay=weight_decay, clip=clip, zero_grad_like_torch_1_10=zero_grad_like_torch_1_10)
    scheduler = CosineWarmRestarts(stepper.opt, T_0=T_max)
    prev_val_loss = 100
    if checkpoint is not None:
        stepper.opt.load_state_dict(checkpoint["optimizer_state_dict"])
        sched = checkpoint.get("scheduler_state_dict", checkpoint.get("scheduler"))      # the reference saves 'scheduler' (:223) and reads 'scheduler_state_dict' (:155)
        if sched is not None:
            scheduler.load_state_dict(sched)
    for epoch_num in range(epochs):
        if wandb is not None and getattr(wandb, "run", None) is not None:
            wandb.log({"epoch": epoch_num, "learning_rate": scheduler.get_last_lr()[0]})
        stepper.opt.zero_grad()
        prev_val_loss = one_epoch(epoch_num, train_dataloader, val_dataloader, model, PREFormer, criterion, stepper, scheduler, epoch_switch, patience,
                                  Metric, prev_val_loss, path, log_val)
        if PATIENCE_ITER == patience:
            return model, PREFormer
    return model, PREFormer


def evaluate_tav(model, PREFormer, test_dataloader, Metric):
    validate(test_dataloader, model, PREFormer, None, Metric, name="test")


def log(Metric, loss, check="train"):
    if Metric is None:
        return
    multiAcc, multiF1, multiRec, multiPrec, Acc, F1Macro, F1Weighted, Rec, Prec, cm = Metric.compute_scores(f"{check}")
    d1 = {f"{check}/loss": loss, f"{check}/acc": Acc, f"{check}/precision": Prec, f"{check}/recall": Rec, f"{check}/weighted-f1-score": F1Weighted,
          f"{check}/macro-f1-score": F1Macro}
    print(f"\n in {check} \n loss = {loss:.5f} acc = {Acc:.4f} \n Confusion Matrix = {cm} \n", flush=True)
    if wandb is not None and getattr(wandb, "run", None) is not None:
        wandb.log({**d1, **multiF1, **multiRec, **multiPrec, **multiAcc})
    Metric.reset_metrics()
