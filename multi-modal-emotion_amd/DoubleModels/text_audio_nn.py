"""Entrypoint of the text+audio dual path (reference DoubleModels/text_audio_nn.py, which does not run: SURVEY.md §8f row 2).
Same main()/runModel() shape as tav_nn.py; synthetic batches of 10 s audio (160000 samples -> 499 audio frames) and 128 text tokens
(BASELINE.json configs[3])."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import config as C
from .. import runtime, synthetic
from ..utils.global_functions import CrossEntropyLoss, Metrics, NewCrossEntropyLoss, arg_parse
from .models.text_audio import BertAudioClassifier
from .train_model.text_audio_training import TextAudioTrainStep, get_statistics


class SyntheticTextAudioBatches(Dataset):
    def __init__(self, cfg, n_utterances, batch_size, seed, s_text=128, t_audio=160000):
        self.cfg, self.n, self.bs, self.seed, self.s_text, self.t_audio = cfg, max(1, n_utterances // batch_size), batch_size, seed, s_text, t_audio

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        (tx, au, _), lab = synthetic.make_batch(self.cfg, self.bs, seed=self.seed + i, s_text=self.s_text, t_audio=self.t_audio, with_video=False)
        return [tx, au], lab


def prepare_dataloader(df, batch_size, label_task, epoch_switch, pin_memory=True, num_workers=0, check="train"):
    return DataLoader(df, batch_size=None, shuffle=False, num_workers=num_workers, pin_memory=pin_memory)


def runModel(accelerator, df_train, df_val, df_test, param_dict, model_param):
    device = accelerator
    if param_dict["loss"] == "CrossEntropy":
        criterion = CrossEntropyLoss()
    else:
        criterion = NewCrossEntropyLoss(class_weights=param_dict["weights"].to(device), epoch_switch=param_dict["epoch_switch"])
    Metric = Metrics(num_classes=model_param["output_dim"], id2label=param_dict["id2label"], rank=device)
    model = BertAudioClassifier(model_param).to(device)
    step = TextAudioTrainStep(model, criterion, lr=param_dict["lr"], weight_decay=param_dict["weight_decay"], clip=param_dict["clip"])
    for epoch in range(param_dict["epoch"]):
        total = 0.0
        for inp, lab in prepare_dataloader(df_train, param_dict["batch_size"], None, None):
            loss, _ = step(inp, lab, check="train", epoch=epoch)
            total += loss.item()
        print(f"epoch {epoch}: train loss {total / max(1, len(df_train)):.4f}", flush=True)
        with torch.no_grad():
            vl = [get_statistics(i, l, model, criterion, Metric, check="val", epoch=epoch).item() for i, l in prepare_dataloader(df_val, 0, None, None)]
        print(f"epoch {epoch}: val loss {sum(vl) / max(1, len(vl)):.4f}", flush=True)
    return model


def main(argv=None):
    args = arg_parse("TextAudio", argv)
    np.random.seed(args.seed)
    torch.random.manual_seed(args.seed)
    C.set_default_preset(args.preset)
    runtime.set_precision(args.dtype)
    cfg = C.default_config()
    id2label = {i: f"class{i}" for i in range(args.output_dim)}
    param_dict = {"epoch": args.epoch, "patience": args.patience, "lr": args.learning_rate, "clip": args.clip, "batch_size": args.batch_size,
                  "weight_decay": args.weight_decay, "loss": args.loss, "epoch_switch": args.epoch_switch,
                  "weights": torch.linspace(0.6, 0.95, args.output_dim), "id2label": id2label}
    model_param = {"output_dim": args.output_dim, "dropout": args.dropout}
    small = cfg["video"]["image"] != 224
    mk = lambda n, seed: SyntheticTextAudioBatches(cfg, n, args.batch_size, seed, s_text=16 if small else 128, t_audio=8000 if small else 160000)   # noqa: E731
    return runModel("cuda", mk(args.synthetic, 1000), mk(max(args.batch_size, args.synthetic // 4), 2000), None, param_dict, model_param)


if __name__ == "__main__":
    main()
