"""Drop-in for the reference's tav_nn.py (entrypoint of the TAV path): same main() / runModel() / prepare_dataloader()
shape and the same CLI flags (utils/global_functions.arg_parse).  The reference reads a dataset pickle + mp4/wav files and a
wandb sweep config (tav_nn.py:116-188); offline there is neither, so utterances come from synthetic.make_batch with the
contract of collate_batch."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import config as C
from . import runtime, synthetic
from .models.tav import PreFormer, TAVForMAE
from .train_model.tav_train import evaluate_tav, train_tav_network
from .utils.global_functions import CrossEntropyLoss, Metrics, NewCrossEntropyLoss, arg_parse


class SyntheticTAVBatches(Dataset):
    """Each item is one already-collated batch (the reference's DataLoader yields ([text, audio, visual], labels))."""

    def __init__(self, cfg, n_utterances, batch_size, seed, s_text=70, t_audio=80000):
        self.cfg, self.n, self.bs, self.seed, self.s_text, self.t_audio = cfg, max(1, n_utterances // batch_size), batch_size, seed, s_text, t_audio

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        nv = 104 if self.cfg["video"]["image"] == 224 else 4
        return synthetic.make_batch(self.cfg, self.bs, seed=self.seed + i, s_text=self.s_text, t_audio=self.t_audio, n_visual_true=nv)


def prepare_dataloader(df, batch_size, label_task, epoch_switch, pin_memory=True, num_workers=0, check="train"):
    """`df` is a SyntheticTAVBatches here (reference: a pandas frame of file paths, tav_nn.py:28-57)."""
    return DataLoader(df, batch_size=None, shuffle=False, num_workers=num_workers, pin_memory=pin_memory)


def runModel(accelerator, df_train, df_val, df_test, param_dict, model_param):
    device = accelerator
    if param_dict["loss"] == "CrossEntropy":
        criterion = CrossEntropyLoss()
    else:
        criterion = NewCrossEntropyLoss(class_weights=param_dict["weights"].to(device), epoch_switch=param_dict["epoch_switch"])
    Metric = Metrics(num_classes=model_param["output_dim"], id2label=param_dict["id2label"], rank=device)
    bs, lt, es = param_dict["batch_size"], param_dict["label_task"], param_dict["epoch_switch"]
    dl_train = prepare_dataloader(df_train, bs, lt, es, check="train")
    dl_val = prepare_dataloader(df_val, bs, lt, es, check="val")
    dl_test = prepare_dataloader(df_test, bs, lt, es, check="val")
    model = TAVForMAE(model_param).to(device)
    PREFormer = PreFormer().to(device)
    model, PREFormer = train_tav_network(model, PREFormer, dl_train, dl_val, criterion, param_dict["lr"], param_dict["epoch"], param_dict["weight_decay"],
                                         param_dict["T_max"], Metric, param_dict["patience"], param_dict["clip"], es, None)
    evaluate_tav(model, PREFormer, dl_test, Metric)
    return model, PREFormer


def main(argv=None):
    args = arg_parse("TAV", argv)
    np.random.seed(args.seed)
    torch.random.manual_seed(args.seed)
    C.set_default_preset(args.preset)
    runtime.set_precision(args.dtype)
    cfg = C.default_config()
    weights = torch.linspace(0.6, 0.95, args.output_dim)            # reference: 1 - class frequency (tav_nn.py:171)
    id2label = {i: f"class{i}" for i in range(args.output_dim)}
    param_dict = {"epoch": args.epoch, "patience": args.patience, "lr": args.learning_rate, "clip": args.clip, "batch_size": args.batch_size,
                  "weight_decay": args.weight_decay, "model": args.model, "T_max": args.T_max, "seed": args.seed, "label_task": args.label_task,
                  "mask": args.mask, "loss": args.loss, "beta": args.beta, "epoch_switch": args.epoch_switch, "weights": weights,
                  "label2id": {v: k for k, v in id2label.items()}, "id2label": id2label}
    model_param = {"output_dim": args.output_dim, "dropout": args.dropout, "early_div": args.early_div, "num_layers": args.num_layers,
                   "learn_PosEmbeddings": args.learn_PosEmbeddings}
    small = cfg["video"]["image"] != 224
    mk = lambda n, seed: SyntheticTAVBatches(cfg, n, args.batch_size, seed, s_text=16 if small else 70, t_audio=8000 if small else 80000)   # noqa: E731
    print(f" in main \n param_dict = { {k: v for k, v in param_dict.items() if k != 'weights'} } \n model_param = {model_param} \n synthetic utterances = {args.synthetic}")
    return runModel("cuda", mk(args.synthetic, 1000), mk(max(args.batch_size, args.synthetic // 4), 2000), mk(max(args.batch_size, args.synthetic // 4), 3000),
                    param_dict, model_param)


if __name__ == "__main__":
    main()
