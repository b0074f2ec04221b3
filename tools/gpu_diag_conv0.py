"""Diagnostic (GPU box): per-tensor gradient errors of the audio front-end at full audio length, fp32 policy, against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import tav_amd
from oracle import tav_oracle as O
from tav_amd import config as C, engine as E, runtime, synthetic
from tav_amd.models.tav import PreFormer, TAVForMAE

ARGS = dict(output_dim=7, dropout=0.5, learn_PosEmbeddings=True, num_layers=12)
preset = sys.argv[1] if len(sys.argv) > 1 else "B"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
cfg = C.preset(preset)
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 1
IMG = int(sys.argv[4]) if len(sys.argv) > 4 else 32
for k in ("text", "audio", "video", "fusion"):
    cfg[k]["layers"] = min(NL, cfg[k]["layers"])
cfg["video"]["image"] = IMG
print(f"### preset {preset} T {T} layers {NL} image {IMG}")
torch.manual_seed(0)
pre, model = PreFormer(cfg), TAVForMAE(ARGS, cfg)
synthetic.seeded_init_(pre, 1); synthetic.seeded_init_(model, 2)
(tx, au, vi), lab = synthetic.make_batch(cfg, 2, t_audio=T, n_visual_true=4 if IMG == 32 else 104)
batch = dict(input_ids=tx["input_ids"], text_mask=tx["attention_mask"], audio_features=au["audio_features"], audio_mask=au["attention_mask"],
             video_embeds=vi["visual_embeds"], visual_mask=vi["attention_mask"])
sdp = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pre.state_dict().items()}
sdm = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
o_logits, o_loss = O.tav_step(sdm, sdp, cfg, batch, lab.long())
o_loss.backward()
pre.cuda(); model.cuda()
for ms in (True, False):
    runtime.multistream[0] = ms
    for policy in ("fp32",):
        runtime.set_precision(policy)
        runs = []
        for rep in range(2):
            for p in list(pre.parameters()) + list(model.parameters()):
                p.grad = None
            tav, emb, amask = pre(input_ids=batch["input_ids"], audio_features=batch["audio_features"], video_embeds=batch["video_embeds"], text_mask=batch["text_mask"],
                                  audio_mask=batch["audio_mask"], visual_mask=batch["visual_mask"], device="cuda", train=False)
            logits = model(batch["input_ids"], batch["text_mask"], batch["audio_features"], batch["video_embeds"], batch["visual_mask"], tav, emb, amask, batch_size=2, check="val")
            loss = E.CrossEntropyFn.apply(logits, lab.long().cuda(), None)
            loss.backward()
            torch.cuda.synchronize()
            runs.append({("model." + k): p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
            runs[-1].update({("pre." + k): p.grad.clone() for k, p in pre.named_parameters() if p.grad is not None})
        same = all(torch.equal(runs[0][k], runs[1][k]) for k in runs[0])
        print(f"== multistream {ms} policy {policy}: bitwise repeatable {same}")
        gmax = max(v.grad.abs().max().item() for v in list(sdp.values()) + list(sdm.values()) if getattr(v, 'grad', None) is not None)
        for k in sorted(runs[0]):
            tag, name = k.split(".", 1)
            og = (sdm if tag == "model" else sdp)[name].grad
            g = runs[0][k].cpu()
            if ((g - og).abs().max() / (og.abs().max() + 1e-3 * gmax)).item() < 1e-3:
                continue
            print(f"      product |g|max {g.abs().max().item():.3e} mean|g| {g.abs().mean().item():.3e}; ref mean|g| {og.abs().mean().item():.3e}; cos {torch.nn.functional.cosine_similarity(g.flatten(), og.flatten(), dim=0).item():.4f}; g[0,0,:5] {g.flatten()[:5].tolist()} ref {og.flatten()[:5].tolist()}")
            print(f"   {k:75s} |ref|max {og.abs().max().item():.3e} abs err {(g - og).abs().max().item():.3e} rel {((g - og).abs().max() / og.abs().max()).item():.2e}  (gmax {gmax:.2e})")
