"""Synthetic MELD-shaped batches: the output contract of the reference's collate_batch
(models/tav.py:174-246) without decoding any file.  Seeded torch.Generator, seed = 1234 + rank."""
import torch


def make_batch(cfg, batch_size, *, seed=1234, s_text=128, t_audio=80000, n_visual_true=104, device="cpu", text_only=False, with_video=True):
    """Returns ([text, audio, visual] dicts exactly as collate_batch yields them, labels float [B]).  text_only / with_video=False skip
    the (large) modalities the single- and dual-modal entrypoints do not read (their dicts come back as None)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    tc, vc = cfg["text"], cfg["video"]
    B = batch_size
    ids = torch.randint(3, tc["vocab"], (B, s_text), generator=g)
    npad = s_text // 4
    text_mask = torch.ones(B, s_text)
    if npad:
        ids[:, s_text - npad:] = tc["pad_id"]
        text_mask[:, s_text - npad:] = 0
    if text_only:
        labels = torch.randint(0, 7, (B,), generator=g).float()
        return [{"input_ids": ids.to(device), "attention_mask": text_mask.to(device)}, None, None], labels.to(device)
    audio = torch.randn(B, t_audio, generator=g) * 0.1
    audio_mask = torch.ones(B, t_audio)
    cut = int(0.8 * t_audio)
    audio[0, cut:] = 0
    audio_mask[0, cut:] = 0
    if not with_video:
        labels = torch.randint(0, 7, (B,), generator=g).float()
        return [{"input_ids": ids.to(device), "attention_mask": text_mask.to(device)},
                {"audio_features": audio.to(device), "attention_mask": audio_mask.to(device)}, None], labels.to(device)
    video = torch.randn(B, vc["frames"], 3, vc["image"], vc["image"], generator=g)
    ntok = (vc["image"] // vc["patch"]) ** 2 * (vc["frames"] // vc["tubelet"])
    vmask = torch.zeros(B, ntok, dtype=torch.bool)
    for b in range(B):
        vmask[b, torch.randperm(ntok, generator=g)[:n_visual_true]] = True      # exactly n True per row
    labels = torch.randint(0, 7, (B,), generator=g).float()
    text = {"input_ids": ids.to(device), "attention_mask": text_mask.to(device)}
    audio_d = {"audio_features": audio.to(device), "attention_mask": audio_mask.to(device)}
    visual = {"visual_embeds": video.to(device), "attention_mask": vmask.to(device)}
    return [text, audio_d, visual], labels.to(device)


def seeded_init_(module, seed=0):
    """Re-draw every parameter from a seeded generator with the module's own init scale (std of the current values),
    so the oracle and the product can be given bit-identical random weights without shipping blobs."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            std = float(p.detach().float().std().item()) if p.numel() > 1 else 0.0
            mean = float(p.detach().float().mean().item())
            if std == 0.0:                       # constant tensors (biases = 0, LayerNorm weight = 1): keep, add small noise
                std = 0.02
            new = torch.randn(p.shape, generator=g) * std + mean
            if name.endswith("parametrizations.weight.original0"):
                new = new.abs() + 0.5
            p.copy_(new.to(p.device))
    return module
