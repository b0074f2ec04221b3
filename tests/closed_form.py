"""Closed-form (RNG-free) weights and inputs shared by the golden generator and the tests.

Every tensor is a deterministic function of its key name and shape, so fixtures hold only expected OUTPUTS."""
import zlib

import numpy as np
import torch


def _phase(name):
    h = zlib.crc32(name.encode())
    return 0.37 + (h % 1000) / 1000.0, ((h >> 10) % 1000) / 159.0


def tensor_for(name, shape, kind=None):
    n = int(np.prod(shape)) if len(shape) else 1
    b, c = _phase(name)
    base = np.sin(np.arange(n, dtype=np.float64) * b + c)
    leaf = name.rsplit(".", 1)[-1]
    if kind is None:
        if len(shape) == 1 and leaf == "weight":
            kind = "norm_weight"
        elif leaf in ("bias", "q_bias", "v_bias") or len(shape) == 1:
            kind = "bias"
        elif "original0" in name:
            kind = "wn_g"
        else:
            kind = "weight"
    if kind == "norm_weight":
        v = 1.0 + 0.1 * base
    elif kind == "bias":
        v = 0.05 * base
    elif kind == "wn_g":
        v = 1.0 + 0.2 * np.abs(base)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if "embeddings" in name and len(shape) == 2 and shape[0] > shape[1]:
            fan_in = 1.0 / 0.25          # embedding tables: O(0.5) entries
        v = base * (1.7 / np.sqrt(fan_in))
    return torch.tensor(v.reshape(shape), dtype=torch.float32)


def state_dict_for(shapes):
    """shapes: {key: shape} -> {key: tensor}"""
    return {k: tensor_for(k, tuple(s)) for k, s in shapes.items()}


def fill_module_(module):
    """Overwrite every parameter of an nn.Module with its closed-form value (keys = state_dict names)."""
    sd = module.state_dict()
    new = {k: (tensor_for(k, tuple(v.shape)) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def batch_for(B, S_text, T_audio, frames, image, vocab, pad_id, nkeep_fusion, seed_name="batch"):
    """Synthetic MELD-shaped batch in the collate_batch output contract (reference models/tav.py:174-246)."""
    ntok = (image // 16) ** 2 * (frames // 2)
    ids = (3 + (np.arange(B * S_text) * 7919 + 13) % (vocab - 3)).reshape(B, S_text)
    npad = max(1, S_text // 4)
    ids[:, S_text - npad:] = pad_id
    text_mask = np.ones((B, S_text), dtype=np.float32)
    text_mask[:, S_text - npad:] = 0
    audio = 0.1 * np.sin(np.arange(B * T_audio, dtype=np.float64) * 0.0137 + 0.3).reshape(B, T_audio) \
        + 0.05 * np.sin(np.arange(B * T_audio, dtype=np.float64) * 0.311).reshape(B, T_audio)
    audio_mask = np.ones((B, T_audio), dtype=np.float32)
    cut = int(0.8 * T_audio)
    audio[0, cut:] = 0
    audio_mask[0, cut:] = 0
    video = np.sin(np.arange(B * frames * 3 * image * image, dtype=np.float64) * 0.00731 + 1.1).reshape(B, frames, 3, image, image)
    vmask = np.zeros((B, ntok), dtype=bool)
    for b in range(B):
        idx = (np.arange(nkeep_fusion) * (ntok // nkeep_fusion) + b) % ntok
        vmask[b, idx] = True
        assert vmask[b].sum() == nkeep_fusion
    labels = (np.arange(B) * 3 + 1) % 7
    return dict(input_ids=torch.tensor(ids, dtype=torch.long), text_mask=torch.tensor(text_mask),
                audio_features=torch.tensor(audio, dtype=torch.float32), audio_mask=torch.tensor(audio_mask),
                video_embeds=torch.tensor(video, dtype=torch.float32), visual_mask=torch.tensor(vmask)), torch.tensor(labels, dtype=torch.long)
