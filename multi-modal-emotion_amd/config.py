"""Model geometry presets (SURVEY.md §8 'Presets').

'A' = what the reference's hard-coded from_pretrained names resolve to (models/tav.py:257-263,438-457):
      distilroberta (6 L) + wav2vec2-large-xlsr (24 L, 1024-d, stable LN, 'layer' feature norm) + videomae-base.
'B' = the BASELINE.json trio: bert-base + wav2vec2-base + videomae-base  (the benchmarked configuration).
Suffix '-tiny' keeps every width but cuts depth (and the video frame to 32x32) for tests.
"""
import copy

_CONV = dict(conv_dim=[512] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2])
_VIDEO = dict(layers=12, hidden=768, heads=12, inter=3072, frames=16, image=224, patch=16, tubelet=2, eps=1e-12)
_FUSION = dict(layers=12, hidden=768, heads=12, inter=3072, eps=1e-12)

_PRESETS = {
    "A": dict(
        text=dict(kind="roberta", layers=6, hidden=768, heads=12, inter=3072, vocab=50265, max_pos=514, type_vocab=1, pad_id=1, eps=1e-5),
        audio=dict(layers=24, hidden=1024, heads=16, inter=4096, feat_norm="layer", stable_ln=True, conv_bias=True, pos_k=128, pos_groups=16,
                   eps=1e-5, **_CONV),
        video=_VIDEO, fusion=_FUSION, output_dim=7),
    "B": dict(
        text=dict(kind="bert", layers=12, hidden=768, heads=12, inter=3072, vocab=30522, max_pos=512, type_vocab=2, pad_id=0, eps=1e-12),
        audio=dict(layers=12, hidden=768, heads=12, inter=3072, feat_norm="group", stable_ln=False, conv_bias=False, pos_k=128, pos_groups=16,
                   eps=1e-5, **_CONV),
        video=_VIDEO, fusion=_FUSION, output_dim=7),
}

# BASELINE config 5: preset B with videomae-large (24 L, 1024 / 16 heads / 4096) on 32 frames (3136 tubelet tokens, 209 fed to the fusion stack).
# The reference hard-codes 768-wide video features (models/tav.py:446 LayerNorm(768), :372 concat with 768-wide text/audio tokens), so a
# 1024-wide video encoder needs the same bridge the reference already uses for its 1024-wide audio encoder (:363 wav_2_768, :457/478
# wav_2_768_2): a Linear(1024, 768) after the patch embedding in PreFormer ("vid_2_768") and after the encoder in TAVForMAE ("vid_2_768_2").
_PRESETS["B5"] = dict(_PRESETS["B"], video=dict(layers=24, hidden=1024, heads=16, inter=4096, frames=32, image=224, patch=16, tubelet=2, eps=1e-12))

_current = ["A"]


def preset(name="B"):
    base = name.split("-")[0]
    cfg = copy.deepcopy(_PRESETS[base])
    if name.endswith("-tiny"):
        cfg["text"]["layers"] = 2
        cfg["text"]["vocab"] = 1000
        cfg["audio"]["layers"] = 2
        cfg["video"]["layers"] = 2
        cfg["video"]["image"] = 32
        cfg["fusion"]["layers"] = 2
    return cfg


def set_default_preset(name_or_cfg):
    """What `PreFormer()` / `TAVForMAE(args)` build when constructed with the reference's signatures."""
    _current[0] = name_or_cfg


def default_config():
    c = _current[0]
    return copy.deepcopy(c) if isinstance(c, dict) else preset(c)
