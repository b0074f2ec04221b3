"""multi-modal-emotion_amd: MI355X-native TAV (text + audio + video) fusion forward/backward.

The directory name carries a hyphen (it mirrors the reference repo's name), so it is imported through the
`tav_amd` alias module at the repository root:  `import tav_amd`  ->  this package.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
