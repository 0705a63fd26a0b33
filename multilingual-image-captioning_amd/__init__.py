"""MI355X-native CLIP-Vision + mBART-50 captioning hot path (train step + greedy/beam generate).

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed/RCCL).  All arithmetic:
hand-written HIP for gfx950 in `csrc/`, reached through the C ABI of `include/mic_hip.h`.
Importable as `mic_amd` (see the `mic_amd.py` shim at the repo root: the directory name has a hyphen).
"""
from . import _lib  # noqa: F401
from .configuration_clip_vision_mbart import CLIPVisionMBartConfig  # noqa: F401
from .modeling_clip_vision_mbart import FlaxCLIPVisionMBartForConditionalGeneration  # noqa: F401
from .image_transform import Transform  # noqa: F401
from .train import Trainer, create_learning_rate_fn, loss_rows, packed_rows, shift_tokens_right  # noqa: F401

__all__ = ["CLIPVisionMBartConfig", "FlaxCLIPVisionMBartForConditionalGeneration", "Trainer", "Transform", "create_learning_rate_fn",
           "loss_rows", "packed_rows", "shift_tokens_right"]
