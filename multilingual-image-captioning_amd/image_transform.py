"""The reference's `Transform` module (main.py:165-179; evaluation.py:35-54) on the GPU: bicubic Resize([size]) ->
CenterCrop(size) -> ConvertImageDtype(float) -> Normalize(CLIP mean/std), plus the batched NHWC form `collate_fn` builds
(`torch.stack(...).permute(0, 2, 3, 1)`, main.py:494; evaluation.py:60).  One HIP kernel per batch (`mic_image_transform`);
decoding the image file to uint8 (torchvision `read_image`, main.py:220) stays on the host."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import ops

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class Transform:
    def __init__(self, image_size: int, device=None, mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD):
        self.image_size = int(image_size)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.mean, self.std = tuple(mean), tuple(std)

    def _dev(self, x) -> torch.Tensor:
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        if t.dtype != torch.uint8:
            raise TypeError("Transform: expects uint8 images (what read_image returns); got " + str(t.dtype))
        return t.to(self.device).contiguous()

    def __call__(self, x) -> torch.Tensor:
        """x: uint8 [3,H,W] -> float32 [3,size,size] on the device (Transform.forward)."""
        out = torch.empty((1, 3, self.image_size, self.image_size), dtype=torch.float32, device=self.device)
        ops.image_transform([self._dev(x)], self.image_size, self.mean, self.std, out, chw_out=True)
        return out[0]

    forward = __call__

    def batch(self, images) -> torch.Tensor:
        """list of uint8 [3,H,W] (or [H,W,3]) images of any sizes -> float32 [B,size,size,3] NHWC `pixel_values`."""
        out = torch.empty((len(images), self.image_size, self.image_size, 3), dtype=torch.float32, device=self.device)
        ops.image_transform([self._dev(im) for im in images], self.image_size, self.mean, self.std, out)
        return out
