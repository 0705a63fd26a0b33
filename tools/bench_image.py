"""Throughput of mic_image_transform: 256 uint8 images 480x640 -> [256,224,224,3] fp32 (one collate batch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import Transform

dev = torch.device("cuda:0")
imgs = [torch.randint(0, 256, (3, 480, 640), dtype=torch.uint8, device=dev) for _ in range(256)]
tf = Transform(224, device=dev)
for _ in range(3):
    out = tf.batch(imgs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    out = tf.batch(imgs)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
src = 256 * 3 * 480 * 640
dst = out.numel() * 4
# bytes the cropped window actually needs: 224x224 outputs * scale^2 source pixels (480/224)^2 * 3 B, plus the output
need = 256 * (224 * 224 * (480 / 224) ** 2 * 3) + dst
print(f"{ms * 1e3:.1f} us per 256-image batch (incl. host descriptor setup) -> {256 / ms * 1e3:.0f} images/s; "
      f"algorithmic {need / 1e6:.0f} MB -> {need / ms / 1e9:.2f} TB/s (source total {src / 1e6:.0f} MB, output {dst / 1e6:.0f} MB)")
