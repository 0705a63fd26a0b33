"""Generates tests/golden/image_small.npz: uint8 images and the result of the reference Transform's arithmetic as torch
itself computes it in this container (torch.nn.functional.interpolate bicubic, align_corners=False, no antialias — the
tensor path of torchvision Resize — then round/clamp, center crop, /255, normalise).  Run: python tests/golden/make_golden_image.py"""
import os

import numpy as np
import torch
import torch.nn.functional as F

MEAN = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1)
STD = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1)


def torch_transform(img_u8: torch.Tensor, S: int):
    _, H, W = img_u8.shape
    nh, nw = (int(S * H / W), S) if W <= H else (S, int(S * W / H))
    r = F.interpolate(img_u8[None].float(), size=[nh, nw], mode="bicubic", align_corners=False)[0]
    r = torch.round(r).clamp(0, 255).to(torch.uint8)
    top, left = int(round((nh - S) / 2.0)), int(round((nw - S) / 2.0))
    r = r[:, top: top + S, left: left + S]
    return r, ((r.float() / 255) - MEAN) / STD


if __name__ == "__main__":
    g = torch.Generator().manual_seed(7)
    out = {}
    for i, (H, W, S) in enumerate(((37, 53, 24), (61, 40, 24), (24, 24, 24), (19, 30, 32))):
        # smooth-ish content (random low-res upsampled + noise) so the taps see gradients, not only white noise
        base = F.interpolate(torch.rand(1, 3, 5, 7, generator=g) * 255, size=[H, W], mode="bilinear", align_corners=False)[0]
        img = (base + torch.randn(3, H, W, generator=g) * 12).clamp(0, 255).to(torch.uint8)
        u8, f = torch_transform(img, S)
        out[f"img{i}"], out[f"u8_{i}"], out[f"out{i}"], out[f"S{i}"] = img.numpy(), u8.numpy(), f.numpy(), np.int64(S)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_small.npz"), **out)
    print({k: v.shape for k, v in out.items()})
