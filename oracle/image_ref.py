"""ORACLE (test infrastructure only — never imported by the product): CPU restatement of the reference's image `Transform`
(main.py:165-179; evaluation.py:35-54): Resize([S], BICUBIC) -> CenterCrop(S) -> ConvertImageDtype(float) ->
Normalize(CLIP mean, std), on one uint8 CHW image.

The arithmetic lives in torchvision (tensor path) + torch.nn.functional.interpolate(mode="bicubic", align_corners=False),
neither part of /root/reference [UNVERIFIED-3P: torchvision is not pinned by requirements.txt and not installed here].
Pinning: `tests/test_oracle_cpu.py::test_image_transform_matches_torch_interpolate` checks this restatement against
torch's own bicubic kernel in this container (the rounding to uint8 absorbs summation-order differences except at exact
.5 ties) and against the committed golden `tests/golden/image_small.npz` made from it (tests/golden/make_golden_image.py).
Every fp32 operation is written out in the order the HIP kernel uses, so the two agree bit for bit."""
import numpy as np

F32 = np.float32
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _cc1(x, A):
    return ((A + F32(2)) * x + (-(A + F32(3)))) * x * x + F32(1)


def _cc2(x, A):
    return ((A * x + (-(F32(5) * A))) * x + F32(8) * A) * x + (-(F32(4) * A))


def _taps(o, scale, n):
    """o: int array of resized-space coordinates -> (idx [len,4] clamped source indices, w [len,4] fp32 weights)."""
    A = F32(-0.75)
    real = scale * (o.astype(F32) + F32(0.5)) + F32(-0.5)
    fl = np.floor(real)
    t = (real + (-fl)).astype(F32)
    i0 = fl.astype(np.int64)
    idx = np.clip(i0[:, None] - 1 + np.arange(4)[None, :], 0, n - 1)
    w = np.stack([_cc2(t + F32(1), A), _cc1(t, A), _cc1(F32(1) + (-t), A), _cc2(F32(2) + (-t), A)], axis=1).astype(F32)
    return idx, w


def resize_dims(H, W, S):
    """torchvision Resize([S]) on a tensor: shorter side -> S, longer side int(S * long / short)."""
    if W <= H:
        return int(S * H / W), S
    return S, int(S * W / H)


def transform(img_chw_u8: np.ndarray, S: int, mean=CLIP_MEAN, std=CLIP_STD, return_u8=False):
    """uint8 [3,H,W] -> float32 [3,S,S]."""
    assert img_chw_u8.dtype == np.uint8 and img_chw_u8.ndim == 3 and img_chw_u8.shape[0] == 3
    _, H, W = img_chw_u8.shape
    nh, nw = resize_dims(H, W, S)
    top, left = int(round((nh - S) / 2.0)), int(round((nw - S) / 2.0))  # CenterCrop: Python round (half to even)
    iy, wy = _taps(np.arange(S) + top, F32(H) / F32(nh), H)
    ix, wx = _taps(np.arange(S) + left, F32(W) / F32(nw), W)
    src = img_chw_u8.astype(F32)
    out = np.empty((3, S, S), F32)
    for c in range(3):
        acc = None
        for r in range(4):
            rows = src[c][iy[:, r]]                      # [S, W]
            row = None
            for k in range(4):
                v = (rows[:, ix[:, k]] * wx[None, :, k]).astype(F32)   # [S, S]
                row = v if row is None else (row + v).astype(F32)
            v = (row * wy[:, r][:, None]).astype(F32)
            acc = v if acc is None else (acc + v).astype(F32)
        out[c] = np.clip(np.rint(acc), 0, 255)           # torch.round (half to even) + clamp to the uint8 range
    if return_u8:
        return out.astype(np.uint8)
    for c in range(3):
        out[c] = ((out[c] / F32(255)).astype(F32) + (-F32(mean[c]))).astype(F32) / F32(std[c])
    return out.astype(F32)
