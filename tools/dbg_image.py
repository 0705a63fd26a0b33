import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mic_amd
from mic_amd import Transform
from oracle import image_ref as I
rng = np.random.default_rng(3)
tf = Transform(224, device="cuda:0")
for (H, W) in ((224, 224), (300, 451), (512, 333)):
    img = rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
    got = tf(torch.from_numpy(img)).cpu().numpy()
    ref = I.transform(img, 224)
    d = np.abs(got - ref)
    idx = np.argwhere(d > 0)
    print(H, W, "mismatch", len(idx), "max", d.max(), idx[:5].tolist())
    if len(idx):
        c, y, x = idx[0]
        print(" got", got[c, y, x], "ref", ref[c, y, x], "delta levels", (got[c,y,x]-ref[c,y,x]) * 255 * I.CLIP_STD[c])
