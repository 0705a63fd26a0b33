"""Host side of the sampling PRNG: `jax.random.PRNGKey` / `jax.random.split` of jax 0.2.16 (threefry2x32), restated from
the published algorithm so `_sample` (gen:561, 610) walks the same key sequence as the reference.  Four 32-bit words
per decoder step — numpy on the host; the [R, V] noise itself is generated on the GPU (`mic_sample_rows`)."""
from __future__ import annotations

import numpy as np

U = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def threefry2x32(key, x0, x1):
    x0, x1 = np.array(x0, dtype=U), np.array(x1, dtype=U)
    k0, k1 = U(key[0]), U(key[1])
    ks = (k0, k1, U(k0 ^ k1 ^ U(0x1BD11BDA)))
    with np.errstate(over="ignore"):
        x0, x1 = (x0 + ks[0]).astype(U), (x1 + ks[1]).astype(U)
        for g in range(5):
            for r in _ROT[g % 2]:
                x0 = (x0 + x1).astype(U)
                x1 = ((x1 << U(r)) | (x1 >> U(32 - r))).astype(U)
                x1 = (x1 ^ x0).astype(U)
            x0 = (x0 + ks[(g + 1) % 3]).astype(U)
            x1 = (x1 + ks[(g + 2) % 3] + U(g + 1)).astype(U)
    return x0, x1


def prng_key(seed) -> np.ndarray:
    """jax.random.PRNGKey(seed) -> uint32[2] = [high word, low word]; an existing 2-word key passes through."""
    a = np.asarray(seed)
    if a.shape == (2,):
        return a.astype(U)
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([s >> 32, s & 0xFFFFFFFF], dtype=U)


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split: counters iota(2*num) split in halves, outputs concatenated, reshaped (num, 2)."""
    cnt = np.arange(2 * num, dtype=U)
    a, b = threefry2x32(key, cnt[:num], cnt[num:])
    return np.concatenate([a, b]).reshape(num, 2)
