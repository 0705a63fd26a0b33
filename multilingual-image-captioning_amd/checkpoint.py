"""Checkpoint wire format of the reference: `flax_model.msgpack` (flax.serialization.to_bytes/from_bytes:
msgpack with ndarray ext-type 1 = (shape, dtype name, raw bytes)) + `config.json`
(`modeling_clip_vision_utils.py:323-333, 441-445`).  flax is not installed here; the format is restated from its
published serializer and round-trip tested (tests/test_checkpoint.py)."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import msgpack
import numpy as np

_EXT_NDARRAY = 1


def _np_dtype(name: str):
    if name == "bfloat16":
        return None
    return np.dtype(name)


def _encode(obj):
    if isinstance(obj, np.ndarray) or np.isscalar(obj) and not isinstance(obj, (int, float, bool, str, bytes)):
        a = np.asarray(obj)
        payload = msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True)
        return msgpack.ExtType(_EXT_NDARRAY, payload)
    raise TypeError(f"cannot serialise {type(obj)}")


def _decode(code, data):
    if code == _EXT_NDARRAY:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        if dtype_name == "bfloat16":
            u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
            return u.view(np.float32).reshape(shape)
        return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()
    return msgpack.ExtType(code, data)


def save_flax_msgpack(path: str, tree) -> None:
    with open(path, "wb") as f:
        f.write(msgpack.packb(tree, default=_encode, use_bin_type=True, strict_types=False))


def load_flax_msgpack(path: str):
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_decode, raw=False, strict_map_key=False)


def load_component(path: str, config=None):
    """A CLIP-vision or mBART checkpoint directory (`config.json` + `flax_model.msgpack`) -> object with `.config`
    (dict) and `.params` (nested numpy tree), the two things `from_clip_vision_mbart_pretrained` reads (modeling:740-770)."""
    if not os.path.isdir(path):
        raise EnvironmentError(f"{path}: only local checkpoint directories can be loaded (no network in this build)")
    if config is None:
        with open(os.path.join(path, "config.json")) as f:
            config = json.load(f)
            config = config.get("vision_config", config) if "vision_config" in config and "hidden_size" not in config else config
    return SimpleNamespace(config=config, params=load_flax_msgpack(os.path.join(path, "flax_model.msgpack")))
