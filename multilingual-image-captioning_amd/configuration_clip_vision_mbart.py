"""`CLIPVisionMBartConfig` — host-side mirror of the reference's composite config
(`models/flax_clip_vision_mbart/configuration_clip_vision_mbart.py:10-51`): `.mbart_config`,
`.clip_vision_config`, `is_encoder_decoder = True`, `from_clip_vision_mbart_configs`, `to_dict`.

The sub-configs are plain attribute bags (the reference used transformers' MBartConfig / CLIPVisionConfig; a
transformers config object or a dict is accepted and copied attribute by attribute), with the hub values of
`facebook/mbart-large-50` and `openai/clip-vit-base-patch32` as defaults.

Build-only switches for the three numerically relevant facts of the un-vendored Flax dependency that cannot be
verified offline: `mbart_config.gelu_variant` ("tanh" | "erf") and `mbart_config.decoder_ln_eps`.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict

MBART_DEFAULTS: Dict[str, Any] = dict(
    vocab_size=250054, d_model=1024, decoder_layers=12, decoder_attention_heads=16, decoder_ffn_dim=4096,
    max_position_embeddings=1024, activation_function="gelu", dropout=0.1, attention_dropout=0.0, activation_dropout=0.0,
    scale_embedding=True, init_std=0.02, pad_token_id=1, bos_token_id=0, eos_token_id=2, decoder_start_token_id=2,
    forced_eos_token_id=2, forced_bos_token_id=None, num_beams=5, max_length=200, min_length=0, early_stopping=True,
    length_penalty=1.0, do_sample=False, no_repeat_ngram_size=0, top_k=50, top_p=1.0, temperature=1.0,
    tie_word_embeddings=True,
    # build-only [UNVERIFIED-3P] switches
    gelu_variant="tanh", decoder_ln_eps=1e-6,
)
CLIP_DEFAULTS: Dict[str, Any] = dict(
    hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, image_size=224, patch_size=32,
    hidden_act="quick_gelu", layer_norm_eps=1e-5, attention_dropout=0.0, dropout=0.0,
)


class _SubConfig:
    _defaults: Dict[str, Any] = {}

    def __init__(self, **kwargs):
        for k, v in self._defaults.items():
            self.__dict__[k] = copy.deepcopy(v)
        for k, v in kwargs.items():
            self.__dict__[k] = v

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if _jsonable(v)}


def _jsonable(v) -> bool:
    try:
        json.dumps(v)
        return True
    except TypeError:
        return False


class MBartSubConfig(_SubConfig):
    _defaults = MBART_DEFAULTS

    @property
    def hidden_size(self):  # MBartConfig attribute_map: hidden_size -> d_model (used at modeling:54)
        return self.__dict__["d_model"]


class CLIPVisionSubConfig(_SubConfig):
    _defaults = CLIP_DEFAULTS


def _as_dict(cfg) -> Dict[str, Any]:
    if cfg is None:
        return {}
    if isinstance(cfg, dict):
        return dict(cfg)
    if hasattr(cfg, "to_dict"):
        return dict(cfg.to_dict())
    return dict(vars(cfg))


class CLIPVisionMBartConfig:
    model_type = "clip-vision-mbart"
    is_composition = True

    def __init__(self, **kwargs):
        if "mbart_config" not in kwargs:
            raise ValueError("`mbart_config` can not be `None`.")  # cfg:18-19
        if "clip_vision_config" not in kwargs:
            raise ValueError("`clip_vision_config` can not be `None`.")  # cfg:21-22
        mb, cv = _as_dict(kwargs.pop("mbart_config")), _as_dict(kwargs.pop("clip_vision_config"))
        self.mbart_config = MBartSubConfig(**{k: v for k, v in mb.items() if _jsonable(v)})
        self.clip_vision_config = CLIPVisionSubConfig(**{k: v for k, v in cv.items() if _jsonable(v)})
        self.is_encoder_decoder = True  # cfg:31
        self.tie_word_embeddings = kwargs.pop("tie_word_embeddings", True)
        self.output_attentions = False
        self.output_hidden_states = False
        self.return_dict = True
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def from_clip_vision_mbart_configs(cls, clip_vision_config, mbart_config, **kwargs):  # cfg:33-44
        return cls(clip_vision_config=_as_dict(clip_vision_config), mbart_config=_as_dict(mbart_config), **kwargs)

    def to_dict(self) -> Dict[str, Any]:  # cfg:46-51
        out = {k: v for k, v in self.__dict__.items() if k not in ("mbart_config", "clip_vision_config") and _jsonable(v)}
        out["clip_vision_config"] = self.clip_vision_config.to_dict()
        out["mbart_config"] = self.mbart_config.to_dict()
        out["model_type"] = self.model_type
        return out

    def save_pretrained(self, save_directory: str) -> None:
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, path: str, **kwargs) -> "CLIPVisionMBartConfig":
        with open(os.path.join(path, "config.json") if os.path.isdir(path) else path) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.update(kwargs)
        return cls(**d)
