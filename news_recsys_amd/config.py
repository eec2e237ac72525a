"""YAML config loading for the reference's `train_cf_<model>.yaml` schema without OmegaConf.

The reference loads its configs with OmegaConf (BaseModel/base_model.py:75,
DataReader/data_reader.py:28) but only uses plain-YAML features (no interpolation), so
yaml.safe_load + an attribute dict reproduces `cfg.features.sparse_feature_names`,
`cfg.get('paths', {})`, `self.train_hparams.lr`, ... exactly.  OmegaConf is used when present."""
from __future__ import annotations

import os
from typing import Any

import yaml


class AttrDict(dict):
    """dict with attribute access (the subset of DictConfig behaviour the reference relies on)."""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v


def _wrap(x: Any) -> Any:
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def to_container(cfg: Any) -> Any:
    """OmegaConf.to_container(cfg, resolve=True) equivalent."""
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [to_container(v) for v in cfg]
    return cfg


def load_config(config_path: str) -> AttrDict:
    if not os.path.exists(config_path):
        raise FileNotFoundError(f"Config file not found: {config_path}")   # base_model.py:71-72
    with open(config_path, "r", encoding="utf-8") as f:
        data = yaml.safe_load(f) or {}
    return _wrap(data)
