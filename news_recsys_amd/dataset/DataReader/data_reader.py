"""Text feature-file reader: compatibility input of the path.

Mirror of the reference's `src/dataset/DataReader/data_reader.py:7-115`: same constructor
`(config_path, feature_file_path)`, same line format ("name:val name:val ...\\tlabel [label ...]",
written by FeaturesGenerator/feature_extractor_base.py:199-204,247), same per-sample dict
(sparse -> int, dense -> float, array -> LongTensor[L] padded with 0 + FloatTensor mask[L], 'label'
-> FloatTensor), same errors.  It is a per-sample Python parser -- correct but ~5 orders of
magnitude slower than the embedding kernels consume; `columnar.py` converts such a file once into
a columnar binary layout that the GPU path can be fed from."""
from __future__ import annotations

import os
from typing import Dict, List, Union

import torch
from torch.utils.data import Dataset

from ...config import load_config


def parse_feature_line(raw_line: str, idx: int, sparse, dense, array, array_max_length) -> Dict[str, object]:
    """One line -> raw python values {name: int | float | list[int]} + 'label': list[float]."""
    try:
        feature_part, label_part = raw_line.split("\t")
    except ValueError:
        raise ValueError(f"Line {idx} format error: missing tab separator between features and labels.")
    out: Dict[str, object] = {}
    for item in feature_part.split(" "):
        if ":" not in item:
            raise ValueError(f"Feature item format error: '{item}' does not contain ':' separator.")
        name, val = item.split(":", 1)
        if name in sparse:
            out[name] = int(val)
        elif name in dense:
            out[name] = float(val)
        elif name in array:
            if array_max_length.get(name) is None:
                raise ValueError(f"Max length for array feature '{name}' missing in config.")
            out[name] = [int(x) for x in val.split(",")] if val else []
    out["label"] = [float(l) for l in label_part.strip().split(" ")]
    return out


class DataReader(Dataset):
    def __init__(self, config_path: str, feature_file_path: str = None):
        config = load_config(config_path)
        self.sparse_features = set(config.features.sparse_feature_names or [])
        self.dense_features = set(config.features.dense_feature_names or [])
        self.array_features = set(config.features.array_feature_names or [])
        self.array_max_length = dict(config.features.array_max_length or {})
        self.data_path = feature_file_path
        if self.data_path is None:
            raise ValueError("Data file path must be provided.")
        if not os.path.exists(self.data_path):
            raise FileNotFoundError(f"Data file not found: {self.data_path}")
        with open(self.data_path, "r", encoding="utf-8") as f:
            self.data_lines = [line.strip() for line in f if line.strip()]

    def __len__(self) -> int:
        return len(self.data_lines)

    def __getitem__(self, idx: int) -> Dict[str, Union[torch.Tensor, int, float]]:
        raw = parse_feature_line(self.data_lines[idx], idx, self.sparse_features, self.dense_features,
                                 self.array_features, self.array_max_length)
        ret: Dict[str, Union[torch.Tensor, int, float]] = {}
        for name, val in raw.items():
            if name == "label":
                continue
            if name in self.array_features:
                max_len = self.array_max_length[name]
                ids: List[int] = list(val)
                n = len(ids)
                if n < max_len:
                    mask = [1.0] * n + [0.0] * (max_len - n)
                    ids = ids + [0] * (max_len - n)
                else:
                    ids = ids[:max_len]
                    mask = [1.0] * max_len
                ret[name] = torch.tensor(ids, dtype=torch.long)
                ret[f"{name}_mask"] = torch.tensor(mask, dtype=torch.float32)
            else:
                ret[name] = val
        ret["label"] = torch.tensor(raw["label"], dtype=torch.float32)
        return ret
