"""MINDDataModule mirror (reference: src/dataset/DataReader/pl_dataloader.py:10-96): everything comes
from the YAML (`paths.out_basedir`, `dataset.batch_size / num_workers / pin_memory`), files are
`<out_basedir>/extractored_feature/{train,dev}_features.txt`.

Two feeding modes: `train_dataloader()` / `val_dataloader()` return torch DataLoaders over the text
DataReader exactly like the reference; `train_loader_columnar(device)` / `val_loader_columnar(device)`
convert the text file once (cached next to it) and return ColumnarLoaders that deliver device batches."""
from __future__ import annotations

import os
from typing import Optional

from torch.utils.data import DataLoader

from ...config import load_config
from .columnar import ColumnarDataset, ColumnarLoader, convert_features_txt
from .data_reader import DataReader

try:  # pragma: no cover
    import lightning.pytorch as pl
    _Base = pl.LightningDataModule
except Exception:
    class _Base:  # minimal stand-in when Lightning is absent
        def __init__(self):
            pass

        def save_hyperparameters(self, *a, **k):
            pass


class MINDDataModule(_Base):
    def __init__(self, config_path: str):
        super().__init__()
        self.save_hyperparameters()
        self.config_path = config_path
        self.conf = load_config(config_path)
        self.out_basedir = self.conf.paths.out_basedir
        data_cfg = self.conf.get("dataset", {}) or {}
        self.batch_size = data_cfg.get("batch_size", 32)
        self.num_workers = data_cfg.get("num_workers", 4)
        self.pin_memory = data_cfg.get("pin_memory", True)
        feature_dir = os.path.join(self.out_basedir, "extractored_feature")
        self.train_file_path = os.path.join(feature_dir, "train_features.txt")
        self.val_file_path = os.path.join(feature_dir, "dev_features.txt")
        self.train_dataset = None
        self.val_dataset = None

    def setup(self, stage: Optional[str] = None):
        if stage == "fit" or stage is None:
            if not os.path.exists(self.train_file_path):
                raise FileNotFoundError(f"Training feature file missing: {self.train_file_path}")
            if not os.path.exists(self.val_file_path):
                raise FileNotFoundError(f"Validation feature file missing: {self.val_file_path}")
            self.train_dataset = DataReader(config_path=self.config_path, feature_file_path=self.train_file_path)
            self.val_dataset = DataReader(config_path=self.config_path, feature_file_path=self.val_file_path)

    def _loader(self, ds, shuffle):
        return DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle, num_workers=self.num_workers,
                          pin_memory=self.pin_memory, persistent_workers=self.num_workers > 0, drop_last=False)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False)

    # ---- columnar fast path
    def _columnar(self, txt_path: str) -> ColumnarDataset:
        col_dir = txt_path + ".columnar"
        meta = os.path.join(col_dir, "meta.json")
        if not os.path.exists(meta) or os.path.getmtime(meta) < os.path.getmtime(txt_path):
            convert_features_txt(self.config_path, txt_path, col_dir)
        return ColumnarDataset(col_dir)

    def train_loader_columnar(self, device, seed: int = 0, resident: bool = False) -> ColumnarLoader:
        """resident=True keeps the whole split in GPU memory and gathers batches on the device."""
        return ColumnarLoader(self._columnar(self.train_file_path), self.batch_size, device, shuffle=True, seed=seed,
                              resident=resident)

    def val_loader_columnar(self, device, resident: bool = False) -> ColumnarLoader:
        return ColumnarLoader(self._columnar(self.val_file_path), self.batch_size, device, shuffle=False, resident=resident)
