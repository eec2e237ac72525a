"""`lightning` is not installed in the MI355X image.  The model classes subclass the real
LightningModule when it is importable and otherwise this minimal stand-in, which provides the
members the reference's modules touch (save_hyperparameters / log / device / trainer / logger,
BaseModel/base_model.py:63,181-218; sort/deep/model.py:50-51)."""
from __future__ import annotations

import torch
import torch.nn as nn

try:  # pragma: no cover - depends on the environment
    import lightning as L
    LightningModule = L.LightningModule
    seed_everything = L.seed_everything
    HAVE_LIGHTNING = True
except Exception:  # ImportError or a broken install
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._hparams = {}
            self._logged = {}
            self.trainer = None
            self.logger = None
            self.current_epoch = 0

        def save_hyperparameters(self, *args, **kwargs):
            for a in args:
                if isinstance(a, dict):
                    self._hparams.update(a)

        @property
        def hparams(self):
            return self._hparams

        def log(self, name, value, **kwargs):
            self._logged[name] = value

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    def seed_everything(seed: int, workers: bool = False) -> int:
        import random
        import numpy as np
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        return seed
