#!/usr/bin/env python3
"""Minimal end-to-end loop on the MI355X path, shaped like the reference's `make train model=deep`
(src/model/sort/deep/train.py) but without Lightning: text feature files -> columnar device loader ->
Deep (fused HIP embedding path + MLP head) -> AdamW + CosinDecayLR -> on-device validation metrics
(val_log.log in the reference's format).

    python examples/train_deep_minimal.py -c <train_cf_deep.yaml> [--epochs 2]

The YAML is the reference's schema; `paths.out_basedir/extractored_feature/{train,dev}_features.txt` must exist
(the reference's `make fe` output)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd.dataset.DataReader.pl_dataloader import MINDDataModule   # noqa: E402
from news_recsys_amd.lightning_shim import seed_everything                     # noqa: E402
from news_recsys_amd.model.sort.deep.model import Deep                         # noqa: E402


def main(config: str, epochs: int, device: str = "cuda:0"):
    seed_everything(42)
    dm = MINDDataModule(config)
    model = Deep(config).to(device)
    model.setup("fit")
    opt_cfg = model.configure_optimizers()
    opt, sched = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    train = dm.train_loader_columnar(device)
    val = dm.val_loader_columnar(device)
    history = []
    for epoch in range(epochs):
        model.train()
        model.current_epoch = epoch
        total, n = 0.0, 0
        for i, batch in enumerate(train):
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, i)
            loss.backward()
            opt.step()
            sched.step()
            total, n = total + loss.item(), n + 1
        model.eval()
        with torch.no_grad():
            for i, batch in enumerate(val):
                model.validation_step(batch, i)
        res = model.on_validation_epoch_end()
        history.append((total / max(n, 1), res))
        print(f"epoch {epoch}: train_loss {total / max(n, 1):.4f}  val AUC {res['Overall']['AUC']:.4f}  GAUC {res['Overall']['GAUC']:.4f}")
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--epochs", type=int, default=2)
    a = ap.parse_args()
    main(a.config, a.epochs)
