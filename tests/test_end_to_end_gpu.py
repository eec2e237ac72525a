"""End-to-end on the GPU: a synthetic MIND-shaped feature file with a learnable signal goes through the
columnar loader, the Deep model on the HIP path, AdamW + the cosine schedule and the on-device validation;
the loss must fall, validation AUC must beat chance, and val_log.log must carry the reference's block."""
import os

import numpy as np
import pytest
import yaml

from tests.conftest import CONFIGS, ROOT

pytestmark = pytest.mark.gpu


def test_minimal_training_loop_learns(tmp_path):
    import importlib.util
    rng = np.random.default_rng(0)
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, "cf_deep_small.yaml")))
    cfg["paths"]["out_basedir"] = str(tmp_path)
    cfg["dataset"] = {"batch_size": 256, "num_workers": 0, "pin_memory": False}
    cfg["train_hparams"]["lr"] = 5e-3
    cfg["train_hparams"]["lr_milestones"] = [2000, 5000]      # keep the base lr for this short run
    fdir = tmp_path / "extractored_feature"
    os.makedirs(fdir)
    os.makedirs(tmp_path / "preprocess")
    (tmp_path / "preprocess" / "train_user_ids.json").write_text("[" + ",".join(map(str, range(1, 60))) + "]")
    sizes = cfg["embeddings"]["embedding_table_size"]
    liked = {u: set(rng.choice(np.arange(1, sizes["category"]), 4, replace=False).tolist()) for u in range(1, sizes["user_id"])}

    def write(path, n):
        with open(path, "w") as f:
            for _ in range(n):
                u = int(rng.integers(1, sizes["user_id"]))
                cat = int(rng.integers(1, sizes["category"]))
                y = int(rng.random() < (0.85 if cat in liked[u] else 0.1))       # the signal: user x category affinity
                f.write(f"user_id:{u} item_id:{int(rng.integers(1, sizes['item_id']))} category:{cat} "
                        f"subcategory:{int(rng.integers(1, sizes['subcategory']))} "
                        f"user_click_category:{int(rng.integers(1, sizes['user_click_category']))}\t{y}\n")

    write(fdir / "train_features.txt", 6000)
    write(fdir / "dev_features.txt", 1500)
    cpath = tmp_path / "cfg.yaml"
    cpath.write_text(yaml.safe_dump(cfg))
    spec = importlib.util.spec_from_file_location("train_deep_minimal", os.path.join(ROOT, "examples", "train_deep_minimal.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.main(str(cpath), epochs=6)
    assert hist[-1][0] < hist[0][0] - 0.02                                   # training loss falls
    aucs = [h[1]["Overall"]["AUC"] for h in hist]
    assert aucs[-1] > 0.55 and aucs[-1] > aucs[0] + 0.03                    # and validation AUC climbs above chance
    txt = (tmp_path / "val_log.log").read_text()
    assert txt.count("Validation Results") == 6 and "Cold Start Users" in txt
