#!/usr/bin/env python3
"""Input-pipeline throughput: synthetic MIND-shaped feature file (C1 features + a
history array L=50), samples/s of (a) the text DataReader + default_collate (the reference's loader,
restated), (b) the columnar loader to CPU batches, (c) the columnar loader to device batches incl.
async H2D and on-device CSR expansion."""
import os, sys, time, tempfile
import numpy as np, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.utils.data import DataLoader
from news_recsys_amd.dataset.DataReader.data_reader import DataReader
from news_recsys_amd.dataset.DataReader.columnar import convert_features_txt, ColumnarDataset, ColumnarLoader

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
B = 65536
tmp = tempfile.mkdtemp()
cfg = {"name": "bench", "paths": {"out_basedir": tmp, "user_history_path": ""},
       "features": {"sparse_feature_names": ["user_id", "item_id", "category", "subcategory", "user_click_category"],
                    "dense_feature_names": [], "array_feature_names": ["user_history"],
                    "item_feature_names": ["item_id", "category", "subcategory"],
                    "user_feature_names": ["user_id", "user_click_category", "user_history"], "array_max_length": {"user_history": 50}},
       "embeddings": {"embedding_size": {}, "embedding_table_size": {}, "share_emb_table_features": {"user_history": "item_id"}}}
cpath = os.path.join(tmp, "cfg.yaml"); yaml.safe_dump(cfg, open(cpath, "w"))
rng = np.random.default_rng(0)
txt = os.path.join(tmp, "features.txt")
with open(txt, "w") as f:
    uid = rng.integers(1, 94058, N); iid = rng.integers(1, 65239, N); cat = rng.integers(1, 18, N); sub = rng.integers(1, 270, N)
    ucc = rng.integers(1, 18, N); ln = rng.integers(0, 51, N); lab = rng.integers(0, 2, N)
    for i in range(N):
        h = ",".join(map(str, rng.integers(1, 65239, ln[i])))
        f.write(f"user_id:{uid[i]} item_id:{iid[i]} category:{cat[i]} subcategory:{sub[i]} user_click_category:{ucc[i]} user_history:{h}\t{lab[i]}\n")
print(f"{N} samples, text file {os.path.getsize(txt) / 1e6:.1f} MB", flush=True)

t0 = time.perf_counter(); ds = DataReader(cpath, txt); n = 0
sub_n = min(N, 20000)
for b in DataLoader(torch.utils.data.Subset(ds, range(sub_n)), batch_size=512, shuffle=False, num_workers=0): n += b["user_id"].shape[0]
dt = time.perf_counter() - t0
print(f"(a) text DataReader + default_collate, 1 process : {n / dt:12.0f} samples/s  ({sub_n} samples)", flush=True)

t0 = time.perf_counter(); meta = convert_features_txt(cpath, txt, os.path.join(tmp, "col")); dt = time.perf_counter() - t0
print(f"    one-time conversion text -> columnar            : {N / dt:12.0f} samples/s", flush=True)
cds = ColumnarDataset(os.path.join(tmp, "col"))
for dev, tag in (("cpu", "(b) columnar loader -> CPU batches (host gather) "),) + ((("cuda:0", "(c) columnar loader -> device batches (H2D+expand)"),) if torch.cuda.is_available() else ()):
    for shuffle in (False, True):
        loader = ColumnarLoader(cds, B, dev, shuffle=shuffle)
        for _ in loader: pass
        if dev != "cpu": torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for ep in range(3):
            for b in loader: n += b["user_id"].shape[0]
        if dev != "cpu": torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{tag} shuffle={int(shuffle)}: {n / dt:12.0f} samples/s", flush=True)

if torch.cuda.is_available():
    for shuffle in (False, True):
        loader = ColumnarLoader(cds, B, "cuda:0", shuffle=shuffle, resident=True)
        for _ in loader: pass
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for ep in range(10):
            for b in loader: n += b["user_id"].shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"(d) device-resident dataset ({loader.resident_bytes() / 1e6:.0f} MB in HBM), device-side batch gather shuffle={int(shuffle)}: {n / dt:12.0f} samples/s", flush=True)
