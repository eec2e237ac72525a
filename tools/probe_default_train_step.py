#!/usr/bin/env python3
"""Dev probe: one training step of the package's Deep model (src/model/sort/deep/model.py: gather 26 features -> 5-layer MLP -> sigmoid -> BCE) in the
DEFAULT gradient mode (dense table gradients, embeddings.sparse_grad off: a reference config without edits), B = 65 536, 26 tables x 100 k rows x 16,
plain SGD so that the optimizer does not dominate.  Run with and without NRX_DENSE_BWD=atomic: here the planning of the sorted backward has the MLP's
forward and backward to hide behind."""
import os, sys, tempfile, time, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd.model.sort.deep.model import Deep
import torch.nn.functional as F
ops.set_index_check("deferred")
dev = torch.device("cuda:0"); B, D, rows, NF = 65536, 16, 100_000, 26
names = [f"f{i}" for i in range(NF)]
cfg = {"name": "deep", "paths": {"out_basedir": tempfile.gettempdir(), "user_history_path": ""},
       "features": {"sparse_feature_names": names, "dense_feature_names": [], "array_feature_names": [], "item_feature_names": names[:13],
                    "user_feature_names": names[13:], "array_max_length": {}},
       "embeddings": {"embedding_size": {n: D for n in names}, "embedding_table_size": {n: rows for n in names}, "share_emb_table_features": {}},
       "dataset": {"batch_size": B, "num_workers": 0, "pin_memory": False},
       "train_hparams": {"val_freq": 1, "max_epoch": 1, "lr": 1e-3, "min_lr": 5e-6, "lr_milestones": [4, 20], "max_step": 30, "device": "gpu", "gpus": [0]}}
with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
    yaml.safe_dump(cfg, f)
model = Deep(f.name).to(dev); os.unlink(f.name)
gen = torch.Generator(device=dev).manual_seed(1)
batches = []
for _ in range(2):
    b = {n: torch.randint(1, rows, (B,), device=dev, generator=gen) for n in names}
    b["label"] = (torch.rand(B, 2, device=dev, generator=gen) < 0.3).float()
    batches.append(b)
opt = torch.optim.SGD(model.parameters(), lr=1e-3)
it = [0]
def step():
    it[0] += 1
    b = batches[it[0] & 1]
    opt.zero_grad(set_to_none=True)
    loss = F.binary_cross_entropy(model(b).view(-1), b["label"][:, 0])
    model.backward(loss)
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
for rep in range(2):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(50): step()
    e.record(); host = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    print(f"Deep, default dense-gradient mode (NRX_DENSE_BWD={os.environ.get('NRX_DENSE_BWD', 'auto')}): "
          f"{a.elapsed_time(e) / 50 * 1e3:7.1f} us per training step (host {host:6.1f} us)", flush=True)
