#!/usr/bin/env python3
"""Dev: cProfile of the module-path training step at B = 512 (FM model class, 26 x 100 k-row tables, embeddings.sparse_grad: fused, backward through
the LightningModule.backward hook = on the calling thread, FusedSparseAdam.step) -- where the ~200 us of host time per step go."""
import cProfile, os, pstats, sys, tempfile, time, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd.model.sort.fm.model import FM
from news_recsys_amd.model.model_utils.optim import FusedSparseAdam
dev = "cuda:0"; B, F, D, rows = int(os.environ.get("NRX_PROBE_B", 512)), 26, 16, 100_000
names = [f"f{i:02d}" for i in range(F)]
cfg = {"name": "fm", "paths": {"out_basedir": tempfile.gettempdir(), "user_history_path": ""},
       "features": {"sparse_feature_names": names, "dense_feature_names": [], "array_feature_names": [], "item_feature_names": names[:13],
                    "user_feature_names": names[13:], "array_max_length": {}},
       "embeddings": {"embedding_size": {n: D for n in names}, "embedding_table_size": {n: rows for n in names}, "share_emb_table_features": {},
                      "sparse_grad": "fused"},
       "dataset": {"batch_size": B, "num_workers": 0, "pin_memory": False},
       "train_hparams": {"val_freq": 1, "max_epoch": 1, "lr": 1e-3, "min_lr": 5e-6, "lr_milestones": [4, 20], "max_step": 30, "device": "gpu", "gpus": [0]}}
with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
    yaml.safe_dump(cfg, f)
model = FM(f.name).to(dev)
gen = torch.Generator(device=dev).manual_seed(1)
batches = [{n: torch.randint(1, rows, (B,), device=dev, generator=gen) for n in names} for _ in range(2)]
tabs = [model.embedding_tables[n].weight for n in names]
model(batches[0])
opt = FusedSparseAdam(model._sparse_sink, lr=1e-3, params=tabs)
it = [0]
def step():
    it[0] += 1
    p = model(batches[it[0] & 1])
    model.backward(p.sum())
    opt.step()
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per step (wall)")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
