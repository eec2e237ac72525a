#!/usr/bin/env python3
"""Host-side cost per call of the module path (tiny batch so the kernel time is negligible): ops.embed_apply under each
index-check mode, the bound PreparedEmbed call, the autograd path, and the drop-in class itself (a Deep model built from a
26-feature YAML: model.get_embeddings_from_batch(batch, names) and model.inference(batch))."""
import os, sys, tempfile, time
import torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
dev = torch.device("cuda:0")
F, D, rows, B = 26, 16, 1000, 64
tables = [torch.randn(rows, D, device=dev) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
ids = [torch.randint(1, rows, (B,), device=dev) for _ in range(F)]
none = [None] * F
def t(fn, n=3000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for mode in ("sync", "deferred", "off"):
    ops.set_index_check(mode)
    with torch.no_grad():
        print(f"ops.embed_apply (26 feats, no_grad, index check {mode:8s}): {t(lambda: ops.embed_apply(plan, tables, ids, none)):7.1f} us/call")
ops.set_index_check("deferred")
call = ops.PreparedEmbed(plan, tables, ids, none)
print(f"ops.PreparedEmbed.run                                       : {t(call.run):7.1f} us/call")
tg = [x.clone().requires_grad_(True) for x in tables]
print(f"ops.embed_apply with grad graph (deferred check)            : {t(lambda: ops.embed_apply(plan, tg, ids, none)):7.1f} us/call")

# the drop-in class: Deep built from a 26-feature config
names = [f"C{i:02d}" for i in range(F)]
cfg = {"name": "deep", "paths": {"out_basedir": tempfile.gettempdir(), "user_history_path": ""},
       "features": {"sparse_feature_names": names, "dense_feature_names": [], "array_feature_names": [], "item_feature_names": names[:13],
                    "user_feature_names": names[13:], "array_max_length": {}},
       "embeddings": {"embedding_size": {n: D for n in names}, "embedding_table_size": {n: rows for n in names}, "share_emb_table_features": {}},
       "dataset": {"batch_size": B, "num_workers": 0, "pin_memory": False},
       "train_hparams": {"val_freq": 1, "max_epoch": 1, "lr": 1e-3, "min_lr": 5e-6, "lr_milestones": [4, 20], "max_step": 30, "device": "gpu", "gpus": [0]}}
with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
    yaml.safe_dump(cfg, f)
from news_recsys_amd.model.sort.deep.model import Deep
m = Deep(f.name).to(dev).eval()
batch = {n: torch.randint(1, rows, (B,), device=dev) for n in names}
batch["label"] = torch.zeros(B, 1, device=dev)
fn = m.user_feature_names | m.item_feature_names
with torch.no_grad():
    print(f"Deep.get_embeddings_from_batch (26 feats, no_grad, deferred) : {t(lambda: m.get_embeddings_from_batch(batch, fn)):7.1f} us/call")
    print(f"Deep.inference(batch)  (gather + 5-layer MLP + sigmoid)       : {t(lambda: m.inference(batch)):7.1f} us/call")
    bad = dict(batch); bad[names[3]] = torch.full((B,), rows + 5, device=dev)
    m.get_embeddings_from_batch(bad, fn)          # an out-of-range id: nothing raises here (no sync) ...
    torch.cuda.synchronize()
    try:
        m.get_embeddings_from_batch(batch, fn)    # ... the NEXT call does
        print("deferred index check: NOT raised")
    except IndexError as e:
        print("deferred index check raised on the next call:", str(e)[:90])
