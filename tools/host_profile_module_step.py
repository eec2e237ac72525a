#!/usr/bin/env python3
"""Dev: where the HOST time of one module-path training step goes (FM.forward(batch) + backward through autograd, fused row-sparse
gradients), B = 512 so that the GPU time is negligible.  cProfile, top entries by cumulative and by own time."""
import cProfile, os, pstats, sys, tempfile
import torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd.model.sort.fm.model import FM
dev = torch.device("cuda:0")
F, D = 26, 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512            # usage: host_profile_module_step.py [fused|true|...] [B] [rows] [hook]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
HOOK = len(sys.argv) > 4 and sys.argv[4] == "hook"         # enter the backward through LightningModule.backward (what a trainer calls)
names = [f"C{i:02d}" for i in range(F)]
cfg = {"name": "fm", "paths": {"out_basedir": tempfile.gettempdir(), "user_history_path": ""},
       "features": {"sparse_feature_names": names, "dense_feature_names": [], "array_feature_names": [], "item_feature_names": names[:13],
                    "user_feature_names": names[13:], "array_max_length": {}},
       "embeddings": {"embedding_size": {n: D for n in names}, "embedding_table_size": {n: rows for n in names}, "share_emb_table_features": {},
                      "sparse_grad": sys.argv[1] if len(sys.argv) > 1 else "fused"},
       "dataset": {"batch_size": B, "num_workers": 0, "pin_memory": False},
       "train_hparams": {"val_freq": 1, "max_epoch": 1, "lr": 1e-3, "min_lr": 5e-6, "lr_milestones": [4, 20], "max_step": 30, "device": "gpu", "gpus": [0]}}
with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
    yaml.safe_dump(cfg, f)
m = FM(f.name).to(dev)
batch = {n: torch.randint(1, rows, (B,), device=dev) for n in names}
def step():
    p = m(batch)
    if HOOK:
        m.backward(p.sum())
    else:
        p.sum().backward()
    if m._sparse_sink is not None:
        m._sparse_sink.clear()
for _ in range(50):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print(f"step: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us (wall, B = {B})")
torch.autograd.set_multithreading_enabled(False)
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print(f"step: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us (wall, autograd multithreading off)")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
