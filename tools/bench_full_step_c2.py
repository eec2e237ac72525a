#!/usr/bin/env python3
"""Whole training step of a ranker at the C2 scale: the package's `Deep` model (train_cf_deep.yaml schema) over 26 features x 1 M rows x 16
with the reference's MLP head [416, 128, 128, 128, 64, 1], BCE loss, backward, optimizer, at B = 65 536 --
  (a) `embeddings.sparse_grad: fused`: fused gather forward, deterministic row-sparse backward, fused row-sparse Adam on the tables,
      AdamW on the MLP; (a') the same with the MLP layers' weight gradients on nrx_linear_wgrad (opt-in);
  (b) the reference's own arrangement: dense table gradients + AdamW over every row (src/model/sort/deep/model.py:54-65);
  (c) the module code restated in stock PyTorch-ROCm: one nn.Embedding(sparse=True) per feature + torch.cat + the same MLP,
      torch.optim.SparseAdam on the tables / AdamW on the MLP (the closest stock equivalent of (a)).
Prints ms per step and impressions/s.  usage: bench_full_step_c2.py [batch]"""
import os, sys, tempfile, time
import torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd.model.sort.deep.model import Deep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NF, ROWS, D = 26, 1_000_000, 16
dev = "cuda:0"
ops.set_index_check("deferred")
names = [f"c{i:02d}" for i in range(NF)]
user, item = names[:13], names[13:]
ind = lambda xs: "".join(f"    - {x}\n" for x in xs)
yaml_text = f"""name: deep
paths:
  out_basedir: "{tempfile.gettempdir()}"
  user_history_path: ""
features:
  sparse_feature_names:
{ind(names)}  dense_feature_names: []
  array_feature_names: []
  item_feature_names:
{ind(item)}  user_feature_names:
{ind(user)}  array_max_length: {{}}
embeddings:
  sparse_grad: SPARSE_GRAD
  embedding_size:
{''.join(f'    {n}: {D}' + chr(10) for n in names)}  embedding_table_size:
{''.join(f'    {n}: {ROWS}' + chr(10) for n in names)}  share_emb_table_features: {{}}
dataset:
  batch_size: {B}
  num_workers: 0
  pin_memory: false
train_hparams:
  val_freq: 1
  max_epoch: 1
  lr: 1.0e-3
  min_lr: 5.0e-6
  lr_milestones: [4, 20]
  max_step: 1000
  device: "gpu"
  gpus: [0]
"""

def make_model(mode):
    path = os.path.join(tempfile.gettempdir(), f"nrx_c2_{mode}.yaml")
    open(path, "w").write(yaml_text.replace("SPARSE_GRAD", mode))
    torch.manual_seed(0)
    m = Deep(path).to(dev)
    m.setup("fit") if hasattr(m, "setup") else None
    return m, m.configure_optimizers()["optimizer"]

gen = torch.Generator(device=dev).manual_seed(1)
batches = []
for _ in range(4):
    b = {n: torch.randint(1, ROWS, (B,), device=dev, generator=gen) for n in names}
    b["label"] = (torch.rand(B, 1, device=dev, generator=gen) < 0.3).float()
    batches.append(b)

ISSUE = {}

def timeit(step, n=30, warm=40):          # the caching allocator keeps growing for ~40 steps (a device allocation costs milliseconds of host time)
    for i in range(warm): step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): step(batches[i % 4])
    t1 = time.perf_counter()                      # all steps enqueued: host time per step (== the step time when the host is the bound)
    torch.cuda.synchronize()
    ISSUE["ms"] = (t1 - t0) / n * 1e3
    return (time.perf_counter() - t0) / n * 1e3

def run_package(mode, label):
    m, opt = make_model(mode)
    def step(b):
        opt.zero_grad(set_to_none=True)
        out = m(b)
        loss = F.binary_cross_entropy(out.view(-1), b["label"][:, 0])
        loss.backward()
        opt.step()
    n0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    ms = timeit(step)
    print(f"{label:78s} {ms:8.2f} ms/step  {B / ms / 1e3:7.2f} M impressions/s   (host enqueue {ISSUE['ms']:.2f} ms/step)", flush=True)
    if os.environ.get("HOST_BREAKDOWN") == "1":
        print(f"      device allocations during the 70 steps: {torch.cuda.memory_stats().get('num_device_alloc', 0) - n0}", flush=True)
    if os.environ.get("HOST_BREAKDOWN") == "1":       # host time of each part of the step (nothing synchronises inside)
        acc = [0.0] * 5
        for i in range(30):
            b = batches[i % 4]
            t = [time.perf_counter()]
            opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
            out = m(b); t.append(time.perf_counter())
            loss = F.binary_cross_entropy(out.view(-1), b["label"][:, 0]); t.append(time.perf_counter())
            loss.backward(); t.append(time.perf_counter())
            opt.step(); t.append(time.perf_counter())
            for j in range(5): acc[j] += (t[j + 1] - t[j]) / 30 * 1e3
        torch.cuda.synchronize()
        print("      host ms/step: zero_grad %.3f  forward %.3f  loss %.3f  backward %.3f  optimizer %.3f" % tuple(acc), flush=True)
    del m, opt
    torch.cuda.empty_cache()

if os.environ.get("ONLY_A") == "1":               # tools/host_profile_step_c2.py: cProfile of the host side of (a)
    import cProfile, pstats, torch.nn.functional as F_
    m, opt = make_model("fused")
    def step(b):
        opt.zero_grad(set_to_none=True)
        loss = F_.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
        loss.backward()
        opt.step()
    for i in range(10): step(batches[i % 4])
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for i in range(50): step(batches[i % 4])
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
    sys.exit(0)
LEGS = set(os.environ.get("LEGS", "a,a1,a1g,ag,b,c").split(","))          # which lines to produce (rocprofv3 runs pick one)
if "a" in LEGS:
    run_package("fused", "(a) this package, sparse_grad: fused (row-sparse backward + fused row-sparse Adam)")
from news_recsys_amd.model.model_utils import utils as _mlp_utils
_mlp_utils.MLP_WGRAD = True
if "a1" in LEGS:
    run_package("fused", "(a') as (a) with the MLP weight gradients on nrx_linear_wgrad (NRX_MLP_WGRAD=1)")
def run_graphed(label):
    """(a) / (a') replayed from a HIP graph (news_recsys_amd.graph.GraphedStep): the host enqueues one graph launch per step."""
    from news_recsys_amd.graph import GraphedStep
    from news_recsys_amd.model.model_utils.optim import SparseDenseAdam
    m, _ = make_model("fused")
    tabs = [e.weight for e in m.embedding_tables.values()]
    ids = {id(p) for p in tabs}
    opt = SparseDenseAdam(tabs, [p for p in m.parameters() if id(p) not in ids], lr=1e-3, fused_sink=m._sparse_sink, capturable=True)
    def step(b):
        opt.zero_grad(set_to_none=False)
        loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
        loss.backward()
        opt.step()
        return loss
    prev = ops._INDEX_CHECK
    ops.set_index_check("off")
    try:
        gs = GraphedStep(step, batches[0], warmup=3)
        ms = timeit(lambda b: gs(b))
    finally:
        ops.set_index_check(prev)
    print(f"{label:78s} {ms:8.2f} ms/step  {B / ms / 1e3:7.2f} M impressions/s   (host enqueue {ISSUE['ms']:.2f} ms/step)", flush=True)
    del gs, m, opt
    torch.cuda.empty_cache()

if "a1g" in LEGS:
    run_graphed("(a'g) (a') replayed from a HIP graph (GraphedStep; ids not range-checked inside a replay)")
_mlp_utils.MLP_WGRAD = False
if "ag" in LEGS:
    run_graphed("(ag) (a) replayed from a HIP graph")
if "b" in LEGS:
    run_package("false", "(b) this package, the reference's arrangement (dense table grads + AdamW over all rows)")

if "c" not in LEGS:
    sys.exit(0)

class Stock(nn.Module):
    def __init__(self):
        super().__init__()
        self.tabs = nn.ModuleList([nn.Embedding(ROWS, D, padding_idx=0, sparse=True) for _ in range(NF)])
        dims = [NF * D, 128, 128, 128, 64, 1]
        layers = []
        for i in range(len(dims) - 1):
            layers.append(nn.Linear(dims[i], dims[i + 1]))
            if i < len(dims) - 2: layers.append(nn.ReLU())
        self.mlp = nn.Sequential(*layers)
    def forward(self, b):
        x = torch.cat([t(b[n]) for t, n in zip(self.tabs, names)], dim=1)
        return torch.sigmoid(self.mlp(x))

torch.manual_seed(0)
sm = Stock().to(dev)
o1 = torch.optim.SparseAdam([t.weight for t in sm.tabs], lr=1e-3)
o2 = torch.optim.AdamW(sm.mlp.parameters(), lr=1e-3)
def stock_step(b):
    o1.zero_grad(set_to_none=True); o2.zero_grad(set_to_none=True)
    loss = F.binary_cross_entropy(sm(b).view(-1), b["label"][:, 0])
    loss.backward()
    o1.step(); o2.step()
ms = timeit(stock_step, n=10, warm=3)
print(f"{'(c) stock PyTorch-ROCm: nn.Embedding(sparse=True) x 26 + cat + MLP, SparseAdam + AdamW':78s} {ms:8.2f} ms/step  {B / ms / 1e3:7.2f} M impressions/s", flush=True)
