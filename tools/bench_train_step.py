#!/usr/bin/env python3
"""Whole-training-step timing at small batches: (a) the reference's module code restated in plain PyTorch-ROCm eager
(one nn.Embedding per table, masked-mean pooling, torch.cat, the same MLP, Adam), (b) this package's Deep eagerly,
(c) the same step captured in a HIP graph.  Also answers: can a whole training step (fused HIP embedding fwd -> MLP -> BCE -> backward incl. the HIP scatter ->
Adam) be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed?  Compares eager vs replay results and
step time in the host-bound small-batch regime."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd.model.sort.deep.model import Deep
import torch.nn.functional as F

CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "configs", "cf_array_small.yaml")
dev = "cuda:0"
ops.set_index_check("off")


def make(seed):
    torch.manual_seed(seed)
    m = Deep(CFG).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    return m, opt


def batch_like(m, B, gen):
    b = {}
    for n in m.sparse_feature_names:
        rows = m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0]
        b[n] = torch.randint(1, rows, (B,), device=dev, generator=gen)
    for n in m.array_feature_names:
        rows = m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0]
        L = m.array_max_length[n] if hasattr(m, "array_max_length") else 9
        b[n] = torch.randint(1, rows, (B, L), device=dev, generator=gen)
        b[n + "_mask"] = (torch.rand(B, L, device=dev, generator=gen) < 0.6).float()
    for n in m.dense_feature_names:
        b[n] = torch.rand(B, device=dev, generator=gen, dtype=torch.float64)
    b["label"] = (torch.rand(B, 2, device=dev, generator=gen) < 0.3).float()
    return b


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
gen = torch.Generator(device=dev).manual_seed(0)
m_e, opt_e = make(1)
m_g, opt_g = make(1)
batches = [batch_like(m_e, B, gen) for _ in range(6)]


def step(m, opt, b):
    opt.zero_grad(set_to_none=False)
    out = m(b)
    loss = F.binary_cross_entropy(out.view(-1), b["label"][:, 0])
    loss.backward()
    opt.step()
    return loss


static = {k: v.clone() for k, v in batches[0].items()}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):                 # warm-up on a side stream (allocator, lazy init) before capture
    for _ in range(3):
        step(m_g, opt_g, static)
torch.cuda.current_stream().wait_stream(s)
m_g.load_state_dict(m_e.state_dict())
opt_g = torch.optim.Adam(m_g.parameters(), lr=1e-3, capturable=True)
with torch.cuda.stream(s):
    step(m_g, opt_g, static)               # materialise optimizer state outside the graph
torch.cuda.current_stream().wait_stream(s)
m_g.load_state_dict(m_e.state_dict())
for st in opt_g.state.values():
    for k, v in st.items():
        if torch.is_tensor(v): v.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step(m_g, opt_g, static)
m_g.load_state_dict(m_e.state_dict())
for st in opt_g.state.values():
    for k, v in st.items():
        if torch.is_tensor(v): v.zero_()

losses_e, losses_g = [], []
for b in batches:
    losses_e.append(step(m_e, opt_e, b).item())
    for k in static: static[k].copy_(b[k])
    g.replay()
    losses_g.append(static_loss.item())
print("eager  losses:", ["%.6f" % x for x in losses_e])
print("replay losses:", ["%.6f" % x for x in losses_g])
md = max((p.detach() - q.detach()).abs().max().item() for p, q in zip(m_e.parameters(), m_g.parameters()))
print("max |param diff| after 6 steps:", md)

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


class TorchDeep(torch.nn.Module):
    """base_model.py:262-308 + deep/model.py:36-43 in stock torch ops (what the reference runs on any GPU)."""
    def __init__(self, m):
        super().__init__()
        self.tables = torch.nn.ModuleDict({n: torch.nn.Embedding(e.weight.shape[0], e.weight.shape[1], padding_idx=0)
                                           for n, e in m.embedding_tables.items()})
        self.names = sorted(m.user_feature_names | m.item_feature_names)
        self.share = dict(m.share_emb_table_features)
        self.arrays = set(m.array_feature_names)
        import copy
        self.score_fc = copy.deepcopy(m.score_fc)
    def forward(self, b):
        outs = []
        for n in self.names:
            e = self.tables[self.share.get(n, n)](b[n].long())
            if n in self.arrays:
                mk = b[n + "_mask"]
                e = (e * mk.unsqueeze(-1)).sum(dim=1) / (mk.sum(dim=1, keepdim=True) + 1e-8)
            outs.append(e)
        return torch.sigmoid(self.score_fc(torch.cat(outs, dim=1)))

m_t = TorchDeep(m_e).to(dev)
opt_t = torch.optim.Adam(m_t.parameters(), lr=1e-3)
t_ref = timeit(lambda: step(m_t, opt_t, batches[0]))
t_e = timeit(lambda: step(m_e, opt_e, batches[0]))
t_g = timeit(g.replay)
print(f"B={B}: stock-torch eager {t_ref:8.1f} us   this package eager {t_e:8.1f} us   HIP-graph replay {t_g:8.1f} us   "
      f"(x{t_ref / t_g:.1f} vs stock torch)")
