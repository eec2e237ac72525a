#!/usr/bin/env python3
"""Dev: training loops on the module path must hold device memory flat (no tensors kept alive by reference cycles) and
show no periodic stalls.  Runs Deep / DeepFM / DCN(fused gather+cross) for 300 steps each and prints allocated MiB and
the slowest step of every 100."""
import os, sys, time
import torch, yaml, tempfile
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd.model.sort.deep.model import Deep
from news_recsys_amd.model.sort.deepfm.model import DeepFM
from news_recsys_amd.model.sort.dcn.model import DCN
ops.set_index_check("off")
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "configs")
dev = "cuda:0"

def batch_for(m, B, gen):
    b = {}
    for n in m.sparse_feature_names:
        b[n] = torch.randint(1, m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0], (B,), device=dev, generator=gen)
    for n in m.array_feature_names:
        b[n] = torch.randint(1, m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0], (B, 9), device=dev, generator=gen)
        b[n + "_mask"] = (torch.rand(B, 9, device=dev, generator=gen) < 0.6).float()
    b["label"] = (torch.rand(B, 2, device=dev, generator=gen) < 0.4).float()
    return b

def uniform_dcn_cfg():
    cfg = yaml.safe_load(open(os.path.join(CFG, "cf_dcn_small.yaml")))
    for k in cfg["embeddings"]["embedding_size"]: cfg["embeddings"]["embedding_size"][k] = 32
    p = os.path.join(tempfile.mkdtemp(), "dcn32.yaml"); open(p, "w").write(yaml.safe_dump(cfg)); return p

for name, cls, cfg, sg in (("Deep dense", Deep, os.path.join(CFG, "cf_array_small.yaml"), False),
                           ("DeepFM dense", DeepFM, os.path.join(CFG, "cf_fm_small.yaml"), False),
                           ("DeepFM fused-sparse", DeepFM, os.path.join(CFG, "cf_fm_small.yaml"), "fused"),
                           ("DCN fused gather+cross", DCN, uniform_dcn_cfg(), False)):
    torch.manual_seed(0)
    m = cls(cfg).to(dev)
    m.sparse_grad = sg
    opt = m.configure_optimizers()["optimizer"]
    gen = torch.Generator(device=dev).manual_seed(1)
    b = batch_for(m, 4096, gen)
    line = []
    for chunk in range(3):
        worst = 0.0
        for _ in range(100):
            t0 = time.perf_counter()
            opt.zero_grad()
            loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            worst = max(worst, time.perf_counter() - t0)
        line.append(f"{torch.cuda.memory_allocated() >> 10} KiB / worst step {worst * 1e3:.1f} ms")
    print(f"{name:24s}: " + "  |  ".join(line), flush=True)
    del m, opt
