#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE'S OWN CODE on CPU.

This script only runs in the build container (it needs /root/reference, which does not
exist on the GPU box).  It imports the reference's modules -- with three stub modules for
the packages the image lacks (omegaconf / lightning / pytorch_lightning; they only touch
config + trainer plumbing, all arithmetic is the reference's code on genuine torch) -- and
writes small .npz fixtures next to this file.  No reference source is copied: the fixtures
hold tensors only (weights, index batches, outputs, gradients).

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz
    python tests/golden/gen_golden.py --check  # regenerates into a temp dir and compares with the committed fixtures

Reproducibility: the reference iterates Python sets of feature names (base_model.py:85-90), so the order in which its
modules register parameters -- and with it which random numbers land in which table -- follows the process's string-hash
seed.  The script therefore pins PYTHONHASHSEED=0 by re-executing itself before anything is imported (no GPU is involved),
and randomises parameters in sorted-name order; two runs give bit-identical arrays (checked by --check and by
tests/test_golden_reproducible.py).

Reference entry points exercised (paths relative to /root/reference):
  src/model/BaseModel/base_model.py:262-308   get_feature_embedding / array_feature_pooling /
                                              get_embeddings_from_batch
  src/model/sort/deep/model.py:12-43          DeepModel / Deep.forward
  src/model/sort/fm/model.py:12-59            FMModel / FM.get_inp_embedding
  src/model/sort/dcn/dcn_arch.py:5-91         DCNLayer / DCNv2Layer / DCNNet / DCNv2Net
  src/model/sort/dcn/model.py:15-45           DCNModel / DCN.forward
  src/model/sort/widedeep/model.py:14-69      WideDeepModel / WideDeep.get_inp_embedding
  src/model/sort/lr/model.py:24-31            LR.forward
  src/model/recall/DSSM/model.py:26-110,148-180  towers, losses, user/item embedding
  src/model/model_utils/lr_schedule.py:6-28   CosinDecayLR
"""
import os
import sys

if os.environ.get("PYTHONHASHSEED") != "0":          # before torch / the reference are imported: pin the set order
    import subprocess                                # (a child process, not an exec: no exec anywhere in this tree)
    sys.exit(subprocess.run([sys.executable] + sys.argv, env={**os.environ, "PYTHONHASHSEED": "0"}).returncode)

import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE                      # --check writes to a temp dir instead
CFG = os.path.join(HERE, "configs")
REF = "/root/reference"


# --------------------------------------------------------------------------- stubs
class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _wrap(x):
    if isinstance(x, dict):
        return _AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _unwrap(x):
    if isinstance(x, dict):
        return {k: _unwrap(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_unwrap(v) for v in x]
    return x


def install_stubs():
    om = types.ModuleType("omegaconf")

    class OmegaConf:
        @staticmethod
        def load(p):
            with open(p) as f:
                return _wrap(yaml.safe_load(f))

        @staticmethod
        def to_container(c, resolve=True):
            return _unwrap(c)

    om.OmegaConf = OmegaConf
    om.DictConfig = _AttrDict
    sys.modules["omegaconf"] = om

    L = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device("cpu")

    L.LightningModule = LightningModule
    L.seed_everything = lambda s, workers=False: torch.manual_seed(s)
    sys.modules["lightning"] = L
    for n in ["pytorch_lightning", "pytorch_lightning.utilities",
              "pytorch_lightning.utilities.model_summary"]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["pytorch_lightning.utilities.model_summary"].ModelSummary = object
    sys.modules["faiss"] = types.ModuleType("faiss")
    sys.path.insert(0, REF)


def np_(t):
    return t.detach().cpu().numpy()


def save(name, d):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays)")


def randomize_tables(model, g):
    """Explicit table weights (the reference's init order is hash-seed dependent)."""
    with torch.no_grad():
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):     # sorted: independent of registration order
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 if "embedding" in name else 0.2))
        for emb in model.embedding_tables.values():
            emb.weight[0].zero_()  # padding_idx row, as nn.Embedding(padding_idx=0) initialises it


def make_ids(g, rows, shape, pad_frac=0.15):
    ids = torch.randint(1, rows, shape, generator=g)
    ids[torch.rand(shape, generator=g) < pad_frac] = 0
    flat = ids.view(-1)
    flat[0] = rows - 1   # max index
    if flat.numel() > 3:
        flat[1] = flat[2]  # a guaranteed duplicate
    return ids


def model_case(cls, cfg_name, seed, array_cfg=False):
    g = torch.Generator().manual_seed(seed)
    cfg_path = os.path.join(CFG, cfg_name)
    model = cls(cfg_path)
    randomize_tables(model, g)
    cfg = yaml.safe_load(open(cfg_path))
    B = 24
    batch = {}
    tsize = cfg["embeddings"]["embedding_table_size"]
    share = cfg["embeddings"].get("share_emb_table_features") or {}
    for f in cfg["features"]["sparse_feature_names"]:
        batch[f] = make_ids(g, tsize[share.get(f, f)], (B,))
    for f in cfg["features"]["array_feature_names"]:
        Lf = cfg["features"]["array_max_length"][f]
        ids = make_ids(g, tsize[share.get(f, f)], (B, Lf), pad_frac=0.0)
        lens = torch.randint(0, Lf + 1, (B,), generator=g)
        lens[0] = Lf
        lens[1] = 0            # an all-masked bag
        mask = (torch.arange(Lf)[None, :] < lens[:, None]).float()
        ids = ids * mask.long()  # DataReader pads with id 0
        batch[f] = ids
        batch[f + "_mask"] = mask
    labels = (torch.rand(B, 1, generator=g) < 0.4).float()
    batch["label"] = labels
    out = model(batch)
    loss = F.binary_cross_entropy(out.view(-1), labels[:, 0].view(-1), reduction="mean")
    loss.backward()
    d = {}
    for k, v in model.state_dict().items():
        d["param/" + k] = np_(v)
    for k, v in batch.items():
        d["batch/" + k] = np_(v)
    names = model.user_feature_names | model.item_feature_names
    with torch.no_grad():
        feats, dims, fnames = model.get_embeddings_from_batch(batch, names)
    d["out/features"] = np_(feats)
    d["out/dims"] = np.array(dims, dtype=np.int64)
    d["out/names"] = np.array(fnames)
    d["out/forward"] = np_(out)
    d["out/loss"] = np_(loss)
    for k, p in model.named_parameters():
        d["grad/" + k] = np_(p.grad)
    return model, batch, d


def gen_models():
    from src.model.sort.deep.model import Deep
    from src.model.sort.fm.model import FM
    from src.model.sort.dcn.model import DCN
    from src.model.sort.widedeep.model import WideDeep
    from src.model.sort.lr.model import LR

    _, _, d = model_case(Deep, "cf_deep_small.yaml", 1)
    save("model_deep", d)

    m, b, d = model_case(FM, "cf_fm_small.yaml", 2)
    with torch.no_grad():
        w, v = m.get_inp_embedding(b)
    d["out/fm_w"], d["out/fm_v"] = np_(w), np_(v)
    save("model_fm", d)

    m, b, d = model_case(DCN, "cf_dcn_small.yaml", 3)
    with torch.no_grad():
        x = m.get_inp_embedding(b)
        d["out/cross"] = np_(m.score_fc.cross_net(x))
    save("model_dcn", d)

    m, b, d = model_case(WideDeep, "cf_widedeep_small.yaml", 4)
    with torch.no_grad():
        wx, dx = m.get_inp_embedding(b)
    d["out/wide_x"], d["out/deep_x"] = np_(wx), np_(dx)
    save("model_widedeep", d)

    _, _, d = model_case(LR, "cf_lr_small.yaml", 5)
    save("model_lr", d)

    # Deep with array (bag) features, a shared table, masks incl. an all-zero row.
    m, b, d = model_case(Deep, "cf_array_small.yaml", 6)
    # dense feature + non-binary mask weights + mask=None through get_embeddings_from_batch
    g = torch.Generator().manual_seed(66)
    b2 = {k: v.clone() for k, v in b.items()}
    b2["ctr"] = torch.rand(b["user_id"].shape[0], generator=g, dtype=torch.float64)
    b2["user_history_mask"] = torch.rand(b["user_history_mask"].shape, generator=g)
    b2["user_history_mask"][2] = 0.0
    del b2["user_click_cats_mask"]          # -> plain mean over L incl. padding
    names = {"user_id", "ctr", "user_history", "user_click_cats", "category"}
    with torch.no_grad():
        feats, dims, fnames = m.get_embeddings_from_batch(b2, names)
    d["case2/batch/ctr"] = np_(b2["ctr"])
    d["case2/batch/user_history_mask"] = np_(b2["user_history_mask"])
    d["case2/names_in"] = np.array(sorted(names))
    d["case2/features"] = np_(feats)
    d["case2/dims"] = np.array(dims, dtype=np.int64)
    # missing feature in batch -> skipped (reference returns unfiltered name list)
    b3 = {k: v for k, v in b.items() if k != "category"}
    with torch.no_grad():
        feats3, dims3, names3 = m.get_embeddings_from_batch(b3, {"user_id", "category", "item_id"})
    d["case3/features"] = np_(feats3)
    d["case3/dims"] = np.array(dims3, dtype=np.int64)
    d["case3/names_returned"] = np.array(names3)
    save("model_deep_array", d)


def gen_ops():
    """Op-level vectors: pooling, FM, DCN v1/v2, odd dims."""
    from src.model.BaseModel.base_model import BaseModel
    from src.model.sort.fm.model import FMModel
    from src.model.sort.dcn.dcn_arch import DCNNet, DCNv2Net

    g = torch.Generator().manual_seed(100)
    d = {}
    # --- array_feature_pooling (base_model.py:273-282): unbound call, `self` unused
    for tag, (B, L, D) in {"a": (9, 6, 16), "b": (5, 50, 17), "c": (4, 3, 1), "d": (7, 11, 64)}.items():
        emb = torch.randn(B, L, D, generator=g, requires_grad=True)
        lens = torch.randint(0, L + 1, (B,), generator=g)
        lens[0] = 0
        mask = (torch.arange(L)[None] < lens[:, None]).float()
        wmask = torch.rand(B, L, generator=g) * mask
        up = torch.randn(B, D, generator=g)
        for mtag, mk in (("none", None), ("bin", mask), ("w", wmask)):
            emb.grad = None
            out = BaseModel.array_feature_pooling(None, emb, mk)
            (out * up).sum().backward()
            d[f"pool/{tag}/{mtag}/out"] = np_(out)
            d[f"pool/{tag}/{mtag}/gemb"] = np_(emb.grad)
        d[f"pool/{tag}/emb"], d[f"pool/{tag}/mask"] = np_(emb), np_(mask)
        d[f"pool/{tag}/wmask"], d[f"pool/{tag}/up"] = np_(wmask), np_(up)

    # --- FMModel.forward (fm/model.py:18-26)
    for tag, (B, Fn, K) in {"a": (13, 5, 15), "b": (6, 26, 15), "c": (3, 2, 1), "d": (8, 7, 31)}.items():
        fm = FMModel()
        with torch.no_grad():
            fm.bias.fill_(0.37)
        w = torch.randn(B, Fn, generator=g, requires_grad=True)
        v = (torch.randn(B, Fn, K, generator=g) * 0.7).requires_grad_()
        up = torch.randn(B, 1, generator=g)
        out = fm(w, v)
        (out * up).sum().backward()
        d[f"fm/{tag}/w"], d[f"fm/{tag}/v"], d[f"fm/{tag}/up"] = np_(w), np_(v), np_(up)
        d[f"fm/{tag}/bias"] = np_(fm.bias)
        d[f"fm/{tag}/out"] = np_(out)
        d[f"fm/{tag}/gw"], d[f"fm/{tag}/gv"], d[f"fm/{tag}/gbias"] = np_(w.grad), np_(v.grad), np_(fm.bias.grad)

    # --- DCNNet / DCNv2Net (dcn_arch.py:53-91), 1/2/3 layers, nonzero b
    for tag, (B, D, nl) in {"a": (11, 112, 3), "b": (5, 320, 2), "c": (7, 16, 1), "d": (4, 37, 3)}.items():
        net = DCNNet(D, nl)
        with torch.no_grad():
            for lyr in net.cross_net:
                lyr.b.copy_(torch.randn(D, 1, generator=g) * 0.1)
                lyr.w.copy_(torch.randn(D, 1, generator=g) * (1.0 / D ** 0.5))
        x = torch.randn(B, D, generator=g, requires_grad=True)
        up = torch.randn(B, D, generator=g)
        out = net(x)
        (out * up).sum().backward()
        d[f"dcn1/{tag}/x"], d[f"dcn1/{tag}/up"], d[f"dcn1/{tag}/out"] = np_(x), np_(up), np_(out)
        d[f"dcn1/{tag}/w"] = np.stack([np_(l.w)[:, 0] for l in net.cross_net])
        d[f"dcn1/{tag}/b"] = np.stack([np_(l.b)[:, 0] for l in net.cross_net])
        d[f"dcn1/{tag}/gx"] = np_(x.grad)
        d[f"dcn1/{tag}/gw"] = np.stack([np_(l.w.grad)[:, 0] for l in net.cross_net])
        d[f"dcn1/{tag}/gb"] = np.stack([np_(l.b.grad)[:, 0] for l in net.cross_net])

        net2 = DCNv2Net(D, nl)
        lins = [l.linear for l in net2.cross_net if hasattr(l, "linear")]
        with torch.no_grad():
            for lin in lins:
                lin.weight.copy_(torch.randn(D, D, generator=g) * (1.0 / D ** 0.5))
                lin.bias.copy_(torch.randn(D, generator=g) * 0.1)
        x2 = torch.randn(B, D, generator=g, requires_grad=True)
        out2 = net2(x2)
        (out2 * up).sum().backward()
        d[f"dcn2/{tag}/x"], d[f"dcn2/{tag}/out"], d[f"dcn2/{tag}/up"] = np_(x2), np_(out2), np_(up)
        d[f"dcn2/{tag}/W"] = np.stack([np_(l.weight) for l in lins])
        d[f"dcn2/{tag}/b"] = np.stack([np_(l.bias) for l in lins])
        d[f"dcn2/{tag}/gx"] = np_(x2.grad)
        d[f"dcn2/{tag}/gW"] = np.stack([np_(l.weight.grad) for l in lins])
        d[f"dcn2/{tag}/gb"] = np.stack([np_(l.bias.grad) for l in lins])
    save("ops", d)


def gen_dssm():
    """DSSM (recall/DSSM/model.py): stale imports aliased; set iteration order -> sorted order
    is what the fixture pins, so the per-feature pieces are captured separately."""
    import src.model.BaseModel.base_model as bm
    import src.dataset.DataReader.data_reader as dr
    import src.model.model_utils.lr_schedule as ls
    for alias, mod in (("BaseModel", types.ModuleType("BaseModel")), ("BaseModel.base_model", bm),
                       ("DataReader", types.ModuleType("DataReader")), ("DataReader.data_reader", dr),
                       ("model_utils", types.ModuleType("model_utils")), ("model_utils.lr_schedule", ls)):
        sys.modules[alias] = mod
    from src.model.recall.DSSM.model import DSSM
    DSSM.get_features_embedding = bm.BaseModel.get_feature_embedding
    g = torch.Generator().manual_seed(7)
    cfg_path = os.path.join(CFG, "cf_dssm_small.yaml")
    m = DSSM(cfg_path, hparams={"negative_sample_rate": 3, "lr": 1e-3, "min_lr": 1e-5, "lr_milestones": [4, 20]})
    randomize_tables(m, g)
    cfg = yaml.safe_load(open(cfg_path))
    B, Lh = 16, 9
    tsize = cfg["embeddings"]["embedding_table_size"]
    batch = {"user_id": make_ids(g, tsize["user_id"], (B,)),
             "item_id": make_ids(g, tsize["item_id"], (B,)),
             "category": make_ids(g, tsize["category"], (B,))}
    lens = torch.randint(0, Lh + 1, (B,), generator=g)
    lens[0], lens[1] = Lh, 0
    mask = (torch.arange(Lh)[None] < lens[:, None]).float()
    batch["user_history"] = make_ids(g, tsize["item_id"], (B, Lh), 0.0) * mask.long()
    batch["user_history_mask"] = mask
    batch["label"] = torch.stack([(torch.rand(B, generator=g) < 0.5).float(),
                                  (torch.rand(B, generator=g) < 0.7).float()], dim=1)
    d = {}
    for k, v in m.state_dict().items():
        d["param/" + k] = np_(v)
    for k, v in batch.items():
        d["batch/" + k] = np_(v)
    # The reference iterates python sets (hash-seed dependent); pin the SORTED order by
    # temporarily replacing the sets with sorted lists -- the loop bodies are unchanged.
    m.user_feature_names = sorted(m.user_feature_names)
    m.item_feature_names = sorted(m.item_feature_names)
    with torch.no_grad():
        uvec = m.get_user_embedding(batch)
        ivec = m.get_item_embedding(batch)
        uemb = F.normalize(m.user_fc(uvec), p=2, dim=1)
        iemb = F.normalize(m.item_fc(ivec), p=2, dim=1)
        raw_item = m.item_fc(ivec)
    d["out/user_vector"], d["out/item_vector"] = np_(uvec), np_(ivec)
    d["out/user_emb"], d["out/item_emb"] = np_(uemb), np_(iemb)
    # explicit permutations replace torch.randperm (model.py:63) for the loss goldens
    perms = torch.stack([torch.randperm(B, generator=g) for _ in range(3)])
    neg = F.normalize(torch.stack([raw_item[p] for p in perms], dim=1), p=2, dim=-1)
    d["out/perms"] = np_(perms)
    d["out/neg_item_emb"] = np_(neg)
    msk = batch["label"][:, 1]
    d["out/infonce"] = np_(m.infoNCE_loss(uemb, iemb, neg, mask=msk))
    d["out/triplet"] = np_(m.triplet_loss(uemb, iemb, neg, mask=msk))
    save("model_dssm", d)


def gen_lr_schedule():
    from src.model.model_utils.lr_schedule import CosinDecayLR
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = CosinDecayLR(opt, lrs=[1e-3, 5e-6], milestones=[4, 20])
    lrs = []
    for _ in range(30):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    save("lr_schedule", {"lrs": np.array(lrs, dtype=np.float64), "milestones": np.array([4, 20]),
                         "lr": np.array([1e-3, 5e-6])})


def gen_datareader():
    """Input pipeline goldens: a small synthetic feature file (data generated here, committed under
    tests/golden/data/) parsed by the REFERENCE DataReader (src/dataset/DataReader/data_reader.py) and
    collated by torch's default_collate exactly as MINDDataModule's DataLoader does (pl_dataloader.py:77-96)."""
    from torch.utils.data import DataLoader
    from src.dataset.DataReader.data_reader import DataReader
    rng = np.random.default_rng(2026)
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    path = os.path.join(OUT, "data", "features_small.txt")
    lines = []
    for i in range(23):
        n_hist = [0, 9, 7, 1][i] if i < 4 else int(rng.integers(0, 10))       # empty, over-long (truncated to 7), exact, one
        hist = ",".join(str(int(x)) for x in rng.integers(1, 41, n_hist))
        cats = ",".join(str(int(x)) for x in rng.integers(1, 18, int(rng.integers(0, 6))))
        feats = [f"user_id:{int(rng.integers(1, 53))}", f"item_id:{int(rng.integers(1, 41))}",
                 f"category:{int(rng.integers(1, 18))}", f"ctr:{rng.random():.6f}", f"user_history:{hist}",
                 f"user_click_cats:{cats}", f"ignored_feature:{i}"]
        rng.shuffle(feats)                                                      # order inside a line is free
        lines.append(" ".join(feats) + "\t" + f"{int(rng.integers(0, 2))} {int(rng.integers(0, 2))}")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n\n")                                     # trailing blank line is skipped
    ds = DataReader(os.path.join(CFG, "cf_array_small.yaml"), path)
    d = {"n": np.array(len(ds))}
    for bi, batch in enumerate(DataLoader(ds, batch_size=7, shuffle=False)):
        for k, v in batch.items():
            d[f"b{bi}/{k}"] = np_(v)
    save("datareader", d)


def gen_validation():
    """Validation-metric goldens: synthetic (uid, score, label) streams pushed through the REFERENCE's
    BaseModel.on_validation_epoch_end (base_model.py:333-528) exactly as validation_step fills
    user_scores_dict (:320-330); the printed val_log text is captured (the function returns nothing)."""
    import contextlib, io
    from types import SimpleNamespace
    from src.model.BaseModel.base_model import BaseModel
    d = {}
    for case, (n_users, n, tie, seed, warm_frac) in {"a": (40, 600, False, 1, 0.6), "b": (300, 5000, True, 2, 0.5),
                                                     "c": (5, 40, True, 3, 1.0), "d": (60, 900, False, 4, 0.0)}.items():
        rng = np.random.default_rng(seed)
        uid = rng.integers(1, n_users + 1, n)
        score = rng.random(n).astype(np.float32)
        if tie:
            score = np.round(score, 1).astype(np.float32)            # heavy ties
        label = (rng.random(n) < 0.15 + 0.5 * score).astype(np.float32)
        label[uid == 1] = 0.0                                         # a user without positives
        label[uid == 2] = 1.0                                         # a user with a single class (no AUC)
        warm = sorted(rng.choice(np.arange(1, n_users + 1), int(n_users * warm_frac), replace=False).tolist())
        stub = SimpleNamespace(user_scores_dict={}, user_in_train_set=set(warm), current_epoch=3)
        for u, s_, y in zip(uid, score, label):                       # validation_step's loop (:326-329)
            stub.user_scores_dict.setdefault(u, []).append((s_, y))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            BaseModel.on_validation_epoch_end(stub)
        d[f"{case}/uid"], d[f"{case}/score"], d[f"{case}/label"] = uid, score, label
        d[f"{case}/warm"] = np.array(warm, dtype=np.int64)
        d[f"{case}/log"] = np.array(buf.getvalue())
    save("validation", d)


def generate_all():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    gen_models()
    gen_ops()
    gen_dssm()
    gen_lr_schedule()
    gen_datareader()
    gen_validation()


def compare_dirs(ref_dir, new_dir):
    """Array-for-array, bit-for-bit comparison of every fixture (npz containers carry zip timestamps, so the
    arrays are compared, not the container bytes).  Returns a list of differences."""
    bad = []
    names = sorted(f for f in os.listdir(new_dir) if f.endswith(".npz"))
    for f in names:
        if not os.path.exists(os.path.join(ref_dir, f)):
            bad.append(f"{f}: not committed")
            continue
        a, b = np.load(os.path.join(ref_dir, f)), np.load(os.path.join(new_dir, f))
        if sorted(a.files) != sorted(b.files):
            bad.append(f"{f}: key sets differ")
            continue
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
                bad.append(f"{f}:{k}")
    for f in sorted(os.listdir(ref_dir)):
        if f.endswith(".npz") and f not in names:
            bad.append(f"{f}: committed but not regenerated")
    a = open(os.path.join(ref_dir, "data", "features_small.txt"), "rb").read()
    b = open(os.path.join(new_dir, "data", "features_small.txt"), "rb").read()
    if a != b:
        bad.append("data/features_small.txt")
    return bad


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    install_stubs()
    if "--check" in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as tmp:
            OUT = tmp
            generate_all()
            bad = compare_dirs(HERE, tmp)
        if bad:
            sys.exit("golden fixtures differ from a regeneration:\n  " + "\n  ".join(bad))
        print("OK: a regeneration reproduces every committed fixture bit for bit")
    else:
        generate_all()
