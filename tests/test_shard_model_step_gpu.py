"""shard_step.shard_model_step_: a reference-shaped model (FM / Deep with an array feature, the reference's own small YAML schemas) converted to
row-sharded tables trained by the bound step, at world 1, against THE SAME MODEL UNSHARDED in `embeddings.sparse_grad: fused` mode:
  * forward output equal (FM: the logit comes from the pass over the finished concat: rtol 1e-5; Deep: bit for bit);
  * after three optimizer steps (SparseDenseAdam: FusedSparseAdam on the looked-up rows + AdamW on the head) every table row and every dense
    parameter equal to rtol 1e-5 (the pooled bag's gradient is added in another grouping; everything else is the same arithmetic);
  * the full (reference-shaped) state_dict comes back through full_state_dict and loads into a fresh unsharded model.
Reference functions behind it: BaseModel.get_embeddings_from_batch (src/model/BaseModel/base_model.py:284-308) + configure_optimizers
(src/model/sort/deep/model.py:54-65; tables by SparseAdam: a documented deviation)."""
import os

import numpy as np
import pytest
import torch

from news_recsys_amd import shard_step, sharding
from news_recsys_amd.model.sort.deep.model import Deep
from news_recsys_amd.model.sort.fm.model import FM
from tests.conftest import CONFIGS, GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(cls, cfg, g):
    m = cls(os.path.join(CONFIGS, cfg))
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
    m = m.to(DEV)
    m.sparse_grad = "fused"
    return m


@pytest.mark.parametrize("cls,cfg,gname", [(Deep, "cf_array_small.yaml", "model_deep_array"), (FM, "cf_fm_small.yaml", "model_fm")])
def test_bound_sharded_model_trains_like_the_unsharded_fused_model(cls, cfg, gname):
    g = dict(np.load(os.path.join(GOLDEN, gname + ".npz"), allow_pickle=False))
    batch = {k[6:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("batch/")}
    ref, shd = _model(cls, cfg, g), _model(cls, cfg, g)
    keys_before = sorted(shd.state_dict())
    shard_step.shard_model_step_(shd, 0, 1)
    assert sorted(shd.state_dict()) == keys_before
    opt_r = ref.configure_optimizers()["optimizer"]
    opt_s = shd.configure_optimizers()["optimizer"]
    for it in range(3):
        for m, opt in ((ref, opt_r), (shd, opt_s)):
            opt.zero_grad()
            out = m(batch)
            loss = m.bceLoss(out, batch["label"][:, 0])
            loss.backward()
            opt.step()
            if m is ref:
                o_ref = out.detach().clone()
        torch.testing.assert_close(out.detach(), o_ref, rtol=1e-5, atol=1e-6)
    full = sharding.full_state_dict(shd)
    want = ref.state_dict()
    assert sorted(full) == sorted(want)
    for k in want:
        torch.testing.assert_close(full[k], want[k], rtol=1e-5, atol=1e-6, msg=lambda s, k=k: f"{k}: {s}")
    fresh = _model(cls, cfg, g)
    fresh.load_state_dict(full, strict=True)                       # reference-shaped: loads anywhere
    again = _model(cls, cfg, g)
    shard_step.shard_model_step_(again, 0, 1)
    sharding.load_full_state_dict_(again, full)                    # and scatters back into arenas
    with torch.no_grad():
        torch.testing.assert_close(again(batch), fresh(batch), rtol=1e-5, atol=1e-6)


def test_bound_sharded_dssm_trains_like_the_unsharded_fused_model():
    """The DSSM (recall/DSSM/model.py:51-73, 148-180): two towers = two bound steps per forward, the history bag through the pooled channel next to the
    user id group (two exchange groups side by side), the news table shared by the towers (two sink entries for one table, merged by the
    optimizer).  Explicit negative permutations (the reference draws randperm, :63); infoNCE loss; three optimizer steps; tables and towers equal
    to the unsharded model in `sparse_grad: fused` mode to rtol 1e-5 (the pooled bag's gradient is added in another grouping)."""
    from news_recsys_amd.model.recall.DSSM.model import DSSM
    g = dict(np.load(os.path.join(GOLDEN, "model_dssm.npz"), allow_pickle=False))
    batch = {k[6:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("batch/")}
    perms = torch.from_numpy(g["out/perms"])

    def make():
        m = DSSM(os.path.join(CONFIGS, "cf_dssm_small.yaml"), hparams={"negative_sample_rate": 3, "lr": 1e-3, "min_lr": 1e-5, "lr_milestones": [4, 20]})
        m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
        m = m.to(DEV)
        m.sparse_grad = "fused"
        return m

    ref, shd = make(), make()
    shard_step.shard_model_step_(shd, 0, 1)
    opt_r = ref.configure_optimizers()["optimizer"]
    opt_s = shd.configure_optimizers()["optimizer"]
    mask = batch["label"][:, 1]
    for it in range(3):
        outs = {}
        for name, m, opt in (("ref", ref, opt_r), ("shd", shd, opt_s)):
            opt.zero_grad()
            u, i, n = m(batch, perms=perms)
            loss = m.infoNCE_loss(u, i, n, mask=mask)
            loss.backward()
            opt.step()
            outs[name] = (u.detach().clone(), i.detach().clone(), loss.item())
        for a, b in zip(outs["ref"][:2], outs["shd"][:2]):
            torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5)
        assert abs(outs["ref"][2] - outs["shd"][2]) <= 1e-4 * max(1.0, abs(outs["ref"][2]))
    full = sharding.full_state_dict(shd)
    want = ref.state_dict()
    assert sorted(full) == sorted(want)
    for k in want:
        torch.testing.assert_close(full[k], want[k], rtol=1e-4, atol=1e-5, msg=lambda s, k=k: f"{k}: {s}")
