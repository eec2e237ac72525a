"""CPU tests of the drop-in module surface (host logic only; no kernel runs here): the classes build
from the reference's YAML schema, expose the reference's attributes, carry the reference's state_dict
keys/shapes (checked against the goldens captured from the reference), plan the fused launch in the
reference's feature order, and refuse to run without the HIP path."""
import os

import numpy as np
import pytest
import torch

from news_recsys_amd import _lib, config
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_DENSE, NRX_SPARSE
from news_recsys_amd.model.BaseModel.base_model import BaseModel
from news_recsys_amd.model.model_utils.lr_schedule import CosinDecayLR
from news_recsys_amd.model.recall.DSSM.model import DSSM
from news_recsys_amd.model.sort.dcn.model import DCN
from news_recsys_amd.model.sort.deep.model import Deep
from news_recsys_amd.model.sort.deepfm.model import DeepFM
from news_recsys_amd.model.sort.fm.model import FM
from news_recsys_amd.model.sort.lr.model import LR
from news_recsys_amd.model.sort.widedeep.model import WideDeep
from tests.conftest import CONFIGS, GOLDEN

DSSM_HP = {"negative_sample_rate": 3, "lr": 1e-3, "min_lr": 1e-5, "lr_milestones": [4, 20]}
CASES = [(Deep, "cf_deep_small.yaml", "model_deep"), (FM, "cf_fm_small.yaml", "model_fm"),
         (DCN, "cf_dcn_small.yaml", "model_dcn"), (WideDeep, "cf_widedeep_small.yaml", "model_widedeep"),
         (LR, "cf_lr_small.yaml", "model_lr"), (Deep, "cf_array_small.yaml", "model_deep_array")]


def cfg(name):
    return os.path.join(CONFIGS, name)


def gold(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def state_of(g):
    return {k[len("param/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}


@pytest.mark.parametrize("cls,cfg_name,gname", CASES)
def test_state_dict_matches_reference_and_loads_strict(cls, cfg_name, gname):
    m = cls(cfg(cfg_name))
    ref = state_of(gold(gname))
    mine = m.state_dict()
    assert sorted(mine) == sorted(ref)                      # same keys as the reference's checkpoint
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    m.load_state_dict(ref, strict=True)                     # load_model() path: strict=True (base_model.py:536)


def test_dssm_state_dict_matches_reference():
    m = DSSM(cfg("cf_dssm_small.yaml"), hparams=DSSM_HP)
    ref = state_of(gold("model_dssm"))
    assert sorted(m.state_dict()) == sorted(ref)
    m.load_state_dict(ref, strict=True)
    assert m.user_input_dim == 32 and m.item_input_dim == 24


def test_attributes_of_the_reference_surface():
    m = Deep(cfg("cf_deep_small.yaml"))
    assert m.sparse_feature_names == {"user_id", "item_id", "category", "subcategory", "user_click_category"}
    assert m.item_feature_names == {"item_id", "category", "subcategory"} and m.user_feature_names == {"user_id", "user_click_category"}
    assert m.dense_feature_names == set() and m.array_feature_names == set()
    assert (m.user_input_dim, m.item_input_dim) == (48, 64)            # reference probe: 32+16, 32+16+16
    assert m.embedding_size["user_id"] == 32 and m.embedding_table_size["item_id"] == 61
    assert m.train_hparams.lr == 1e-3 and list(m.train_hparams.lr_milestones) == [4, 20]
    assert m.config.get("name") == "deep"
    assert isinstance(m.embedding_tables, torch.nn.ModuleDict)
    for name, emb in m.embedding_tables.items():
        assert emb.padding_idx == 0 and torch.all(emb.weight[0] == 0)   # row 0 = padding (base_model.py:164)
    assert m.dense_feature_dim == 1                                     # fixes the latent bug at base_model.py:129
    opt = m.configure_optimizers()
    assert isinstance(opt["optimizer"], torch.optim.AdamW) and opt["lr_scheduler"]["interval"] == "step"


def test_shared_tables_and_array_dims():
    m = Deep(cfg("cf_array_small.yaml"))
    assert sorted(m.embedding_tables.keys()) == ["category", "item_id", "user_click_cats", "user_id"]   # user_history shares item_id
    assert m._get_emb_feature_name("user_history") == "item_id"
    assert m.user_input_dim == 16 + 32 + 12 and m.item_input_dim == 32 + 8
    assert m.array_max_length == {"user_history": 7, "user_click_cats": 5}


def test_plan_follows_reference_order_and_routing():
    m = Deep(cfg("cf_array_small.yaml"))
    B = 4
    batch = {"user_id": torch.zeros(B, dtype=torch.long), "item_id": torch.zeros(B, dtype=torch.long),
             "category": torch.zeros(B, dtype=torch.long), "user_history": torch.zeros(B, 7, dtype=torch.long),
             "user_history_mask": torch.ones(B, 7), "user_click_cats": torch.zeros(B, 5, dtype=torch.long),
             "ctr": torch.zeros(B, dtype=torch.float64)}
    names = {"user_id", "ctr", "user_history", "user_click_cats", "category", "not_in_batch"}
    plan, tables, dims, present = m._plan(batch, names, False, ())
    assert present == ["category", "ctr", "user_click_cats", "user_history", "user_id"]       # sorted, missing skipped
    assert dims == [8, 1, 12, 32, 16]
    assert [s.kind for s in plan.slots] == [NRX_SPARSE, NRX_DENSE, NRX_BAG_MEAN, NRX_BAG_MASKED_MEAN, NRX_SPARSE]
    assert [s.out_col for s in plan.slots] == [0, 8, 9, 21, 53] and plan.out_width == 69
    assert tables == ["category", "user_click_cats", "item_id", "user_id"]
    assert plan.slots[3].bag_len == 7 and plan.slots[3].table == 2          # history reads the item_id table
    assert m._plan(batch, names, False, ()) is m._plan(batch, names, False, ())   # cached

    w = WideDeep(cfg("cf_widedeep_small.yaml"))
    batch = {n: torch.zeros(B, dtype=torch.long) for n in w.sparse_feature_names}
    plan, _, dims, present = w._plan(batch, w.user_feature_names | w.item_feature_names, False, tuple(w.wide_feature_names))
    assert present == ["category", "item_id", "subcategory", "user_click_category", "user_id"]
    assert [s.wide_col for s in plan.slots] == [0, -1, 1, 2, -1]
    assert [s.out_col for s in plan.slots] == [0, 16, 48, 64, 80] and plan.out_width == 112 and plan.wide_width == 3
    assert w.score_fc.deep_network.network[0].in_features == 112          # user+item - len(wide)  (widedeep/model.py:39)


def test_errors_match_the_reference():
    m = Deep(cfg("cf_deep_small.yaml"))
    with pytest.raises(ValueError, match="Embedding table not found"):       # base_model.py:268-269
        m.get_feature_embedding("nope", torch.zeros(2, dtype=torch.long))
    with pytest.raises(FileNotFoundError):                                   # base_model.py:71-72
        Deep("/nonexistent.yaml")
    with pytest.raises(NotImplementedError):
        BaseModel(cfg("cf_deep_small.yaml")).forward({})
    out, dims, names = m.get_embeddings_from_batch({}, {"user_id"})           # empty -> (tensor([]), [], [])
    assert out.numel() == 0 and dims == [] and names == []
    f = FM(cfg("cf_deep_small.yaml"))                                        # mixed dims 32/16: torch.stack would fail
    batch = {n: torch.zeros(2, dtype=torch.long) for n in f.sparse_feature_names}
    with pytest.raises(RuntimeError, match="equal size"):
        f._plan(batch, f.user_feature_names | f.item_feature_names, True, ())


def test_no_cpu_fallback_in_the_product_path():
    m = Deep(cfg("cf_deep_small.yaml"))
    batch = {n: torch.ones(3, dtype=torch.long) for n in m.sparse_feature_names}
    with pytest.raises(_lib.NrxError, match="no CPU"):
        m(batch)
    with pytest.raises(_lib.NrxError):
        m.array_feature_pooling(torch.zeros(2, 3, 4), None)


def test_dense_feature_lookup_is_a_cast():
    m = Deep(cfg("cf_array_small.yaml"))
    v = torch.tensor([0.25, 0.5], dtype=torch.float64)
    out = m.get_feature_embedding("ctr", v)
    assert out.dtype == torch.float32 and out.shape == (2, 1)               # base_model.py:264-265


def test_deepfm_config_block():
    m = DeepFM(cfg("cf_fm_small.yaml"))
    assert m.fm_feature_names == m.user_feature_names | m.item_feature_names
    assert m.score_fc.deep_network.network[0].in_features == 80
    import yaml, tempfile
    base = yaml.safe_load(open(cfg("cf_fm_small.yaml")))
    for bad in ({"fm_feature_names": ["nope"]}, {"fm_dim": 3}):
        base["deepfm_cfg"] = bad
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            yaml.safe_dump(base, f)
        with pytest.raises(ValueError):
            DeepFM(f.name)
        os.unlink(f.name)
    base["deepfm_cfg"] = {"fm_feature_names": ["user_id", "item_id"], "fm_dim": 15}
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(base, f)
    assert DeepFM(f.name).fm_feature_names == {"user_id", "item_id"}
    os.unlink(f.name)


def test_dcn_config_defaults_to_reference():
    m = DCN(cfg("cf_dcn_small.yaml"))
    assert len(m.score_fc.cross_net.cross_net) == 3 and m.score_fc.version == 1      # dcn/model.py:36
    assert m.score_fc.cross_net.cross_net[0].w.shape == (112, 1)
    assert m.score_fc.score_fc.network[0].in_features == 224


def test_cosine_schedule_matches_reference_trace():
    g = gold("lr_schedule")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = CosinDecayLR(opt, lrs=list(g["lr"]), milestones=list(g["milestones"]))
    lrs = []
    for _ in range(len(g["lrs"])):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)


def test_config_loader_attr_access():
    c = config.load_config(cfg("cf_widedeep_small.yaml"))
    assert c.wide_and_deep_cfg.wide_feature_names == ["category", "subcategory", "user_click_category"]
    assert c.get("paths", {}).get("out_basedir") == "tests/tmp" and c.get("missing", 7) == 7
    assert isinstance(config.to_container(c), dict)


def test_dssm_keeps_configured_user_history_and_maps_ids(tmp_path):
    """DSSM.__init__ must not discard the history BaseModel loaded from paths.user_history_path
    (reference base_model.py:55-58), and hit_rate's filter must find JSON's string keys (model.py:205-217)."""
    import json
    import yaml
    from news_recsys_amd.model.recall.DSSM.model import DSSM
    hist = {"3": {"5": 1, "7": 1, "99": 1}, "4": {}}
    hp = tmp_path / "user_history.json"
    hp.write_text(json.dumps(hist))
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, "cf_dssm_small.yaml")))
    cfg.setdefault("paths", {})["user_history_path"] = str(hp)
    cfg_path = tmp_path / "cf.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    m = DSSM(str(cfg_path), {}, {"negative_sample_rate": 1, "item_id_feature": "item_id"})
    assert m.user_history == hist
    m._item_pos = {1: 0, 5: 1, 7: 2}                    # what build_item_index leaves: item id -> index position
    m._item_pos_true = None
    assert m._history_positions(3) == [1, 2]            # int user id from the batch, string keys in the JSON; 99 not indexed
    assert m._history_positions(4) == [] and m._history_positions(12345) == []
    # the reference's id translation (emb_idx_2_val_dict, model.py:205,215), when the caller provides it
    m.emb_idx_2_val_dict = {"user_id": {"3": "u3"}, "item_id": {"1": "A", "5": "B", "7": "C"}}
    m.user_history = {"u3": {"C": 1, "A": 1}}
    assert m._history_positions(3) == [0, 2]


def test_deferred_index_report_with_several_plans_in_flight(monkeypatch):
    """Host half of the 'deferred' index check (ops._deferred_status / _raise_deferred): with two plans launched between two
    checks, the offender's feature index is resolved against BOTH name lists -- named plainly only when they agree."""
    import torch
    from news_recsys_amd import ops
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(ops, "_host_status", torch.zeros(4, dtype=torch.int32))
    monkeypatch.setattr(ops, "_host_status_names", [])
    a, b = ["category", "user_history"], ["item_id", "user_id"]
    st = ops._deferred_status(a)
    assert ops._deferred_status(b) is st and ops._host_status_names == [a, b]
    st.copy_(torch.tensor([3, 0, 7, 99], dtype=torch.int32))              # what a kernel of EITHER plan would have written
    with pytest.raises(IndexError) as ei:
        ops._deferred_status(a)
    msg = str(ei.value)
    assert "one of: 'category', 'item_id'" in msg and "sample 7" in msg and "id 99" in msg and "3 lookup" in msg
    assert int(st[0]) == 0 and ops._host_status_names == []               # cleared: the next launch starts a new window
    ops._deferred_status(a)
    st.copy_(torch.tensor([1, 1, 2, 5], dtype=torch.int32))
    with pytest.raises(IndexError) as ei:
        ops._deferred_status(a)
    assert "feature 'user_history'" in str(ei.value)                      # one plan in the window: named plainly


def test_lightning_backward_hook_gives_the_plain_backward_and_restores_the_engine_mode():
    """BaseModel.backward (the LightningModule hook a trainer calls) runs the same backward on the calling thread: same gradients as
    loss.backward(), and the process-wide autograd threading mode is what it was afterwards."""
    m = Deep(cfg("cf_deep_small.yaml"))
    head = torch.nn.Sequential(*[mod for mod in m.score_fc.modules() if isinstance(mod, torch.nn.Linear)][:1])      # the head's first layer (CPU: no kernels here)
    x = torch.randn(6, head[0].in_features)
    (head(x) ** 2).sum().backward()
    want = [p.grad.clone() for p in head.parameters()]
    for p in head.parameters():
        p.grad = None
    before = torch.autograd.is_multithreading_enabled()
    m.backward((head(x) ** 2).sum())
    assert torch.autograd.is_multithreading_enabled() == before
    for p, w in zip(head.parameters(), want):
        assert torch.equal(p.grad, w)
