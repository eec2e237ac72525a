"""The ctypes stub shown in INTEGRATION.md (section B) is executed verbatim (only the library path is
made absolute) and must reproduce the reference's `get_embeddings_from_batch` goldens."""
import os
import re

import numpy as np
import pytest
import torch

from news_recsys_amd import _lib
from news_recsys_amd.model.sort.deep.model import Deep
from tests.conftest import CONFIGS, GOLDEN, ROOT


def _stub_namespace():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", md, flags=re.S).group(1)
    code = code.replace('C.CDLL("libnrx_hip.so")', f'C.CDLL(r"{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    return ns


def test_stub_parses_and_binds_without_gpu():
    ns = _stub_namespace()
    assert callable(ns["fused_embeddings"]) and ns["_lib"].nrx_embed_fwd.restype is not None


@pytest.mark.gpu
@pytest.mark.parametrize("gname,cfg,exact", [("model_deep", "cf_deep_small.yaml", True), ("model_deep_array", "cf_array_small.yaml", False)])
def test_stub_reproduces_reference_goldens(gname, cfg, exact):
    ns = _stub_namespace()
    g = dict(np.load(os.path.join(GOLDEN, gname + ".npz"), allow_pickle=False))
    m = Deep(os.path.join(CONFIGS, cfg))
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
    m = m.to("cuda:0")
    batch = {k[6:]: torch.from_numpy(v).to("cuda:0") for k, v in g.items() if k.startswith("batch/")}
    out, dims, names = ns["fused_embeddings"](m, batch, m.user_feature_names | m.item_feature_names)
    assert dims == list(g["out/dims"]) and names == list(g["out/names"])
    if exact:
        assert np.array_equal(out.cpu().numpy(), g["out/features"])
    else:
        np.testing.assert_allclose(out.cpu().numpy(), g["out/features"], rtol=1e-6, atol=1e-6)
