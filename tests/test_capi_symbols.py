"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/nrx_embed.h declares; the ctypes binding covers exactly that set."""
import ctypes
import os
import re

import pytest

from news_recsys_amd import _lib
from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nrx_embed.h")


def declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"^NRX_API\s+[\w\s\*]+?\b(nrx_[a-z0-9_]+)\s*\(", txt, flags=re.M)))


def test_header_declares_the_path():
    names = declared_symbols()
    for must in ("nrx_embed_fwd", "nrx_embed_bwd", "nrx_fm_fwd", "nrx_dcn_v1_fwd", "nrx_dcn_v2_layer_fwd",
                 "nrx_bag_pool_fwd", "nrx_bucketize_by_owner", "nrx_gather_rows_segmented"):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build() first (tests need the built library)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libnrx_hip.so does not export {name}"


def test_binding_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_load_and_abi_version_no_gpu_needed():
    lib = _lib.load()
    assert lib.nrx_abi_version() == _lib.NRX_ABI_VERSION == 3
    assert lib.nrx_bucketize_workspace(5000, 8) == 3 * 8 + 8     # 3 chunks of 2048 ids + offsets


def test_bad_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    rc = lib.nrx_embed_fwd(None, 0, 4, None, 0, None, 0, None, None, None)
    assert rc == -1 and b"n_feats" in lib.nrx_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "nrx_embed_fwd")


def test_struct_layout_matches_c():
    # 3 pointers + int64 + 8 int32 = 64 bytes, as in include/nrx_embed.h
    assert ctypes.sizeof(_lib.NrxFeature) == 64


def test_product_path_has_no_cpu_fallback():
    import torch
    from news_recsys_amd import ops
    plan = ops.EmbedPlan([ops.Slot("a", _lib.NRX_SPARSE, 0, 4)], out_width=4)
    with pytest.raises(_lib.NrxError):
        ops.embed_apply(plan, [torch.zeros(3, 4)], [torch.tensor([1, 2])], [None])
    with pytest.raises(_lib.NrxError):
        ops.dcn_v1(torch.zeros(2, 4), torch.zeros(1, 4), torch.zeros(1, 4))


def test_the_drivers_build_check_follows_the_bindings_abi_version():
    """__graft_entry__.build() -- the driver's "does it build" check -- must compare the library's ABI version with _lib.NRX_ABI_VERSION, not with a
    literal (a literal 2 survived the bump to 3 and failed the check until the end of round 6)."""
    import inspect
    import re
    import __graft_entry__ as entry
    src = inspect.getsource(entry.build)
    assert "_lib.NRX_ABI_VERSION" in src and re.search(r"nrx_abi_version\(\)\s*==\s*\d", src) is None
