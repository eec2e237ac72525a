"""ctypes binding of libnrx_hip.so (the C-ABI declared in include/nrx_embed.h).

There is deliberately NO CPU fallback here: if the shared library is missing or a tensor is
not on a ROCm device the call raises.  Build the library with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C news_recsys_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch must be imported BEFORE libnrx_hip.so is dlopen'ed: PyTorch-ROCm bundles its own
# libamdhip64.so, and the process must end up with exactly one HIP runtime (torch's), which our
# library then shares (same SONAME).  Loading ours first would pull in /opt/rocm's copy and leave
# torch and the kernels on two different runtimes ("no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRX_LIB") or os.path.join(_HERE, "lib", "libnrx_hip.so")      # NRX_LIB: another build of the library (A/B runs)

NRX_ABI_VERSION = 3
NRX_MAX_FEATURES = 64
NRX_MAX_DCN_LAYERS = 8
NRX_OK = 0
NRX_ERR_BAD_ARG = -1
NRX_ERR_LAUNCH = -2
NRX_ERR_UNSUPPORTED = -3
NRX_PLAN_SPLIT_PADDING = 1          # nrx_sparse_plan_ex flags
NRX_PLAN_PAIRS = 2
NRX_PLAN_PAYLOAD = 4

# enum nrx_feature_kind
NRX_SPARSE, NRX_DENSE, NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM = 0, 1, 2, 3, 4
NRX_FEAT_ROW0_IS_DATA = 1
NRX_FEAT_BAG_CSR = 2
NRX_FEAT_MANY_PER_ROW = 4


class NrxFeature(C.Structure):
    """struct nrx_feature (include/nrx_embed.h)."""
    _fields_ = [
        ("table", C.c_void_p),
        ("index", C.c_void_p),
        ("weight", C.c_void_p),
        ("rows", C.c_int64),
        ("dim", C.c_int32),
        ("bag_len", C.c_int32),
        ("kind", C.c_int32),
        ("index_bits", C.c_int32),
        ("out_col", C.c_int32),
        ("wide_col", C.c_int32),
        ("fm_field", C.c_int32),
        ("flags", C.c_int32),
    ]


class NrxFmGrad(C.Structure):
    """struct nrx_fm_grad (include/nrx_embed.h)."""
    _fields_ = [("g_fm", C.c_void_p), ("fm_sums", C.c_void_p), ("sums_ld", C.c_int64), ("feat", C.c_void_p), ("feat_ld", C.c_int64)]


class NrxError(RuntimeError):
    pass


_i32, _i64, _p = C.c_int32, C.c_int64, C.c_void_p

# name -> (restype, argtypes); mirrors include/nrx_embed.h one to one
SIGNATURES = {
    "nrx_abi_version": (C.c_int, []),
    "nrx_last_error": (C.c_char_p, []),
    "nrx_device_info": (C.c_int, [C.c_int, C.POINTER(_i64)]),
    "nrx_stream_copy": (C.c_int, [_p, _p, _i64, _p]),
    "nrx_embed_fwd": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _p, _i64, _p, _i64, _p, _p, _p]),
    "nrx_set_small_batch_max": (_i64, [_i64]),
    "nrx_embed_fwd_train": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p]),
    "nrx_embed_bwd": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _p, _i64, _p, _i64, C.POINTER(NrxFmGrad), _p]),
    "nrx_embed_bwd_small": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _p, _i64, _p, _i64, C.POINTER(NrxFmGrad), _i32, _p]),
    "nrx_embed_bwd_small_sparse": (C.c_int, [C.POINTER(NrxFeature), C.POINTER(C.c_int32), _i32, _i64, _p, _i64, _p, _i64, C.POINTER(NrxFmGrad),
                                             _p, _p, _i64, _p]),
    "nrx_rows_mark": (C.c_int, [_p, _i64, _p, _p, _p, _i32, _i32, _p]),
    "nrx_rows_merge": (C.c_int, [_p, _p, _i64, _p, _p, _p, _p, _i32, _i32, _p]),
    "nrx_dense_adamw_rows": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _p, _i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p, _p]),
    "nrx_embed_bwd_sorted_workspace": (_i64, [_i64, _i32]),
    "nrx_embed_bwd_sorted": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p,
                                       C.POINTER(NrxFmGrad), _p, _p, _p]),
    "nrx_embed_bwd_workspace_for": (_i64, [C.POINTER(NrxFeature), _i32, _i64, _i32]),
    "nrx_embed_bwd_placed": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p,
                                       C.POINTER(NrxFmGrad), _p, C.c_uint64, _p, _p, _p, _p, _i64, _p]),
    "nrx_embed_bwd_placed_dense": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p,
                                             C.POINTER(NrxFmGrad), C.POINTER(_p), _i32, _i32, C.c_uint64, _p, _p, _p, _p, _i64, _p]),
    "nrx_embed_bwd_dense_sorted_workspace": (_i64, [C.POINTER(NrxFeature), _i32, _i64, _i32, _i32]),
    "nrx_embed_bwd_dense_sorted": (C.c_int, [C.POINTER(NrxFeature), C.POINTER(_i32), _i32, _i32, _i64, _i32, _p, _i64, _p, _i64,
                                             C.POINTER(NrxFmGrad), C.POINTER(_p), _i32, _i32, _p, _i64, _p]),
    "nrx_embed_bwd_dense_planned_workspace": (_i64, [C.POINTER(NrxFeature), _i32, _i64, _i32, _i32]),
    "nrx_embed_bwd_dense_planned": (C.c_int, [C.POINTER(NrxFeature), C.POINTER(_i32), _i32, _i32, _i64, _i32, _p, _i64, _p, _i64,
                                              C.POINTER(NrxFmGrad), C.POINTER(_p), _i32, _i32, _p, _p, _p, _i64, _p]),
    "nrx_embed_bwd_sparse_planned_workspace": (_i64, [C.POINTER(NrxFeature), _i32, _i64, _i32, _i32]),
    "nrx_embed_bwd_sparse_planned": (C.c_int, [C.POINTER(NrxFeature), C.POINTER(_i32), _i32, _i32, _i64, _i32, _p, _i64, _p, _i64,
                                               C.POINTER(NrxFmGrad), _p, _p, _p, _i32, _p, _p, _p, _i64, _p]),
    "nrx_make_table_keys": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), C.POINTER(_i32), _i32, _i32, _p, _p]),
    "nrx_bag_pool_fwd": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    "nrx_bag_pool_bwd": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    "nrx_fm_fwd": (C.c_int, [_p, _i64, _i32, _i32, _i64, _p, _p]),
    "nrx_fm_fwd_train": (C.c_int, [_p, _i64, _i32, _i32, _i64, _p, _p, _i64, _p]),
    "nrx_fm_bwd": (C.c_int, [_p, _i64, _i32, _i32, _i64, _p, _p, _i64, _p, _i64, _p]),
    "nrx_fm_head_fwd": (C.c_int, [_p, _p, _p, _i64, _p]),
    "nrx_fm_head_state_bytes": (_i64, []),
    "nrx_fm_head_bwd": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _p]),
    "nrx_dcn_v1_fwd": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _p]),
    "nrx_dcn_v1_bwd": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p]),
    "nrx_dcn_v1_bwd_ordered_workspace": (C.c_int64, [_i32, _i32]),
    "nrx_dcn_v1_bwd_ordered": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p]),
    "nrx_embed_dcn_v1_fwd": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _p]),
    "nrx_dcn_v2_layer_fwd": (C.c_int, [_p, _p, _i64, _i64, _i32, _p, _p, _i32, _p, _i64, _p, _p]),
    "nrx_dcn_v2_layer_bwd_workspace": (_i64, [_i64, _i32]),
    "nrx_dcn_v2_layer_bwd": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i64, _i32, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _p, _p, _p, _p]),
    "nrx_linear_wgrad": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p]),
    "nrx_linear_wgrad_ordered_workspace": (C.c_int64, [_i64, _i32, _i32]),
    "nrx_linear_wgrad_ordered": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p]),
    "nrx_route_feat_state_bytes": (_i64, [_i32, _i64, _i32]),
    "nrx_route_feat": (C.c_int, [C.POINTER(_p), _i32, _i64, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_inbox_transpose": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i64, _p]),
    "nrx_pool_inbox_expand": (C.c_int, [_i64, _i32, _i64, _i32, _i64, _p, _p, _p, _p, _i32, _p, _i32, _p, _p, _p]),
    "nrx_shard_dest_combine": (C.c_int, [_p, _p, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _p, _p]),
    "nrx_embed_bwd_scatter_multi": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, C.POINTER(NrxFmGrad), _p, C.POINTER(_p), _i32, _i32, _p]),
    "nrx_embed_bwd_walk": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _p, _p, _p, _i64, _p, C.POINTER(NrxFmGrad), _p,
                                     C.c_uint64, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "nrx_pool_inbox_owner_ids": (C.c_int, [_i64, _i32, _i64, _i32, _i64, _p, _p, _p, _i32, _p, _p, _p]),
    "nrx_pool_order_remap": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p]),
    "nrx_gather_place_feat": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), C.POINTER(_i32), _i32, _i32, _i64, _p, _p, _i32, C.POINTER(_p), _i64, _i64, _p, _p]),
    "nrx_embed_bwd_scatter": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _p, _i64, C.POINTER(NrxFmGrad), _p, _p, _p]),
    "nrx_bucketize_workspace": (_i64, [_i64, _i32]),
    "nrx_bucketize_by_owner": (C.c_int, [_p, _i32, _i64, _i32, _p, _p, _p, _p, _p]),
    "nrx_gather_rows_segmented": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, _p, _p, _i32, _i64, _i32, _p, _p, _p, _p]),
    "nrx_scatter_add_rows_segmented": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, _p, _p, _i32, _i64, _i32, _p, _p, _i32, _p]),
    "nrx_route_workspace": (_i64, [_i64, _i32]),
    "nrx_route_ids": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p]),
    "nrx_route_ids_pos": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_gather_inbox_place": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, C.POINTER(_i32), _i32, _i32, _i64, _p, _p, _p, _i32,
                                         C.POINTER(_p), _i64, _i64, C.POINTER(_i32), _p, _p]),
    "nrx_route_dedup_workspace": (_i64, [_i64, _i32]),
    "nrx_route_ids_dedup": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64), _i32, _i32, _i32, _i32, _i64,
                                      _p, _p, _p, _p, _p, _p]),
    "nrx_unique_inverse_workspace": (_i64, [_i64]),
    "nrx_unique_inverse": (C.c_int, [_p, _i32, _i64, _p, _p, _p, _p, _p]),
    "nrx_bag_norm_weights": (C.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "nrx_bag_norm_weights_inv": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p]),
    "nrx_route_bags_one_state_bytes": (_i64, [C.POINTER(_i32), _i32, _i64, _i32]),
    "nrx_route_bags_one": (C.c_int, [C.POINTER(_p), C.POINTER(_p), C.POINTER(_i32), _i32, _i32, _i64, _i32, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_bag_upstream_rows": (C.c_int, [_p, _i64, _i32, _i32, _i64, _p, _i32, _i64, _p, _p]),
    "nrx_route_bags_runs_state_bytes": (_i64, [C.POINTER(_i32), _i32, _i64, _i32]),
    "nrx_route_bags_runs": (C.c_int, [C.POINTER(_p), C.POINTER(_p), C.POINTER(_i32), C.POINTER(_i32), _i32, _i32, _i64, _i32, _i64, _p, _p, _p,
                                      C.POINTER(_p), _p, _p, _p, _p]),
    "nrx_pool_inbox_fwd_runs": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, C.POINTER(_i32), _i32, _i64, _i32, _i64, _p, _p, _p, _p, _i32,
                                          _p, _p, _p]),
    "nrx_pool_inbox_runs_words": (C.c_int, [_i64, _i32, _i64, _i32, _i64, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "nrx_route_bags": (C.c_int, [C.POINTER(_p), C.POINTER(_p), C.POINTER(_i32), _i32, _i32, _i64, _i32, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_pool_inbox_workspace": (_i64, [_i32, _i64, _i32]),
    "nrx_pool_inbox_fwd": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, C.POINTER(_i32), _i32, _i64, _i32, _i64, _p, _p, _p, _p, _i32,
                                     _p, _p, _p, _p]),
    "nrx_pool_inbox_bwd": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, C.POINTER(_i32), _i32, _i64, _i32, _i64, _p, _p, _p, _p, _i32,
                                     _p, _i32, _p]),
    "nrx_gather_inbox": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, C.POINTER(_i32), _i32, _i32, _i64, _p, _p, _i32, _p, _p, _p]),
    "nrx_scatter_add_inbox": (C.c_int, [C.POINTER(_p), C.POINTER(_i64), _i32, C.POINTER(_i32), _i32, _i32, _i64, _p, _p, _i32, _p, _i32, _p]),
    "nrx_csr_to_padded": (C.c_int, [_p, _i32, _p, _p, _i64, _i32, _p, _p, _p]),
    "nrx_user_rank_metrics": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "nrx_mask_lengths": (C.c_int, [_p, _i64, _i32, _p, _p]),
    "nrx_sparse_plan_workspace": (_i64, [_i64]),
    "nrx_sparse_plan": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    "nrx_sparse_plan_place": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, C.c_uint64, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_sparse_plan_lds_state_bytes": (_i64, []),
    "nrx_sparse_plan_lds_workspace": (_i64, [_i64]),
    "nrx_sparse_plan_lds_ok": (C.c_int, [_p, _p, _p, _i32, _i32]),
    "nrx_sparse_plan_lds": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_sparse_plan_stats": (C.c_int, [_p, _p, _i64, _p, _p]),
    "nrx_sparse_plan_ex": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, C.c_uint64, C.c_uint32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nrx_embed_bwd_placed_pairs": (C.c_int, [C.POINTER(NrxFeature), _i32, _i64, _i32, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p,
                                             C.POINTER(NrxFmGrad), _p, C.POINTER(_p), _i32, _i32, C.c_uint64, _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "nrx_sparse_adam_step": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _p, _i64, _p, C.c_float, _p, C.c_float, C.c_float, C.c_float,
                                       C.c_float, _p]),
    "nrx_rows_to_dense": (C.c_int, [_p, _i32, _i32, _p, _p, _i64, _p, _i32, _p]),
    "nrx_topk_workspace": (_i64, [_i64, _i64, _i32]),
    "nrx_topk_ip": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
}

_lib: Optional[C.CDLL] = None


def load(path: Optional[str] = None) -> C.CDLL:
    """Load the shared library (once).  Loading needs no GPU; calling compute entry points does."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NrxError(
            f"{p} not found: the HIP extension is not built. Run `make -C news_recsys_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback for the product path.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.nrx_abi_version() != NRX_ABI_VERSION:
        raise NrxError(f"ABI version mismatch: library reports {lib.nrx_abi_version()}, binding expects {NRX_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != NRX_OK:
        msg = load().nrx_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise NrxError(f"{what} failed (code {rc}): {msg}")


def is_available() -> bool:
    return os.path.exists(LIB_PATH)
