"""torch-facing operators of the path: thin autograd wrappers over the C-ABI (include/nrx_embed.h).

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every operator runs a
hand-written HIP kernel from libnrx_hip.so and raises if the library is missing or a tensor is not
on a ROCm device -- there is no eager/CPU fallback.

Operator                          replaces (reference file:line)
  embed_apply / EmbedPlan          BaseModel.get_embeddings_from_batch  BaseModel/base_model.py:284-308
                                   (+ wide split widedeep/model.py:53-69, FM fm/model.py:18-26,48-59)
  bag_pool                         BaseModel.array_feature_pooling      base_model.py:273-282
  fm_interaction                   FM.get_inp_embedding + FMModel.forward (pre-sigmoid)  fm/model.py
  dcn_v1 / dcn_v1_cat_             DCNLayer / DCNNet (+ torch.cat of dcn/model.py:29)  dcn/dcn_arch.py:14-30,63-70
  dcn_v2                           DCNv2Layer / DCNv2Net                dcn/dcn_arch.py:33-50,73-91
  bucketize_by_owner, gather_rows_segmented, mask_lengths   (new: row-sharded tables)
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_DENSE, NRX_ERR_UNSUPPORTED, NRX_FEAT_BAG_CSR, NRX_PLAN_PAIRS, NRX_PLAN_SPLIT_PADDING, NRX_FEAT_ROW0_IS_DATA, NRX_MAX_FEATURES, NRX_SPARSE,
                   NrxFeature, NrxFmGrad, check)

# ------------------------------------------------------------------------------- helpers
_INDEX_CHECK = "sync"      # "sync": raise IndexError in the offending call (reference behaviour on CPU; one device sync per call)
                           # "deferred": no sync, no copies -- the kernels record an offence in a host-mapped status word that
                           #             the NEXT call (and flush_index_checks()) reads as plain host memory
                           # "lazy": device status + event per call, checked once the event has passed; "off": never read back
_pending_status: List[Tuple[torch.Tensor, "torch.cuda.Event", Sequence[str]]] = []
_host_status: Optional[torch.Tensor] = None       # pinned int32[4] the GPU writes through its mapped address
_host_status_names: List[Sequence[str]] = []       # feature-name lists of the launches since the status word was last clear


def _collections_counter():
    import collections
    return collections.Counter()


def set_index_check(mode: str) -> None:
    global _INDEX_CHECK
    if mode not in ("sync", "deferred", "lazy", "off"):
        raise ValueError("index check mode must be 'sync', 'deferred', 'lazy' or 'off'")
    _INDEX_CHECK = mode


def _deferred_status(names: Sequence[str]) -> torch.Tensor:
    """The process-wide host-mapped status word of the 'deferred' mode.  An offence recorded by an EARLIER launch raises
    here (IndexError, one call late instead of never); the word is only ever written by a kernel that met a bad id, so
    the check is a read of four host integers.  Several plans may launch between two checks (DSSM's towers, the wide and
    deep halves): every launch's name list is remembered until the word has been seen clear again, and the report names
    the feature only as far as those lists agree."""
    global _host_status
    if _host_status is None:
        _host_status = torch.zeros(4, dtype=torch.int32).pin_memory()
    elif _host_status[0] != 0:
        _raise_deferred()
    # (the name lists are NOT trimmed when the word merely reads clear: with a deep launch queue an offence of a launch still in flight would
    # later be reported against the wrong names; the 16-entry ring bounds the list, flush_index_checks -- device idle -- empties it)
    if not any(n is names for n in _host_status_names):
        if len(_host_status_names) >= 16:      # launches still in flight are at most a few calls back
            del _host_status_names[0]
        _host_status_names.append(names)
    return _host_status


def _raise_deferred() -> None:
    # the kernel publishes the count (atomicAdd) BEFORE the first offender's feature / sample / id words: a non-zero
    # count can be visible with a stale payload.  This is the error path -- wait for the device, then read
    torch.cuda.synchronize()
    st = _host_status.tolist()
    _host_status.zero_()
    cands = []
    for names in _host_status_names:
        n = names[st[1]] if 0 <= st[1] < len(names) else None
        if n is not None and n not in cands:
            cands.append(n)
    del _host_status_names[:]
    if len(cands) == 1:
        fname = f"'{cands[0]}'"
    elif cands:
        fname = f"#{st[1]} of its launch (one of: {', '.join(repr(c) for c in cands)})"
    else:
        fname = f"#{st[1]}"
    raise IndexError(f"index out of range in self (reported by an earlier launch): {st[0]} lookup(s); first: feature "
                     f"{fname}, sample {st[2]}, id {st[3]}")


def _raise_if_oob(status: torch.Tensor, names: Sequence[str]) -> None:
    st = status.tolist()
    if st[0] != 0:
        fname = names[st[1]] if 0 <= st[1] < len(names) else f"#{st[1]}"
        raise IndexError(f"index out of range in self: {st[0]} lookup(s); first: feature '{fname}', "
                         f"sample {st[2]}, id {st[3]}")


def _remember_status_names(name_lists) -> None:
    """Re-register the feature-name lists of launches that will write the deferred status word without passing through _deferred_status
    (the kernels of a replayed HIP graph: graph.GraphedStep) -- so that a late IndexError can name the feature."""
    for names in name_lists:
        if not any(n is names for n in _host_status_names):
            if len(_host_status_names) >= 16:
                del _host_status_names[0]
            _host_status_names.append(names)


def deferred_index_error_pending() -> bool:
    """True when a launch has recorded an out-of-range id in the deferred status word since it was last cleared (a host read, no sync: the word
    may lag the device by the launches still in flight)."""
    return _host_status is not None and bool(_host_status[0] != 0)


def flush_index_checks() -> None:
    """Raise IndexError for any out-of-range id seen by earlier 'lazy' / 'deferred' calls (synchronises the device)."""
    if _host_status is not None:
        torch.cuda.synchronize()
        if _host_status[0] != 0:
            _raise_deferred()
        del _host_status_names[:]          # the device is idle and the word is clear: no launch can report against these lists any more
    while _pending_status:
        status, ev, names = _pending_status.pop(0)
        ev.synchronize()
        _raise_if_oob(status, names)


def _stream_ptr(t: torch.Tensor) -> int:
    """Raw handle of the current stream of t's device (one C call: torch.cuda.current_stream() costs ~6 us of Python per call)."""
    return torch._C._cuda_getCurrentRawStream(t.device.index)


_BIND = None        # the compiled host binding (csrc/nrx_bind.cpp), False once it is known to be absent


def _binding():
    """news_recsys_amd/lib/nrx_bind*.so, or None: the compiled form of the per-batch host work of the module path (tensor
    validation, descriptor refresh, output allocation, launch).  Optional -- without it the ctypes path below does the same work
    more slowly; NRX_NO_BIND=1 disables it (host-overhead A/B)."""
    global _BIND
    if _BIND is None:
        _BIND = False
        if os.environ.get("NRX_NO_BIND") != "1":
            try:
                import glob
                import importlib.util
                cand = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "nrx_bind*.so"))
                if cand:
                    spec = importlib.util.spec_from_file_location("nrx_bind", cand[0])
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    _BIND = mod
            except Exception:       # noqa: BLE001 -- an unloadable binding (other torch build) must not take the package down
                _BIND = False
    return _BIND or None


def _bound_plan(plan):
    """The plan's BoundPlan of the compiled binding (created once), or None."""
    bp = plan.__dict__.get("_bound", 0)
    if bp == 0:
        bp = None
        mod = _binding()
        if mod is not None and 0 < len(plan.slots) <= NRX_MAX_FEATURES:
            lib = _lib.load()
            addr = C.cast(lib.nrx_embed_fwd_train, C.c_void_p).value
            bp = mod.BoundPlan([(s.kind, s.table if s.kind != NRX_DENSE else -1, s.dim, s.bag_len, s.out_col, s.wide_col, s.fm_field, s.flags)
                                for s in plan.slots], plan.out_width, plan.wide_width, bool(plan.use_fm), addr)
        plan.__dict__["_bound"] = bp
    return bp


def _raw_stream(dev: torch.device) -> int:
    return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())


_stream_objs = {}


def _cur_stream(dev: torch.device) -> "torch.cuda.Stream":
    """torch.cuda.current_stream(dev), through a cache keyed by the raw handle (the Stream object is only needed for event calls)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    s = _stream_objs.get(key)
    if s is None:
        if len(_stream_objs) > 64:
            _stream_objs.clear()
        s = _stream_objs[key] = torch.cuda.current_stream(dev)
    return s


def _dev(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.NrxError(f"{what}: expected a ROCm device tensor, got device '{t.device}'. The HIP path has no CPU "
                            "fallback (move the module and the batch to cuda).")
    return t


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    _dev(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------- embed plan
@dataclass
class Slot:
    """One feature of a launch (host mirror of struct nrx_feature)."""
    name: str
    kind: int
    table: int          # index into the plan's table list; -1 for NRX_DENSE
    dim: int
    bag_len: int = 0
    out_col: int = 0
    wide_col: int = -1
    fm_field: int = 0
    flags: int = 0      # NRX_FEAT_* bits


@dataclass
class EmbedPlan:
    slots: List[Slot]
    out_width: int                 # columns written in `out`
    wide_width: int = 0
    use_fm: bool = False
    names: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.names = [s.name for s in self.slots]


def _fill_features(plan: EmbedPlan, lo: int, hi: int, tables: Sequence[torch.Tensor], inputs, weights,
                   table_ptrs: Optional[Sequence[int]] = None, fm: bool = True, cache_key: Optional[str] = None):
    """C descriptor array for slots [lo, hi).  With `cache_key` the array lives on the plan: the static
    fields (kind, dim, bag_len, columns, flags) are written once and a call only refreshes the pointers
    (the library copies the descriptors into the kernel arguments during the call, so reuse is safe)."""
    n = hi - lo
    cache = plan.__dict__.setdefault("_arr_cache", {}) if cache_key is not None else None
    key = (cache_key, lo, hi, fm)
    arr = cache.get(key) if cache is not None else None
    fresh = arr is None
    if fresh:
        arr = (NrxFeature * n)()
        if cache is not None:
            cache[key] = arr
    for i in range(n):
        s = plan.slots[lo + i]
        f = arr[i]
        idx = inputs[lo + i]
        if fresh:
            f.kind = s.kind
            f.dim = s.dim
            f.bag_len = s.bag_len
            f.out_col = s.out_col
            f.wide_col = s.wide_col
            f.fm_field = s.fm_field if fm else 0
            f.flags = s.flags
        f.index = idx.data_ptr()
        f.index_bits = idx.element_size() * 8
        if s.kind == NRX_DENSE:
            if fresh:
                f.table, f.rows = None, 0
        else:
            t = tables[s.table]
            f.table = table_ptrs[s.table] if table_ptrs is not None else t.data_ptr()
            f.rows = t.shape[0] if t is not None else 0
        w = weights[lo + i]
        f.weight = None if w is None else w.data_ptr()
    return arr


def _prep_inputs(plan: EmbedPlan, tables, inputs, weights):
    if len(inputs) != len(plan.slots) or len(weights) != len(plan.slots):
        raise ValueError("inputs / weights must have one entry per slot")
    B = None
    ins, ws = [], []
    i64, i32, f32, f64 = torch.int64, torch.int32, torch.float32, torch.float64
    for s, x, w in zip(plan.slots, inputs, weights):
        if not x.is_cuda:
            _dev(x, f"feature '{s.name}'")
        dt = x.dtype
        if s.flags & NRX_FEAT_BAG_CSR:
            # CSR bag: x = the concatenated ids of all bags [nnz], w = int64 offsets [B + 1] (no mask: every entry counts)
            if s.kind < NRX_BAG_MASKED_MEAN:
                raise ValueError(f"feature '{s.name}': NRX_FEAT_BAG_CSR on a non-bag feature")
            if w is None or w.dim() != 1 or w.numel() < 1 or w.dtype is not i64 or not w.is_cuda:
                raise ValueError(f"feature '{s.name}': a CSR bag needs device int64 offsets [B + 1] in place of the mask")
            if x.dim() != 1:
                raise ValueError(f"feature '{s.name}': CSR bag ids must be a 1-D [nnz] tensor, got {tuple(x.shape)}")
            if dt is not i64 and dt is not i32:
                x = x.long()
            if x.numel() == 0:
                x = torch.zeros(1, dtype=x.dtype, device=x.device)      # never read (every bag is empty); keeps the pointer valid
            nb = w.numel() - 1
            if B is None:
                B = nb
            elif nb != B:
                raise ValueError(f"feature '{s.name}': batch {nb} != {B}")
            ins.append(x.contiguous())
            ws.append(w.contiguous())
            continue
        if s.kind == NRX_DENSE:
            if dt is not f32 and dt is not f64:
                x = x.float()
        elif dt is not i64 and dt is not i32:
            x = x.long()                      # reference: feature_value.long()  (base_model.py:271)
        if not x.is_contiguous():
            x = x.contiguous()
        shp = x.shape
        if s.kind >= NRX_BAG_MASKED_MEAN:
            if len(shp) != 2 or shp[1] != s.bag_len:
                raise ValueError(f"feature '{s.name}': expected ids of shape [B, {s.bag_len}], got {tuple(shp)}")
        elif len(shp) != 1:
            raise ValueError(f"feature '{s.name}': expected a 1-D [B] tensor, got {tuple(shp)}")
        if B is None:
            B = shp[0]
        elif shp[0] != B:
            raise ValueError(f"feature '{s.name}': batch {shp[0]} != {B}")
        if w is not None:
            w = _f32c(w, f"mask of '{s.name}'")
            if w.shape != shp:
                raise ValueError(f"mask of '{s.name}': shape {tuple(w.shape)} != ids {tuple(shp)}")
        elif s.kind == NRX_BAG_MASKED_MEAN:
            raise ValueError(f"feature '{s.name}': masked mean needs a mask")
        ins.append(x)
        ws.append(w)
    for t in tables:
        if not (t.is_cuda and t.dtype is f32 and t.is_contiguous()):
            _f32c(t, "embedding table")
            raise ValueError("embedding tables must be contiguous [rows, dim] fp32")
    return B, ins, ws


def _csr_plan_to_padded(plan: EmbedPlan, inputs, weights):
    """The same launch with every CSR bag expanded to the reference's padded ids + mask (nrx_csr_to_padded): what the
    row-sparse backward's planner takes.  The derived plan is cached on the plan."""
    import dataclasses
    padded = plan.__dict__.get("_padded_plan")
    if padded is None:
        padded = EmbedPlan([dataclasses.replace(s, flags=s.flags & ~NRX_FEAT_BAG_CSR) for s in plan.slots],
                           out_width=plan.out_width, wide_width=plan.wide_width, use_fm=plan.use_fm)
        plan.__dict__["_padded_plan"] = padded
    inputs, weights = list(inputs), list(weights)
    for i, s in enumerate(plan.slots):
        if s.flags & NRX_FEAT_BAG_CSR:
            ids, mask = csr_to_padded(inputs[i], weights[i], s.bag_len)
            inputs[i], weights[i] = ids, (None if s.kind == NRX_BAG_MEAN else mask)
    return padded, inputs, weights


def _table_meta(ctx):
    """[(shape, device)] of the node's tables, formed on first use in the backward (26 tables x 2 attribute reads are ~8 us of
    host time that an inference-only or discarded forward never needs)."""
    tm = getattr(ctx, "_tm", None)
    if tm is None:
        tm = ctx._tm = [(t.shape, t.device) for t in ctx.tables_ref]
    return tm


def _zero_grad_tables(meta):
    """Zero-filled dense gradient tables: ONE allocation and ONE fill for all of them when they live on one device (26 tables
    were 26 fill launches, ~100 us of host time per step); each table a view of it (row-aligned when all share one row width,
    else 256-byte aligned)."""
    if len(meta) < 2 or any(d != meta[0][1] for _, d in meta):
        return [torch.zeros(shape, dtype=torch.float32, device=d) for shape, d in meta]
    if all(len(shape) == 2 and shape[1] == meta[0][0][1] for shape, _ in meta):      # one row width: one split, no per-table slicing
        rows = [shape[0] for shape, _ in meta]
        return list(torch.zeros((sum(rows), meta[0][0][1]), dtype=torch.float32, device=meta[0][1]).split_with_sizes(rows))
    offs, tot = [], 0
    for shape, _ in meta:
        offs.append(tot)
        tot += (math.prod(shape) + 63) & ~63
    buf = torch.zeros(tot, dtype=torch.float32, device=meta[0][1])
    return [buf[o:o + math.prod(shape)].view(shape) for o, (shape, _) in zip(offs, meta)]


def _dense_sorted_ok(plan, tables, sparse_grad, B, csr_ok=False) -> bool:
    """Default (dense-gradient) mode: form the table grads by the sorted reduction + nrx_rows_to_dense?  The sorted path costs a fixed
    ~25 launches (~0.2 ms of host and launch time per step), the atomic scatter ~0.2 us per 1000 lookups: from DENSE_SORTED_MIN lookups
    per launch on (786 k: the C4 tower from B = 15 k) the sorted path is as fast or faster, below it the single atomic launch is
    (profiles/r03_dense_backward.txt, r05_dense_default.txt).  DENSE_BWD_SORTED: True = always (deterministic gradients at any size), False = never."""
    if sparse_grad or DENSE_BWD_SORTED is False or B <= 0 or not tables or not tables[0].is_cuda:
        return False
    # (while a HIP graph is being captured the plan is made inline in the backward -- no side stream, no events to query -- and the reduction
    # reads the unique-row count on the device: nothing in the sorted path reads back, so it captures like the atomic launch does)
    if len(tables) > NRX_MAX_FEATURES:          # the dense destination names a table by its slot in a <= 64-entry argument array
        return False
    lookups = 0
    for s in plan.slots:
        if s.kind == NRX_DENSE:
            continue
        if s.flags & ((0 if csr_ok else NRX_FEAT_BAG_CSR) | NRX_FEAT_ROW0_IS_DATA):      # (the planner's row 0 never trains)
            return False
        lookups += max(1, s.bag_len)
    if lookups <= 0:
        return False
    if DENSE_BWD_SORTED in (True, "det") or B * lookups >= DENSE_SORTED_MIN:
        return True
    # launches the one-kernel planner may take (single-valued features over mid-size tables): the planned reduction is ONE plan launch + placement +
    # walk there -- within 1.03-1.10x of the atomic step from a few thousand samples on, faster from ~30 k (tools/ab_dense_default.py,
    # profiles/r05_dense_default.txt) -- and bit-reproducible: the default from DENSE_LDS_MIN lookups on
    if PLAN_LDS == "0" or not SPARSE_PLACE or B * lookups < max(DENSE_LDS_MIN, 1) or any(s.kind not in (NRX_SPARSE, NRX_DENSE) for s in plan.slots):
        return False
    if DENSE_SMALL_DET and _small_shapes(plan, B):
        return False                     # (every table fed by <= 4096 lookups: the one-launch deterministic kernel's)
    return all(g["all_sparse"] and _group_policy(g, B, len(tables)) is not None and _group_policy(g, B, len(tables)).eligible
               for g in _sparse_group_cache(plan, tables))


# which form produced the dense table gradients, counted per backward launch: "small" (one-launch deterministic kernel), "sorted" (planned
# reduction), "atomic" (float atomics).  GraphedStep(deterministic=True) checks that its capture took no "atomic" launch.
dense_bwd_paths = _collections_counter()
_atomic_warned = set()


def _warn_atomic_fallback(plan, n_tables: int, B: int) -> None:
    why = []
    if n_tables > NRX_MAX_FEATURES:
        why.append(f"{n_tables} tables (the planned reduction names a table in a {NRX_MAX_FEATURES}-entry argument array)")
    if any(s_.flags & NRX_FEAT_ROW0_IS_DATA for s_ in plan.slots):
        why.append("a feature reads a routed-row buffer whose row 0 is data (the planner's row 0 never trains)")
    if not why:
        why.append("the launch is outside both deterministic forms")
    key = "; ".join(why)
    if key not in _atomic_warned:
        _atomic_warned.add(key)
        import warnings
        warnings.warn("NRX_DENSE_BWD / GraphedStep(deterministic=True): this backward launch (batch %d) falls back to float atomics -- its gradients are "
                      "NOT bit-reproducible: %s" % (B, key), UserWarning, stacklevel=3)


def _small_shapes(plan, B) -> bool:
    """Every table of the launch fed by <= 4096 lookups (the one-block-per-table kernels' limit; the library has the last word)."""
    if B > 4096 or B <= 0:
        return False
    memo = plan.__dict__.setdefault("_small_shapes_memo", {})
    if B in memo:
        return memo[B]
    memo[B] = res = _small_shapes_uncached(plan, B)
    return res


def _small_shapes_uncached(plan, B) -> bool:
    per = {}
    for s in plan.slots:
        if s.kind == NRX_DENSE:
            continue
        if s.flags & NRX_FEAT_BAG_CSR or s.dim > 256:
            return False
        per[s.table] = per.get(s.table, 0) + B * max(1, s.bag_len)
    return bool(per) and max(per.values()) <= 4096


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan: EmbedPlan, inputs, weights, out_ld, need_out, sparse_grad, index_check, *tables):
        lib = _lib.load()
        mode = index_check or _INDEX_CHECK
        ctx.sink = sparse_grad if isinstance(sparse_grad, SparseGradSink) else None
        ctx.sparse_grad = bool(sparse_grad)
        ctx.tables = list(tables) if ctx.sink is not None else None
        if sparse_grad and any(s.flags & NRX_FEAT_ROW0_IS_DATA for s in plan.slots):
            raise NotImplementedError("sparse_grad: the sorted backward treats row 0 of every table as the padding row; "
                                      "a feature flagged NRX_FEAT_ROW0_IS_DATA (routed-row buffers) needs the dense-gradient mode")
        if sparse_grad and any(s.flags & NRX_FEAT_BAG_CSR for s in plan.slots):
            plan, inputs, weights = _csr_plan_to_padded(plan, inputs, weights)      # the planner sorts padded [B, L] lookups
        narrow = bool(out_ld) and int(out_ld) < 0          # embed_apply(narrow=True): a padded row stride, the caller gets the [B, out_width] view
        out_ld = abs(int(out_ld)) if out_ld else None
        ctx.narrow = narrow
        ld = int(out_ld) if out_ld else plan.out_width
        if ld < plan.out_width:
            raise ValueError("out_ld smaller than the plan's out_width")
        if narrow and plan.use_fm:
            raise ValueError("narrow=True is for plans without the FM epilogue (its backward reads the forward concat at the forward's stride)")
        n_slots = len(plan.slots)
        bp = _bound_plan(plan) if tables and tables[0].is_cuda else None
        done = None
        if bp is not None and (need_out or not plan.use_fm or n_slots <= NRX_MAX_FEATURES):
            # compiled binding: validation + descriptors + allocation + launch in one call (None: a batch that needs a conversion)
            dev = tables[0].device
            status = None
            if mode == "deferred":
                sp = _deferred_status(plan.names).data_ptr()
            elif mode == "off":
                sp = 0
            else:
                status = torch.zeros(4, dtype=torch.int32, device=dev)
                sp = status.data_ptr()
            # (autograd only builds this node when some table requires grad: the field sums are always wanted in training form)
            done = bp.forward(list(tables), inputs, weights, ld, need_out, sp, _raw_stream(dev), bool(plan.use_fm and need_out))
            if type(done) is int:
                check(done, "nrx_embed_fwd_train")
        if done is not None:
            B, out, wide, fm, sums = done
            ins, ws = inputs, weights
            scratch_out = False
        else:
            B, ins, ws, dev, out, wide, fm, sums, status, scratch_out = _EmbedFn._forward_ctypes(
                lib, plan, tables, inputs, weights, ld, need_out, mode)
        if status is not None and mode != "deferred":
            if mode == "sync":
                _raise_if_oob(status, plan.names)
            else:
                ev = torch.cuda.Event()
                ev.record()
                _pending_status.append((status, ev, plan.names))
                while _pending_status and _pending_status[0][1].query():
                    st, _, nm = _pending_status.pop(0)
                    _raise_if_oob(st, nm)
        return _EmbedFn._finish(ctx, plan, tables, B, ld, ins, ws, out, wide, fm, sums, need_out, scratch_out)

    @staticmethod
    def _forward_ctypes(lib, plan, tables, inputs, weights, ld, need_out, mode):
        """The launch through ctypes (any batch: converts what needs converting)."""
        B, ins, ws = _prep_inputs(plan, tables, inputs, weights)
        dev = tables[0].device if tables else ins[0].device
        n_slots = len(plan.slots)
        # > 64 FM fields: the cross-field sums span launches, so the FM runs on the concat -- it is needed even when the
        # caller did not ask for it (returned as None all the same)
        scratch_out = plan.use_fm and n_slots > NRX_MAX_FEATURES and not need_out
        out = torch.empty((B, ld), dtype=torch.float32, device=dev) if (need_out or scratch_out) else None
        wide = torch.empty((B, plan.wide_width), dtype=torch.float32, device=dev) if plan.wide_width else None
        fm = torch.empty((B,), dtype=torch.float32, device=dev) if plan.use_fm else None
        if mode == "deferred":
            status = _deferred_status(plan.names)
        else:
            status = torch.zeros(4, dtype=torch.int32, device=dev) if mode != "off" else None
        stream = _stream_ptr(ins[0])
        n = len(plan.slots)
        single = n <= NRX_MAX_FEATURES
        # training with the FM epilogue: the forward also leaves the per-sample field sums, so that the backward can form
        # d fm / d field on the fly (nrx_fm_grad_t) -- no separate FM backward pass, no [B, sum D] temporary
        fm_dim = max((s.dim for s in plan.slots if s.fm_field), default=0)
        sums = (torch.empty((B, fm_dim), dtype=torch.float32, device=dev)
                if plan.use_fm and single and need_out and fm_dim and any(t.requires_grad for t in tables) else None)
        if B > 0:
            for lo in range(0, n, NRX_MAX_FEATURES):
                hi = min(n, lo + NRX_MAX_FEATURES)
                arr = _fill_features(plan, lo, hi, tables, ins, ws, fm=single, cache_key="fwd")
                check(lib.nrx_embed_fwd_train(arr, hi - lo, B, _ptr(out), ld, _ptr(wide), plan.wide_width,
                                              _ptr(fm) if single else None, _ptr(sums), fm_dim, _ptr(status), stream),
                      "nrx_embed_fwd_train")
            if plan.use_fm and not single:
                # > 64 FM fields: the cross-field sums cannot be split over launches; run FM on the concat
                d0 = plan.slots[0].dim
                check(lib.nrx_fm_fwd(out.data_ptr(), ld, n, d0, B, fm.data_ptr(), stream), "nrx_fm_fwd")
        return B, ins, ws, dev, out, wide, fm, sums, status, scratch_out

    @staticmethod
    def _finish(ctx, plan, tables, B, ld, ins, ws, out, wide, fm, sums, need_out, scratch_out):
        ctx.plan, ctx.B, ctx.ld = plan, B, ld
        ctx.ins, ctx.ws = ins, ws
        ctx.tables_ref = tables          # leaves (the module's parameters): shapes / devices are read from them in the backward
        # the FM backward needs the forward concat.  It is an OUTPUT of this node: it must go through save_for_backward --
        # a plain attribute would make a reference cycle (ctx -> out -> grad_fn -> ctx) that only the garbage collector
        # frees: 100+ MB per step kept alive, and a ~35 ms collection every few dozen steps
        ctx.has_fm_feat = bool(plan.use_fm and need_out)
        ctx.fm_sums = sums
        ctx.plans = None
        # default (dense-gradient) mode: the table grads are formed by the sorted reduction + nrx_rows_to_dense unless
        # NRX_DENSE_BWD=atomic, the launch is being captured (the planner allocates) or a feature reads a routed-row buffer
        # (needs_input_grad follows the grad MODE too: a forward under torch.no_grad() -- leaf tables still say requires_grad -- plans nothing)
        wants_grad = any(ctx.needs_input_grad[7:])
        ctx.dense_sorted = wants_grad and _dense_sorted_ok(plan, tables, ctx.sparse_grad, B, csr_ok=True)
        if ctx.dense_sorted and any(s.flags & NRX_FEAT_BAG_CSR for s in plan.slots):
            # the forward ran on the CSR form; the backward's planner sorts padded [B, L] lookups (nrx_csr_to_padded: one small launch
            # per bag feature) -- from here on the node only describes the backward
            plan, ins, ws = _csr_plan_to_padded(plan, ins, ws)
            ctx.plan, ctx.ins, ctx.ws = plan, ins, ws
        small = ctx.dense_sorted and not ctx.sparse_grad and B * sum(max(1, s_.bag_len) for s_ in plan.slots if s_.kind != NRX_DENSE) < PLAN_AHEAD_MIN
        if ctx.sink is not None and SPARSE_SMALL_DET and _small_shapes(plan, B):
            small = True                 # the sink's one-launch form (nrx_embed_bwd_small_sparse) plans nothing
        if (ctx.sparse_grad or ctx.dense_sorted) and PLAN_AHEAD and B > 0 and not torch.cuda.is_current_stream_capturing() and wants_grad and not small:
            ctx.plans = {}
            for g_ in _sparse_group_cache(plan, tables):
                fs_ = g_["fs"]
                pol_ = _group_policy(g_, B, len(tables))
                if pol_ is not None and pol_.choose() and not PLAN_AHEAD_LDS:
                    continue                 # the one-kernel planner: ONE 43 us launch that holds every compute unit -- next to the forward the two
                                             # only contend (C2: 242 us per step against 214 planned inline, profiles/r05_plan_lds.txt): planned in the backward
                ids_ = [ins[i] for i in fs_]
                dt_ = ids_[0].dtype
                for x in ids_:
                    if x.dtype is not dt_:
                        ids_ = [y.long() for y in ids_]
                        break
                ctx.plans[(g_["dim"], fs_[0])] = (ids_,) + sparse_plan_ahead(ids_, g_["tabs"], g_["rows"], len(tables),
                                                                          g_["pmask"] if SPARSE_PLACE else None, static=g_["static"],
                                                                          policy=pol_, pad=_group_pad(g_, plan, B), pairs=g_["all_sparse"])
        if ctx.has_fm_feat:
            ctx.save_for_backward(out)
        ctx.set_materialize_grads(False)
        if getattr(ctx, "narrow", False) and out is not None and not scratch_out:
            out = out.narrow(1, 0, plan.out_width)      # the node's output IS the view: its gradient arrives as [B, out_width]
        return (None if scratch_out else out), wide, fm

    @staticmethod
    def backward(ctx, g_out, g_wide, g_fm):
        lib = _lib.load()
        plan, B, ld = ctx.plan, ctx.B, ctx.ld
        n_tables = len(_table_meta(ctx))
        if g_out is None and g_wide is None and g_fm is None:
            return (None,) * (7 + n_tables)
        dev = _table_meta(ctx)[0][1]
        stream = _raw_stream(dev)
        if g_out is not None:
            g_out = _f32c(g_out, "grad of the concat")
            if getattr(ctx, "narrow", False):
                ld = ctx.ld = g_out.shape[1]            # the upstream gradient of the narrowed output is [B, out_width], contiguous
        fmg = None
        if g_fm is not None:
            fm_feat = ctx.saved_tensors[0] if getattr(ctx, "has_fm_feat", False) else None
            if fm_feat is None:
                raise RuntimeError("FM backward needs the forward concat: call with need_out=True when training")
            g_fm = _f32c(g_fm, "grad of fm_out")
            sums = getattr(ctx, "fm_sums", None)
            if sums is not None:               # folded into the embedding backward (nrx_fm_grad_t)
                fmg = NrxFmGrad(g_fm.data_ptr(), sums.data_ptr(), sums.shape[1], fm_feat.data_ptr(), ld)
            else:                              # > 64 FM fields: the sums span launches; FM backward on the concat
                d0 = plan.slots[0].dim
                g_in = g_out                   # upstream gradient of the concat: read, never written (no clone)
                g_out = torch.empty((B, ld), dtype=torch.float32, device=dev)
                check(lib.nrx_fm_bwd(fm_feat.data_ptr(), ld, len(plan.slots), d0, B, g_fm.data_ptr(), _ptr(g_in), ld,
                                     g_out.data_ptr(), ld, stream), "nrx_fm_bwd")
        if g_wide is not None:
            g_wide = _f32c(g_wide, "grad of wide_x")
        if ctx.sink is not None:
            _sorted_sparse_grads(ctx, lib, g_out, g_wide, stream, fmg)      # results go to the sink, not to .grad
            return (None,) * (7 + n_tables)
        if ctx.sparse_grad:
            return (None, None, None, None, None, None, None, *_sorted_sparse_grads(ctx, lib, g_out, g_wide, stream, fmg))
        grads = _zero_grad_tables(_table_meta(ctx))
        has_up = g_out is not None or g_wide is not None or fmg is not None
        n = len(plan.slots)

        def small(lo, hi, arr):
            """The reference's own batch sizes: ONE deterministic launch (nrx_embed_bwd_small: block per table, the lookups of rows hit more
            than once sorted in LDS, in-order sums) when every table of the launch is fed by <= 4096 lookups; NRX_ERR_UNSUPPORTED (nothing
            enqueued) sends the launch on to the caller's other path."""
            if not (DENSE_SMALL_DET and B <= 4096):
                return False
            rc = lib.nrx_embed_bwd_small(arr, hi - lo, B, _ptr(g_out), ld, _ptr(g_wide), plan.wide_width, fmg, 1 if lo else 0, stream)
            if rc != 0 and rc != NRX_ERR_UNSUPPORTED:
                check(rc, "nrx_embed_bwd_small")
            return rc == 0

        if getattr(ctx, "dense_sorted", False) and B > 0 and n_tables <= NRX_MAX_FEATURES:
            # "deterministic" (GraphedStep(deterministic=True)): the small kernel where it applies (a single launch of <= 64 features), else
            # the planned reduction -- both bit-reproducible run to run; "sorted" always takes the planned reduction
            if (DENSE_BWD_SORTED == "det" and has_up and n <= NRX_MAX_FEATURES and
                    small(0, n, _fill_features(plan, 0, n, grads, ctx.ins, ctx.ws, table_ptrs=[g.data_ptr() for g in grads], cache_key="bwd"))):
                dense_bwd_paths["small"] += 1
            else:
                dense_bwd_paths["sorted"] += 1
                _sorted_sparse_grads(ctx, lib, g_out, g_wide, stream, fmg, dense_into=grads)
        elif B > 0 and has_up:
            gptrs = [g.data_ptr() for g in grads]
            for lo in range(0, n, NRX_MAX_FEATURES):
                hi = min(n, lo + NRX_MAX_FEATURES)
                arr = _fill_features(plan, lo, hi, grads, ctx.ins, ctx.ws, table_ptrs=gptrs, cache_key="bwd")
                if DENSE_BWD_SORTED is not False and small(lo, hi, arr):
                    dense_bwd_paths["small"] += 1
                    continue
                if DENSE_BWD_SORTED in (True, "det"):
                    # a deterministic mode was asked for and neither deterministic form serves this launch (more than 64 tables, a routed-row
                    # buffer whose row 0 is data, CSR bags / dims outside the small kernel): say so -- once per reason -- instead of silently
                    # handing out gradients that differ in the last place from run to run
                    _warn_atomic_fallback(plan, len(_table_meta(ctx)), B)
                dense_bwd_paths["atomic"] += 1
                check(lib.nrx_embed_bwd(arr, hi - lo, B, _ptr(g_out), ld, _ptr(g_wide), plan.wide_width, fmg, stream),
                      "nrx_embed_bwd")
        return (None, None, None, None, None, None, None, *grads)


class SparseGradSink:
    """Where the row-sparse backward leaves its result when the optimizer is fused (optim.FusedSparseAdam):
    per launch group the device-resident (unique keys, row gradients, counts) of nrx_sparse_plan /
    nrx_embed_bwd_sorted plus the table tensors the keys index -- no COO tensors, no host synchronisation.
    Pass the sink as `sparse_grad=` to embed_apply; the optimizer drains it in step()."""

    def __init__(self):
        self.pending = []       # dicts: tables (list of tensors, index = table id in the keys), dim, uniq, values, counts, cap
                                # (filler=True: no counts -- unused slots anywhere in uniq carry key -1: nrx_embed_bwd_small_sparse's layout)

    def clear(self):
        self.pending.clear()


SPARSE_BWD_SYNC_FREE = False   # True: size the reduction for the worst case and read the count on the device


_plan_streams = {}
# default-mode (dense) table grads.  NRX_DENSE_BWD = auto (default: sorted reduction from DENSE_SORTED_MIN lookups per launch on, float atomics
# below -- except that launches whose tables each take <= 4096 lookups, the reference's own batch sizes, use the one-launch deterministic kernel
# nrx_embed_bwd_small) | sorted (always the planned reduction: bit-reproducible gradients at any batch) | deterministic (bit-reproducible by
# the cheaper of the two: the small kernel where it applies, else the planned reduction) | atomic (never).
# DENSE_BWD_SORTED: None = auto, True = sorted, "det" = deterministic, False = atomic.
DENSE_BWD_SORTED = {"sorted": True, "atomic": False, "deterministic": "det"}.get(os.environ.get("NRX_DENSE_BWD", "auto"))
DENSE_SORTED_MIN = int(os.environ.get("NRX_DENSE_SORTED_MIN", 3 << 18))      # 786 k lookups: from there the planned reduction replays as fast as the atomic
                                                                            # launch on the C4 tower (B = 14 336: 125 vs 118 us, 16 384: 128 vs 130;
                                                                            # profiles/r05_dense_default.txt) and is bit-reproducible
SPARSE_SMALL_DET = os.environ.get("NRX_SPARSE_SMALL", "1") != "0"     # fused row-sparse mode (sink), small launches: the one-launch form likewise
DENSE_SMALL_DET = os.environ.get("NRX_DENSE_SMALL", "1") != "0"       # auto mode, small launches: the one-launch deterministic kernel where it applies
PLAN_AHEAD = True      # row-sparse training: plan the backward (sort, unique rows, segments) at FORWARD time on a side stream
DENSE_LDS_MIN = int(os.environ.get("NRX_DENSE_LDS_MIN", 1 << 16))       # default-mode launches from this many lookups on go through the one call that
                                                                         # takes the planner as an argument (nrx_embed_bwd_dense_planned)
SINK_ONE_CALL = os.environ.get("NRX_SINK_ONE_CALL", "1") != "0"           # sink mode, inline plans: plan + reduction as ONE library call (A/B knob)
PLAN_AHEAD_LDS = os.environ.get("NRX_PLAN_AHEAD_LDS", "0") != "0"      # plan ahead (side stream) even when the group takes the one-kernel planner
PLAN_AHEAD_MIN = int(os.environ.get("NRX_PLAN_AHEAD_MIN", 1 << 20))      # default-mode launches below this many lookups plan inline (one fused call: the
                                                                         # side-stream path's allocations and events were 240 us of host time per step
                                                                         # at B = 8192 on the C4 tower against 123 inline)


def _plan_stream(dev) -> "torch.cuda.Stream":
    s = _plan_streams.get(dev)
    if s is None:
        s = _plan_streams[dev] = torch.cuda.Stream(device=dev)
    return s


import collections as _collections
_plan_keepalive = _collections.deque()


# Planner of the row-sparse backward.  NRX_PLAN_LDS = auto (default) | 1 | 0.  The one-kernel planner (nrx_sparse_plan_lds: row bitmaps in LDS, no
# sort; C2: 43 us against 74-80) is FAST only on near-unique id batches -- rows looked up three times or more are sorted by single blocks there.
# auto: every plan leaves its duplicate statistics in mapped host memory (no copy, no synchronisation: the kernel writes them, the host reads
# whatever the last finished plan left), and the NEXT batch of the launch group takes the one-kernel planner iff the last one was near-unique.
PLAN_LDS = os.environ.get("NRX_PLAN_LDS", "auto")
_lds_states = {}


_aux_streams = {}
PAIRS_AUX = os.environ.get("NRX_PAIRS_AUX", "0") != "0"      # the pair pass / walk of a one-kernel plan on a second stream, next to the placement
                                                              # pass: measured, LOSES (C2 213.8 -> 225.1 us: the small launches are chains of dependent
                                                              # round trips and every trip takes longer under the placement pass's traffic); off


def _aux_stream(dev: torch.device) -> Optional[int]:
    """The stream nrx_embed_bwd_placed_pairs forks its small launches onto (one per device)."""
    if not PAIRS_AUX:
        return None
    k = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _aux_streams.get(k)
    if s is None:
        s = _aux_streams[k] = torch.cuda.Stream(device=dev)
    return s.cuda_stream


def _lds_state(dev: torch.device, stream: int) -> Optional[torch.Tensor]:
    """The planner's persistent control block (ticket, epoch, per-block totals), one per (device, stream): zero before the first use."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(stream))
    st = _lds_states.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None                      # (a captured step uses the planner only if an eager step made the block before)
        st = torch.zeros(_lib.load().nrx_sparse_plan_lds_state_bytes(), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)          # once: the fill ran on the current stream, the planner may run on another
        _lds_states[key] = st
    return st


_PINNED_CHUNKS = []           # page-locked int64 chunks, NEVER released: see _pinned_words
_PINNED_USED = 0


def _pinned_words(n: int):
    """n int64 words of page-locked host memory that stay valid for the life of the process: (tensor view, address, numpy view).
    The planners' statistics words are written by kernels through a raw address -- which is also baked into captured graphs -- so the memory
    must outlive every launch and every replay that may still name it.  A pinned tensor owned by a policy object went back to torch's pinned
    caching allocator when the policy was replaced (another batch size) or evicted, and a later plan or graph replay then wrote 32-40 bytes
    into whatever that block had become (a ColumnarLoader staging batch, a status word).  Slices of an arena that is never freed cannot be
    recycled; a policy costs 64 bytes of it."""
    global _PINNED_USED
    n = int(n)
    room = (n + 7) & ~7                       # 64-byte granules
    if not _PINNED_CHUNKS or _PINNED_USED + room > _PINNED_CHUNKS[-1].numel():
        _PINNED_CHUNKS.append(torch.zeros(max(4096, room), dtype=torch.int64).pin_memory())
        _PINNED_USED = 0
    t = _PINNED_CHUNKS[-1][_PINNED_USED:_PINNED_USED + n]
    _PINNED_USED += room
    return t, t.data_ptr(), t.numpy()


class PlanPolicy:
    """Per launch group: may the one-kernel planner take it (shape), and should it (the previous batch's duplicate statistics)."""

    def __init__(self, lens, tof, rws, n, n_tables, total, all_sparse=True):
        lib = _lib.load()
        self.total = int(total)
        self.eligible = bool(all_sparse and PLAN_LDS != "0" and total > 0 and lib.nrx_sparse_plan_lds_ok(lens, tof, rws, n, n_tables))
        self.use_lds = PLAN_LDS == "1"
        self.stats = None
        if self.eligible:
            self.stats, self.stats_ptr, self._np = _pinned_words(4)

    def choose(self) -> bool:
        if not self.eligible:
            return False
        if PLAN_LDS == "1":
            return True
        uniq, walk_rows, walk_look, n = (int(x) for x in self._np)
        if n != self.total:
            return self.use_lds              # nothing recorded yet for this batch size: the sorted planner
        if walk_look < 0:                    # recorded by the sorted planner: rows that are not alone ~ walk rows, lookups not counted
            self.use_lds = (n - uniq + walk_rows) * 10 <= n
        else:                                # by the one-kernel planner: lookups of rows looked up 3+ times, and all duplicates
            self.use_lds = walk_look * 32 <= n and (n - uniq) * 8 <= n
        return self.use_lds


PLAN_PAIRS = os.environ.get("NRX_PLAN_PAIRS", "0") != "0"              # sorted planner: rows looked up exactly twice as pair records (as the one-kernel
                                                                       # planner leaves them) for launches of single-valued features.  Off: measured, it
                                                                       # is not a win there -- the emit kernel's extra loads and the pair pass cost what
                                                                       # the shorter walk saves (C5 394.6 -> 398.0 us, Zipf C2 419.4 -> 415.7)
PAD_SPLIT = os.environ.get("NRX_PAD_SPLIT", "auto")                     # auto: by the previous batch's share of padding lookups; 1 / 0: always / never
PAD_SPLIT_MIN = int(os.environ.get("NRX_PAD_SPLIT_MIN", 3 << 19))      # lookups per launch group from which the split is considered (1.5 M)


class PadPolicy:
    """Per launch group with multi-valued features: should the planner set the padding lookups aside before it sorts
    (NRX_PLAN_SPLIT_PADDING)?  Histories padded to max_len with id 0 (the reference's DataReader) are half padding -- the split plans
    3.4 M such lookups in 114 us instead of 142 -- but a launch without padding loses ~17 us to it, and only the data knows: every plan leaves
    its count of padding lookups in a mapped host word (no synchronisation), the next batch of the same size decides by it -- from a quarter
    of the lookups on."""

    def __init__(self, total: int):
        self.total = int(total)
        self.stats, self.stats_ptr, self._np = _pinned_words(5)

    def choose(self) -> bool:
        if PAD_SPLIT != "auto":
            return PAD_SPLIT == "1"
        n, pads = int(self._np[3]), int(self._np[4])
        return n == self.total and pads * 4 >= n

    def stats_arg(self):
        """The statistics pointer for this plan, or None: the count is a launch of its own (a few microseconds behind every plan), and the share of
        padding in a data set does not move from batch to batch -- taken for the first plans and then every 16th."""
        self._calls = getattr(self, "_calls", 0) + 1
        if self._calls <= 4 or self._calls % 16 == 0 or int(self._np[3]) != self.total:
            return self.stats_ptr
        return None


def sparse_plan_ahead(ids: Sequence[torch.Tensor], table_of: Sequence[int], rows: Sequence[int], n_tables: int,
                      place_feats: Optional[int] = None, static=None, policy: Optional["PlanPolicy"] = None, pad: Optional["PadPolicy"] = None,
                      pairs: bool = False):
    """sparse_plan on a side stream: the planning of the backward depends only on the ids, so it can run while the forward,
    the dense part of the model and its backward occupy the main stream.  Returns ((order, uniq, seg, counts), event);
    the consumer makes its stream wait for `event` before reading the plan."""
    dev = ids[0].device
    cur = _cur_stream(dev)
    side = _plan_stream(dev)
    side.wait_stream(cur)                      # the ids are ready where the caller's stream is now
    # The library takes the stream as an argument: the plan is enqueued on the side stream without making it PyTorch's current stream
    # (`with torch.cuda.stream(side)` and the current_stream() lookups around it were ~50 us of host time per step).  The outputs and the
    # workspace are therefore allocated under the CALLER's stream, after the wait above was recorded: whatever used their blocks before
    # was enqueued on the caller's stream before that point, so the side stream's writes are ordered behind it; they are consumed (and
    # later freed) on the caller's stream behind `ev`.  No record_stream anywhere.
    hold = []
    res = sparse_plan(ids, table_of, rows, n_tables, place_feats, static=static, stream=side.cuda_stream, keep=hold, policy=policy, pad=pad, pairs=pairs)
    ev = torch.cuda.Event()
    ev.record(side)
    # The planner READS the ids on the side stream: an id tensor that dies early (a .long() / .contiguous() temporary, a
    # csr_to_padded output, a forward whose loss is dropped without a backward) must not have its block handed out again on the
    # caller's stream while the sort still reads it.  One reference per plan, held until the plan's event has passed (26
    # record_stream calls were 40 us of host time per step).  The same reference keeps the plan's own tensors (and, through
    # sparse_plan's return value, nothing else) alive when a forward is never followed by a backward.
    _plan_keepalive.append((ev, ids, res, hold))
    while _plan_keepalive and _plan_keepalive[0][0].query():
        _plan_keepalive.popleft()
    return res, ev


SPARSE_PLACE = os.environ.get("NRX_SPARSE_PLACE", "1") != "0"      # row-sparse backward: place single-lookup rows (A/B knob)


def place_mask(kinds: Sequence[int], bag_lens: Optional[Sequence[int]] = None) -> Optional[int]:
    """Bit f set for every single-valued feature: the lookups nrx_sparse_plan_place may place (a bag lookup is scaled).
    None (no placement) when the single-valued features are under a quarter of the launch's lookups (`bag_lens` given): the
    placement outputs cost the plan ~13 us (C4: emit 13.7 -> 22, head count 6 -> 10) and then buy one 5 us pass over 4 % of the rows."""
    m = 0
    for i, k in enumerate(kinds):
        if k == NRX_SPARSE:
            m |= 1 << i
    if bag_lens is not None:
        n_sparse = sum(1 for k in kinds if k == NRX_SPARSE)
        total = sum(1 if k == NRX_SPARSE else max(int(L), 1) for k, L in zip(kinds, bag_lens))
        if n_sparse * 4 < total:
            return None
    return m if m else None


def sparse_plan(ids: Sequence[torch.Tensor], table_of: Sequence[int], rows: Sequence[int], n_tables: int,
                place_feats: Optional[int] = None, static=None, stream: Optional[int] = None, keep: Optional[list] = None,
                policy: Optional["PlanPolicy"] = None, pad: Optional["PadPolicy"] = None, pairs: bool = False):
    """nrx_sparse_plan: group the flat, feature-major lookups `ids` (one device tensor per feature, all int32 or
    all int64) by (table, row).  Returns device int64 tensors (order [n], uniq_keys [n], seg_start [n+1],
    counts [n_tables+2]); only the first counts[0] entries of uniq_keys / counts[0]+1 of seg_start are
    meaningful.  No host synchronisation.
    place_feats (bit mask over the features, see place_mask): nrx_sparse_plan_place -- three more device tensors, dest
    int32 [n] (unique index of a lookup whose row is looked up once, else -1), walk int32 [n] (the other unique rows)
    and n_walk int64 [1]: what nrx_embed_bwd_placed consumes.
    policy (PlanPolicy of the launch group): when it picks the one-kernel planner (nrx_sparse_plan_lds) the result has EIGHT entries -- the
    seven above (order / seg defined for the walk rows only; n_walk int64 [2]: walk rows, pair records) plus pairs int32 [n / 2 + 1, 4], the
    records {unique index, first lookup, second lookup, 0} of the rows looked up twice -- and goes to nrx_embed_bwd_placed_pairs.
    pad (PadPolicy of a launch group with multi-valued features): nrx_sparse_plan_ex -- the same plan, the padding lookups set aside before the
    sort when the previous batch held enough of them, this batch's count left for the next one.
    pairs (every feature single-valued and named in place_feats): the sorted planner also leaves the rows looked up exactly twice as pair records
    (NRX_PLAN_PAIRS) -- the eight-entry form above, from the sort."""
    lib = _lib.load()
    n = len(ids)
    dev = ids[0].device
    total = sum(x.numel() for x in ids)
    order = torch.empty(total, dtype=torch.int64, device=dev)
    uniq = torch.empty(total, dtype=torch.int64, device=dev)
    seg = torch.empty(total + 1, dtype=torch.int64, device=dev)
    counts = torch.empty(n_tables + 2, dtype=torch.int64, device=dev)
    own_stream = stream is None
    if stream is None:
        stream = _raw_stream(dev)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in ids])
    lens = (C.c_int64 * n)(*[x.numel() for x in ids])
    if static is not None:                     # (table_of, rows) as ctypes arrays, built once per plan group
        tof, rws = static
    else:
        tof = (C.c_int32 * n)(*[int(t) for t in table_of])
        rws = (C.c_int64 * n)(*[int(r) for r in rows])
    if policy is not None and place_feats is not None and policy.choose():
        state = _lds_state(dev, stream)
        if state is not None:
            ws = torch.empty(lib.nrx_sparse_plan_lds_workspace(total), dtype=torch.uint8, device=dev)
            dest = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
            walk = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
            n_walk = torch.empty(2, dtype=torch.int64, device=dev)                      # [0] walk rows  [1] pair records
            recs = torch.empty((total // 2 + 1, 4), dtype=torch.int32, device=dev)       # (not `pairs`: that is the caller's flag, read again below
                                                                                           #  when the planner declines the launch)
            rc = lib.nrx_sparse_plan_lds(ptrs, lens, tof, rws, n, ids[0].element_size() * 8, n_tables, order.data_ptr(), uniq.data_ptr(),
                                         seg.data_ptr(), counts.data_ptr(), dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr(),
                                         recs.data_ptr(), n_walk.data_ptr() + 8, policy.stats_ptr, state.data_ptr(), ws.data_ptr(), stream)
            if rc == 0:
                if not own_stream and keep is not None:
                    keep.append(ws)
                return order, uniq, seg, counts, dest, walk, n_walk, recs
            if rc != NRX_ERR_UNSUPPORTED:
                check(rc, "nrx_sparse_plan_lds")
    nbytes = lib.nrx_sparse_plan_workspace(total)
    if nbytes < 0:
        raise ValueError("sparse_plan: too many lookups for one plan")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if not own_stream and keep is not None:
        keep.append(ws)          # a foreign stream: the caller holds the workspace until the plan has run (sparse_plan_ahead)
    if pad is not None and total > 0:
        dest = walk = n_walk = None
        if place_feats is not None:
            dest = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
            walk = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
            n_walk = torch.empty(1, dtype=torch.int64, device=dev)
        check(lib.nrx_sparse_plan_ex(ptrs, lens, tof, rws, n, ids[0].element_size() * 8, n_tables, int(place_feats or 0),
                                     NRX_PLAN_SPLIT_PADDING if pad.choose() else 0, order.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
                                     counts.data_ptr(), _ptr(dest), _ptr(walk), _ptr(n_walk), None, None, pad.stats_arg(), ws.data_ptr(), stream),
              "nrx_sparse_plan_ex")
        return (order, uniq, seg, counts) if dest is None else (order, uniq, seg, counts, dest, walk, n_walk)
    if pairs and PLAN_PAIRS and place_feats is not None and int(place_feats) == (1 << n) - 1 and total > 0:
        dest = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        walk = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        n_walk = torch.empty(2, dtype=torch.int64, device=dev)                      # [0] walk rows  [1] pair records
        recs = torch.empty((total // 2 + 1, 4), dtype=torch.int32, device=dev)
        check(lib.nrx_sparse_plan_ex(ptrs, lens, tof, rws, n, ids[0].element_size() * 8, n_tables, int(place_feats), NRX_PLAN_PAIRS, order.data_ptr(),
                                     uniq.data_ptr(), seg.data_ptr(), counts.data_ptr(), dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr(),
                                     recs.data_ptr(), n_walk.data_ptr() + 8, None, ws.data_ptr(), stream), "nrx_sparse_plan_ex")
        if policy is not None and policy.eligible and PLAN_LDS == "auto":      # what the next batch's choice of planner needs
            lib.nrx_sparse_plan_stats(counts.data_ptr(), n_walk.data_ptr(), total, policy.stats_ptr, stream)
        return order, uniq, seg, counts, dest, walk, n_walk, recs
    if place_feats is not None:
        dest = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        walk = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        n_walk = torch.empty(1, dtype=torch.int64, device=dev)
        check(lib.nrx_sparse_plan_place(ptrs, lens, tof, rws, n, ids[0].element_size() * 8, n_tables, int(place_feats), order.data_ptr(),
                                        uniq.data_ptr(), seg.data_ptr(), counts.data_ptr(), dest.data_ptr(), walk.data_ptr(),
                                        n_walk.data_ptr(), ws.data_ptr(), stream), "nrx_sparse_plan_place")
        if policy is not None and policy.eligible and PLAN_LDS == "auto":      # what the next batch's choice of planner needs
            lib.nrx_sparse_plan_stats(counts.data_ptr(), n_walk.data_ptr(), total, policy.stats_ptr, stream)
        return order, uniq, seg, counts, dest, walk, n_walk
    check(lib.nrx_sparse_plan(ptrs, lens, tof, rws, n, ids[0].element_size() * 8, n_tables, order.data_ptr(), uniq.data_ptr(),
                              seg.data_ptr(), counts.data_ptr(), ws.data_ptr(), stream), "nrx_sparse_plan")
    return order, uniq, seg, counts


def _bwd_sorted(lib, pl, pmask, arr, n, B, D, g_out, ld, g_wide, wide_ld, n_unique, n_unique_dev, fmg, values, lws, stream,
                dense=None, replan=None):
    """nrx_embed_bwd_sorted, or nrx_embed_bwd_placed when the plan `pl` carries a placement (7 tensors), or nrx_embed_bwd_placed_pairs for
    a plan of the one-kernel planner (8 entries).  dense = (grad table pointers, n_tables, accumulate): the dense destination instead of
    values.  Returns the plan that was used: a launch outside the pair pass's shapes (NRX_ERR_UNSUPPORTED, nothing enqueued) is planned
    again by the sorted planner (`replan()`), whose plan every reduction takes."""
    if len(pl) == 8:                    # nrx_sparse_plan_lds's plan: pair rows
        rc = lib.nrx_embed_bwd_placed_pairs(arr, n, B, D, _ptr(g_out), ld, _ptr(g_wide), wide_ld, pl[0].data_ptr(), pl[2].data_ptr(),
                                            pl[1].data_ptr(), n_unique, _ptr(n_unique_dev), fmg,
                                            values.data_ptr() if dense is None else None, dense[0] if dense else None,
                                            dense[1] if dense else 0, dense[2] if dense else 0, int(pmask),
                                            pl[4].data_ptr(), pl[5].data_ptr(), pl[6].data_ptr(), pl[7].data_ptr(), pl[6].data_ptr() + 8,
                                            lws.data_ptr(), lws.numel(), _aux_stream(pl[0].device), stream)
        if rc == 0:
            return pl
        if rc != NRX_ERR_UNSUPPORTED or replan is None:
            check(rc, "nrx_embed_bwd_placed_pairs")
        pl = replan()
        if n_unique_dev is not None:
            n_unique_dev = pl[3]
    order, uniq, seg = pl[0], pl[1], pl[2]
    placed = len(pl) >= 7
    if dense is not None:
        check(lib.nrx_embed_bwd_placed_dense(arr, n, B, D, _ptr(g_out), ld, _ptr(g_wide), wide_ld, order.data_ptr(), seg.data_ptr(),
                                             uniq.data_ptr(), n_unique, _ptr(n_unique_dev), fmg, dense[0], dense[1], dense[2],
                                             int(pmask) if placed else 0, pl[4].data_ptr() if placed else None,
                                             pl[5].data_ptr() if placed else None, pl[6].data_ptr() if placed else None,
                                             lws.data_ptr(), lws.numel(), stream), "nrx_embed_bwd_placed_dense")
        return pl
    check(lib.nrx_embed_bwd_placed(arr, n, B, D, _ptr(g_out), ld, _ptr(g_wide), wide_ld, order.data_ptr(), seg.data_ptr(),
                                   uniq.data_ptr(), n_unique, _ptr(n_unique_dev), fmg, values.data_ptr(), int(pmask) if placed else 0,
                                   pl[4].data_ptr() if placed else None, pl[5].data_ptr() if placed else None,
                                   pl[6].data_ptr() if placed else None, lws.data_ptr(), lws.numel(), stream),
          "nrx_embed_bwd_placed")
    return pl


def _sparse_group_cache(plan: EmbedPlan, tables):
    """Per launch group of the row-sparse backward, built once per (plan, table set): slot indices, the group's sub-plan (its cached
    descriptor array lives there), table ids / row counts as the ctypes arrays nrx_sparse_plan takes, the placement mask."""
    key = (len(tables), tables[0].data_ptr() if len(tables) else 0)
    ent = plan.__dict__.get("_sg")
    if ent is None or ent[0] != key:
        groups = []
        for D, fs in _sparse_groups(plan):
            tabs = [plan.slots[i].table for i in fs]
            n = len(fs)
            groups.append(dict(dim=D, fs=fs, tabs=tabs, n=n,
                               sub=EmbedPlan([plan.slots[i] for i in fs], out_width=plan.out_width, wide_width=plan.wide_width),
                               static=((C.c_int32 * n)(*tabs), (C.c_int64 * n)(*[int(tables[t].shape[0]) for t in tabs])),
                               rows=[int(tables[t].shape[0]) for t in tabs],
                               pmask=place_mask([plan.slots[i].kind for i in fs], [plan.slots[i].bag_len for i in fs]),
                               all_sparse=all(plan.slots[i].kind == NRX_SPARSE for i in fs) and D in (16, 32, 64), policy=None, policy_B=-1))
        ent = plan.__dict__["_sg"] = (key, groups)
    return ent[1]


def _group_policy(grp, B: int, n_tables: int) -> Optional["PlanPolicy"]:
    """The launch group's PlanPolicy at batch size B (built once per batch size; None: the group can never take the one-kernel planner)."""
    if not grp["all_sparse"] or PLAN_LDS == "0" or not SPARSE_PLACE:
        return None
    if grp["policy_B"] != B:
        by_b = grp.setdefault("policies", {})          # one per batch size, kept: an epoch's shorter last batch (or an evaluation at another size)
        pol = by_b.get(B)                              # must neither reset the full batches' statistics nor drop a policy a captured graph names
        if pol is None:
            n = grp["n"]
            lens = (C.c_int64 * n)(*([B] * n))
            pol = by_b[B] = PlanPolicy(lens, grp["static"][0], grp["static"][1], n, n_tables, B * n)
        grp["policy"] = pol
        grp["policy_B"] = B
    return grp["policy"]


def _group_pad(grp, plan: EmbedPlan, B: int) -> Optional["PadPolicy"]:
    """The launch group's PadPolicy at batch size B: groups with multi-valued features from PAD_SPLIT_MIN lookups on, else None."""
    if PAD_SPLIT == "0":
        return None
    cache = grp.setdefault("pad", {})          # per batch size (an epoch's last, shorter batch must not reset the full batches' statistics)
    if B not in cache:
        total = sum(B * max(1, plan.slots[i].bag_len) for i in grp["fs"])
        bags = any(plan.slots[i].kind != NRX_SPARSE for i in grp["fs"])
        cache[B] = PadPolicy(total) if bags and total >= PAD_SPLIT_MIN else None      # (never evicted: 64 bytes of the pinned arena each)
    return cache[B]


def _sparse_groups(plan: EmbedPlan):
    """(dim, slot indices) of every backward launch of the row-sparse mode: table features grouped by embedding dim, in
    chunks of NRX_MAX_FEATURES."""
    by_dim = {}
    for i, s in enumerate(plan.slots):
        if s.kind != NRX_DENSE:
            by_dim.setdefault(s.dim, []).append(i)
    out = []
    for D, fs_all in by_dim.items():
        for c0 in range(0, len(fs_all), NRX_MAX_FEATURES):
            out.append((D, fs_all[c0:c0 + NRX_MAX_FEATURES]))
    return out


def _sorted_sparse_grads(ctx, lib, g_out, g_wide, stream, fmg=None, dense_into=None):
    """Row-sparse, deterministic table grads.  Per group of tables sharing an embedding dim: ONE planning
    call (nrx_sparse_plan: compact (table, row) keys, table-segmented stable radix sort of the row bits, head
    flags + scan -> unique rows, segment starts, per-table split, all on the device), one host read of
    n_tables + 2 integers, ONE segmented-reduction launch (nrx_embed_bwd_sorted) summing the upstream rows
    of every unique (table, row) in sorted order.  No dense zero-fill, no atomics, bit-reproducible;
    padding rows get explicit zeros.  Returns torch.sparse_coo tensors (what nn.Embedding(sparse=True)
    produces), usable with SGD / SparseAdam / Adagrad.
    dense_into (a list of zero-filled [rows, dim] tensors, one per table): the DEFAULT dense-gradient mode -- the same
    planning and reduction, no host read at all, then nrx_rows_to_dense stores every unique row's sum at its place."""
    plan, B, ld = ctx.plan, ctx.B, ctx.ld
    tables = ctx.tables_ref
    n_tables = len(tables)
    grads = [None] * n_tables
    dense_ptrs, dense_seen = None, set()
    MASK = (1 << 40) - 1
    ahead = getattr(ctx, "plans", None) or {}
    for grp in _sparse_group_cache(plan, tables):
        D, fs, tabs, n = grp["dim"], grp["fs"], grp["tabs"], grp["n"]
        dev = tables[tabs[0]].device
        pre = ahead.get((D, fs[0]))
        pmask = grp["pmask"] if SPARSE_PLACE else None
        if pre is None and ctx.sink is not None and SPARSE_SMALL_DET and B <= 4096 and n_tables <= 256:
            # fused row-sparse optimizer at the reference's batch sizes: ONE launch leaves (key, summed row) pairs in the sink's format --
            # per-table regions with -1 fillers (nrx_embed_bwd_small_sparse); NRX_ERR_UNSUPPORTED: the planned reduction below
            ids = [ctx.ins[i] for i in fs]
            dt = ids[0].dtype
            if any(x.dtype != dt for x in ids):
                ids = [x.long() for x in ids]
            total = sum(x.numel() for x in ids)
            if total == 0:
                continue
            arr = _group_features(grp, ids, [ctx.ws[i] for i in fs], fmg is not None)
            np.frombuffer(arr, dtype=_feature_np_dtype())["rows"] = grp["rows"]
            uniq = torch.empty(total, dtype=torch.int64, device=dev)
            values = torch.empty((total, D), dtype=torch.float32, device=dev)
            rc = lib.nrx_embed_bwd_small_sparse(arr, grp["static"][0], n, B, _ptr(g_out), ld, _ptr(g_wide), plan.wide_width, fmg,
                                                uniq.data_ptr(), values.data_ptr(), total, stream)
            if rc == 0:
                ctx.sink.pending.append(dict(tables=ctx.tables, dim=D, uniq=uniq, values=values, counts=None, cap=total, filler=True, table_ids=sorted(set(tabs))))
                continue
            if rc != NRX_ERR_UNSUPPORTED:
                check(rc, "nrx_embed_bwd_small_sparse")
        if pre is None and ctx.sink is not None and SINK_ONE_CALL and n_tables <= NRX_MAX_FEATURES:
            # fused row-sparse optimizer, nothing planned ahead (a launch group the one-kernel planner takes is planned inline; captured steps): plan +
            # reduction in ONE library call, every intermediate in a workspace the group keeps from step to step (same stream: the next step's
            # kernels queue behind this one's) -- three allocations (keys, rows, counts: the sink holds them) and one call instead of thirteen and two
            pol = _group_policy(grp, B, n_tables)
            if pol is not None and pol.eligible and not any(plan.slots[i].flags & NRX_FEAT_BAG_CSR for i in fs):
                ids = [ctx.ins[i] for i in fs]
                dt = ids[0].dtype
                if any(x.dtype != dt for x in ids):
                    ids = [x.long() for x in ids]
                total = B * n
                if total == 0:
                    continue
                arr = _group_features(grp, ids, [ctx.ws[i] for i in fs], fmg is not None)
                np.frombuffer(arr, dtype=_feature_np_dtype())["rows"] = grp["rows"]
                # the workspace: kept per (batch size, stream) between eager steps -- at most four, the least recently used one goes first (its last use
                # was enqueued on its own stream: the caching allocator orders the reuse behind it).  While a HIP graph is being captured the
                # workspace is allocated fresh, from the graph's own pool: a captured launch must not point into a buffer this cache may drop
                if torch.cuda.is_current_stream_capturing():
                    wk = (None, torch.empty(lib.nrx_embed_bwd_sparse_planned_workspace(arr, n, B, D, n_tables), dtype=torch.uint8, device=dev))
                else:
                    cache = grp.setdefault("sink_ws", {})
                    wk = cache.pop((B, stream), None)
                    if wk is None:
                        wk = ((B, stream), torch.empty(lib.nrx_embed_bwd_sparse_planned_workspace(arr, n, B, D, n_tables), dtype=torch.uint8, device=dev))
                        while len(cache) >= 4:
                            cache.pop(next(iter(cache)))
                    cache[(B, stream)] = wk
                uniq = torch.empty(total, dtype=torch.int64, device=dev)
                values = torch.empty((total, D), dtype=torch.float32, device=dev)
                counts = torch.empty(n_tables + 2, dtype=torch.int64, device=dev)
                state = _lds_state(dev, stream) if pol.choose() else None
                check(lib.nrx_embed_bwd_sparse_planned(arr, grp["static"][0], n, n_tables, B, D, _ptr(g_out), ld, _ptr(g_wide), plan.wide_width, fmg,
                                                       uniq.data_ptr(), values.data_ptr(), counts.data_ptr(), 1 if state is not None else 0,
                                                       state.data_ptr() if state is not None else None,
                                                       pol.stats_ptr if PLAN_LDS == "auto" else None, wk[1].data_ptr(), wk[1].numel(), stream),
                      "nrx_embed_bwd_sparse_planned")
                ctx.sink.pending.append(dict(tables=ctx.tables, dim=D, uniq=uniq, values=values, counts=counts, cap=total, table_ids=sorted(set(tabs))))
                continue
        if pre is not None:                 # planned at forward time on the side stream (sparse_plan_ahead)
            ids, pl, ev = pre
            _cur_stream(dev).wait_event(ev)
            total = pl[0].numel()
        elif dense_into is not None and n_tables <= NRX_MAX_FEATURES:
            # default mode, nothing planned ahead (small launches, captured steps): plan + reduction into the dense gradients in ONE library
            # call with ONE workspace allocation (nrx_embed_bwd_dense_sorted) -- the ten allocations and two calls of the general path below
            # were most of a small step's host time
            ids = [ctx.ins[i] for i in fs]
            dt = ids[0].dtype
            if any(x.dtype != dt for x in ids):
                ids = [x.long() for x in ids]
            if sum(x.numel() for x in ids) == 0:
                continue
            if dense_ptrs is None:
                dense_ptrs = (C.c_void_p * n_tables)(*[g.data_ptr() for g in dense_into])
            again = any(t in dense_seen for t in tabs)
            arr = _group_features(grp, ids, [ctx.ws[i] for i in fs], fmg is not None, [dense_into[t].data_ptr() for t in tabs])
            np.frombuffer(arr, dtype=_feature_np_dtype())["rows"] = grp["rows"]
            pol = _group_policy(grp, B, n_tables) if B * n >= DENSE_LDS_MIN else None
            if pol is not None and pol.eligible:
                # a launch the one-kernel planner may take: the same one call with the planner as an argument (this batch's choice from the last
                # batch's statistics, which either planner leaves in the policy's mapped words)
                wsz = grp.get("planned_ws")
                if wsz is None or wsz[0] != B:
                    wsz = grp["planned_ws"] = (B, lib.nrx_embed_bwd_dense_planned_workspace(arr, n, B, D, n_tables))
                fws = torch.empty(wsz[1], dtype=torch.uint8, device=dev)
                state = _lds_state(dev, stream) if pol.choose() else None
                check(lib.nrx_embed_bwd_dense_planned(arr, grp["static"][0], n, n_tables, B, D, _ptr(g_out), ld, _ptr(g_wide), plan.wide_width, fmg,
                                                      dense_ptrs, 1 if again else 0, 1 if state is not None else 0,
                                                      state.data_ptr() if state is not None else None,
                                                      pol.stats_ptr if PLAN_LDS == "auto" else None, fws.data_ptr(), fws.numel(), stream),
                      "nrx_embed_bwd_dense_planned")
                dense_seen.update(tabs)
                continue
            wsz = grp.get("fused_ws")
            if wsz is None or wsz[0] != B:
                wsz = grp["fused_ws"] = (B, lib.nrx_embed_bwd_dense_sorted_workspace(arr, n, B, D, n_tables))
            fws = torch.empty(wsz[1], dtype=torch.uint8, device=dev)
            check(lib.nrx_embed_bwd_dense_sorted(arr, grp["static"][0], n, n_tables, B, D, _ptr(g_out), ld, _ptr(g_wide), plan.wide_width, fmg,
                                                 dense_ptrs, 1 if again else 0, 1 if SPARSE_PLACE else 0, fws.data_ptr(), fws.numel(), stream),
                  "nrx_embed_bwd_dense_sorted")
            dense_seen.update(tabs)
            continue
        else:
            ids = [ctx.ins[i] for i in fs]
            dt = ids[0].dtype
            if any(x.dtype != dt for x in ids):
                ids = [x.long() for x in ids]
            total = sum(x.numel() for x in ids)
            if total == 0:
                continue
            pl = sparse_plan(ids, tabs, grp["rows"], n_tables, pmask, static=grp["static"],
                             policy=_group_policy(grp, B, n_tables) if n_tables <= NRX_MAX_FEATURES else None, pad=_group_pad(grp, plan, B),
                             pairs=grp["all_sparse"])
        uniq, counts = pl[1], pl[3]

        def replan(ids=ids, tabs=tabs, grp=grp, pmask=pmask):
            """The one-kernel planner's plan met a launch the pair pass does not take: the sorted planner's plan (same unique rows, same
            counts), inline; the group stops choosing the one-kernel planner."""
            grp["all_sparse"] = False
            return sparse_plan(ids, tabs, grp["rows"], n_tables, pmask, static=grp["static"])

        # The one host read (n_tables + 2 integers).  Reading it BEFORE the reduction lets the host build the
        # per-table COO tensors while that kernel runs (1.00 ms per C2 step vs 1.11 ms with the sync-free
        # n_unique_dev form of the call, which leaves the host work exposed after the GPU is done).
        arr = _group_features(grp, ids, [ctx.ws[i] for i in fs], fmg is not None)
        wsz = grp.get("lws_bytes")
        if wsz is None or wsz[0] != B:
            wsz = grp["lws_bytes"] = (B, lib.nrx_embed_bwd_workspace_for(arr, n, B, D))
        lws = torch.empty(wsz[1], dtype=torch.uint8, device=dev)   # hot-row work lists, bag scales
        if dense_into is not None:
            if dense_ptrs is None:
                dense_ptrs = (C.c_void_p * n_tables)(*[g.data_ptr() for g in dense_into])
            again = any(t in dense_seen for t in tabs)      # > NRX_MAX_FEATURES features: a table may be fed by two groups
            # the reduction stores every unique row's sum at its place in the table's dense gradient (nrx_embed_bwd_placed_dense: the
            # descriptors' table column carries the gradient of the feature's table) -- no values[] array, no nrx_rows_to_dense pass
            arr = _group_features(grp, ids, [ctx.ws[i] for i in fs], fmg is not None, [dense_into[t].data_ptr() for t in tabs])
            _bwd_sorted(lib, pl, pmask, arr, n, B, D, g_out, ld, g_wide, plan.wide_width, total, counts, fmg, None, lws, stream,
                        dense=(dense_ptrs, n_tables, 1 if again else 0), replan=replan)
            dense_seen.update(tabs)
            continue
        if ctx.sink is not None:
            values = torch.empty((total, D), dtype=torch.float32, device=dev)     # worst case: every lookup unique
            pl = _bwd_sorted(lib, pl, pmask, arr, n, B, D, g_out, ld, g_wide, plan.wide_width, total, counts, fmg, values, lws, stream, replan=replan)
            ctx.sink.pending.append(dict(tables=ctx.tables, dim=D, uniq=pl[1], values=values, counts=pl[3], cap=total, table_ids=sorted(set(tabs))))
            continue
        if SPARSE_BWD_SYNC_FREE:
            values = torch.empty((total, D), dtype=torch.float32, device=dev)     # worst case: every lookup unique
            pl = _bwd_sorted(lib, pl, pmask, arr, n, B, D, g_out, ld, g_wide, plan.wide_width, total, counts, fmg, values, lws, stream, replan=replan)
            cl = pl[3].tolist()
            nu = cl[0]
        else:
            cl = counts.tolist()
            nu = cl[0]
            values = torch.empty((nu, D), dtype=torch.float32, device=dev)
            pl = _bwd_sorted(lib, pl, pmask, arr, n, B, D, g_out, ld, g_wide, plan.wide_width, nu, None, fmg, values, lws, stream, replan=replan)
                                                              # padding rows (id 0) come back as zeros
        uniq = pl[1]          # (after a replan: the plan the reduction took -- the same unique rows and counts, in that plan's buffers)
        rows = (uniq[:nu] & MASK).unsqueeze(0)
        for t in sorted(set(tabs)):
            lo, hi = cl[1 + t], cl[2 + t]
            g = torch.sparse_coo_tensor(rows[:, lo:hi], values[lo:hi], size=tables[t].shape, is_coalesced=True)
            grads[t] = g if grads[t] is None else (grads[t] + g).coalesce()
    if ctx.sink is not None or dense_into is not None:
        return grads                        # the results went to the sink: no (empty) COO tensors to build -- 26 of them were 130 us per step
    for t in range(n_tables):
        if grads[t] is None:
            shape, dev = _table_meta(ctx)[t]
            grads[t] = torch.sparse_coo_tensor(torch.zeros((1, 0), dtype=torch.int64, device=dev),
                                               torch.zeros((0, shape[1]), dtype=torch.float32, device=dev), size=shape)
    return grads


def _group_features(grp, ids, ws, fm: bool, table_ptrs=None):
    """The descriptor array of one backward group: static fields written once per group, the id / weight pointer columns refreshed
    through a numpy view.  table_ptrs (one per feature of the group): the table column -- unused by the row-sparse reduction, the
    gradient of the feature's table for the dense destination (nrx_embed_bwd_placed_dense)."""
    import numpy as np
    key = "arr_fm" if fm else "arr"
    ent = grp.get(key)
    if ent is None:
        sub, n = grp["sub"], grp["n"]
        arr = _fill_features(sub, 0, n, [None] * (max(grp["tabs"]) + 1), ids, ws, table_ptrs=[0] * (max(grp["tabs"]) + 1), fm=fm)
        ent = grp[key] = (arr, np.frombuffer(arr, dtype=_feature_np_dtype()))
        if table_ptrs is None:
            return arr
    arr, view = ent
    view["index"] = [x.data_ptr() for x in ids]
    view["index_bits"] = ids[0].element_size() * 8
    view["weight"] = [0 if w is None else w.data_ptr() for w in ws]
    view["table"] = table_ptrs if table_ptrs is not None else 0
    return arr


_FEATURE_DTYPE = None


def _feature_np_dtype():
    """numpy view of struct nrx_feature (64 bytes): lets a call refresh the pointer columns of a cached descriptor array
    with one vectorised assignment instead of 26 ctypes field stores."""
    global _FEATURE_DTYPE
    if _FEATURE_DTYPE is None:
        import numpy as np
        _FEATURE_DTYPE = np.dtype([("table", "<u8"), ("index", "<u8"), ("weight", "<u8"), ("rows", "<i8"), ("dim", "<i4"),
                                   ("bag_len", "<i4"), ("kind", "<i4"), ("index_bits", "<i4"), ("out_col", "<i4"), ("wide_col", "<i4"),
                                   ("fm_field", "<i4"), ("flags", "<i4")])
        assert _FEATURE_DTYPE.itemsize == C.sizeof(NrxFeature)
    return _FEATURE_DTYPE


class _FastForward:
    """Inference / no-grad form of embed_apply for plans of <= 64 features: the descriptor array lives on the plan with
    its static fields filled once; a call validates the inputs in ONE pass, refreshes the pointer columns through a numpy
    view and enqueues the launch.  ~3x less host time than the autograd path (profiles/r02_host_overhead.txt)."""

    def __init__(self, plan: EmbedPlan):
        import numpy as np
        self.lib = _lib.load()
        self.plan = plan
        n = len(plan.slots)
        self.n = n
        self.arr = (NrxFeature * n)()
        self.view = np.frombuffer(self.arr, dtype=_feature_np_dtype())
        for i, s_ in enumerate(plan.slots):
            f = self.arr[i]
            f.kind, f.dim, f.bag_len, f.out_col, f.wide_col, f.fm_field, f.flags = s_.kind, s_.dim, s_.bag_len, s_.out_col, s_.wide_col, s_.fm_field, s_.flags
        self.table_of = [s_.table for s_ in plan.slots]
        self.kinds = [s_.kind for s_ in plan.slots]
        self.simple = all(k == NRX_SPARSE for k in self.kinds)            # ids only: no masks, no dense values
        self.bag = [(i, s_.bag_len, s_.kind) for i, s_ in enumerate(plan.slots) if s_.kind >= NRX_BAG_MASKED_MEAN]
        self._tables_id = None
        self._tkey = None
        self.fm_dim = max((s_.dim for s_ in plan.slots if s_.fm_field), default=0)
        self.fwd = self.lib.nrx_embed_fwd_train

    def _bind_tables(self, tables):
        i64 = torch.float32
        for t in tables:
            if not (t.is_cuda and t.dtype is i64 and t.is_contiguous()):
                _f32c(t, "embedding table")
                raise ValueError("embedding tables must be contiguous [rows, dim] fp32")
        ptrs = [t.data_ptr() for t in tables]
        rows = [t.shape[0] for t in tables]
        self.view["table"] = [ptrs[k] if k >= 0 else 0 for k in self.table_of]
        self.view["rows"] = [rows[k] if k >= 0 else 0 for k in self.table_of]
        self._tables_id = tables
        self._tkey = (len(tables), ptrs[0], ptrs[-1]) if tables else None
        self.device = tables[0].device if tables else None

    def __call__(self, tables, inputs, weights, out_ld, need_out, mode):
        plan = self.plan
        bp = _bound_plan(plan)
        if bp is not None and type(tables) is list and type(inputs) is list and type(weights) is list and tables and tables[0].is_cuda:
            dev = tables[0].device
            status = None
            if mode == "deferred":
                sp = _deferred_status(plan.names).data_ptr()
            elif mode == "off":
                sp = 0
            else:
                status = torch.zeros(4, dtype=torch.int32, device=dev)
                sp = status.data_ptr()
            res = bp.forward(tables, inputs, weights, int(out_ld) if out_ld else 0, need_out, sp, _raw_stream(dev), False)
            if res is not None:
                if type(res) is int:
                    check(res, "nrx_embed_fwd")
                if status is not None:
                    if mode == "sync":
                        _raise_if_oob(status, plan.names)
                    else:
                        ev = torch.cuda.Event()
                        ev.record()
                        _pending_status.append((status, ev, plan.names))
                        while _pending_status and _pending_status[0][1].query():
                            st, _, nm = _pending_status.pop(0)
                            _raise_if_oob(st, nm)
                return res[1], res[2], res[3]
        if tables is not self._tables_id or (tables and self._tkey != (len(tables), tables[0].data_ptr(), tables[-1].data_ptr())):
            self._bind_tables(tables)
        if self.simple:
            x0 = inputs[0]
            dt, B = x0.dtype, x0.shape[0]
            ptrs = []
            for x in inputs:                                     # the one validating pass
                if x.dtype is not dt or not x.is_cuda or x.dim() != 1 or x.shape[0] != B or not x.is_contiguous():
                    return None                                  # odd input: let the general path convert / complain
                ptrs.append(x.data_ptr())
            if dt is torch.int64:
                bits = 64
            elif dt is torch.int32:
                bits = 32
            else:
                return None
            self.view["index"] = ptrs
            if bits != getattr(self, "_bits", None):
                self.view["index_bits"] = bits
                self._bits = bits
            dev = x0.device
        else:
            B, ins, ws = _prep_inputs(plan, tables, inputs, weights)
            self.view["index"] = [x.data_ptr() for x in ins]
            self.view["index_bits"] = [x.element_size() * 8 for x in ins]
            self.view["weight"] = [0 if w is None else w.data_ptr() for w in ws]      # (ignored by the library for padded NRX_BAG_MEAN)
            self._bits = None
            self._keep = (ins, ws)
            dev = ins[0].device
        ld = int(out_ld) if out_ld else plan.out_width
        if ld < plan.out_width:
            raise ValueError("out_ld smaller than the plan's out_width")
        out = torch.empty((B, ld), dtype=torch.float32, device=dev) if need_out else None
        wide = torch.empty((B, plan.wide_width), dtype=torch.float32, device=dev) if plan.wide_width else None
        fm = torch.empty((B,), dtype=torch.float32, device=dev) if plan.use_fm else None
        if mode == "deferred":
            status = _deferred_status(plan.names)
        else:
            status = torch.zeros(4, dtype=torch.int32, device=dev) if mode != "off" else None
        if B > 0:
            rc = self.fwd(self.arr, self.n, B, _ptr(out), ld, _ptr(wide), plan.wide_width, _ptr(fm), None, 0, _ptr(status),
                          torch.cuda.current_stream(dev).cuda_stream)
            if rc:
                check(rc, "nrx_embed_fwd")
        if status is not None and mode != "deferred":
            if mode == "sync":
                _raise_if_oob(status, plan.names)
            else:
                ev = torch.cuda.Event()
                ev.record()
                _pending_status.append((status, ev, plan.names))
                while _pending_status and _pending_status[0][1].query():
                    st, _, nm = _pending_status.pop(0)
                    _raise_if_oob(st, nm)
        return out, wide, fm


def embed_apply(plan: EmbedPlan, tables: Sequence[torch.Tensor], inputs: Sequence[torch.Tensor],
                weights: Sequence[Optional[torch.Tensor]], out_ld: Optional[int] = None, need_out: bool = True,
                sparse_grad=False, index_check: Optional[str] = None, narrow: bool = False):
    """Run the fused gather(+pool)->concat.  index_check: None = the module-wide mode (set_index_check; default 'sync' =
    IndexError in the offending call, like torch on CPU), or 'deferred' / 'sync' / 'off' for this call.  Returns (out[B, out_ld or out_width] | None,
    wide[B, wide_width] | None, fm[B] | None).  Differentiable w.r.t. `tables`: dense grads by default
    (what the reference's nn.Embedding(sparse=False) produces), or -- sparse_grad=True -- deterministic
    row-sparse COO grads (sorted segmented reduction; no full-table zero-fill), or -- sparse_grad=a
    SparseGradSink -- the same reduction left on the device for optim.FusedSparseAdam (tables get no .grad)."""
    if not need_out and not (plan.use_fm or plan.wide_width):
        raise ValueError("need_out=False only makes sense with an FM or wide output")
    if len(plan.slots) <= NRX_MAX_FEATURES and len(inputs) == len(plan.slots) and \
            (not torch.is_grad_enabled() or not any(t.requires_grad for t in tables)):
        fast = plan.__dict__.get("_fast")
        if fast is None:
            fast = plan.__dict__["_fast"] = _FastForward(plan)
        res = fast(tables, inputs, weights, out_ld, need_out, index_check or _INDEX_CHECK)
        if res is not None:
            if narrow and res[0] is not None and res[0].shape[1] != plan.out_width:
                return res[0].narrow(1, 0, plan.out_width), res[1], res[2]
            return res
    # narrow (with out_ld > out_width: a row stride padded for the kernels' sake, e.g. to a multiple of 4 floats so that the Wide&Deep split
    # can store aligned 16-byte chunks): `out` comes back as the [B, out_width] view of the [B, out_ld] buffer
    return _EmbedFn.apply(plan, list(inputs), list(weights), -int(out_ld) if (narrow and out_ld) else out_ld, need_out, sparse_grad,
                          index_check, *tables)


class PreparedEmbed:
    """A bound, re-launchable forward call (inference / benchmarking): validation, output allocation
    and the C descriptor arrays are done once; `run()` only enqueues the launch(es) on the current
    stream.  The ids / masks are read from the SAME tensors every run (refill them in place, or keep
    one PreparedEmbed per input buffer).  No autograd."""

    def __init__(self, plan: EmbedPlan, tables: Sequence[torch.Tensor], inputs, weights,
                 out_ld: Optional[int] = None, need_out: bool = True, check_index: bool = False,
                 out: Optional[torch.Tensor] = None, fm: Optional[torch.Tensor] = None, fm_sums: Optional[torch.Tensor] = None):
        """fm_sums (optional, [B, >= FM field dim] float32): training form -- the launch also leaves the FM field sums
        there (nrx_embed_fwd_train) for a later PreparedSparseBackward."""
        self.lib = _lib.load()
        self.plan = plan
        self.tables = [t.detach() for t in tables]
        self.B, self.ins, self.ws = _prep_inputs(plan, self.tables, list(inputs), list(weights))
        dev = self.ins[0].device
        self.ld = int(out_ld) if out_ld else plan.out_width
        B = self.B
        if out is not None and (tuple(out.shape) != (B, self.ld) or out.dtype != torch.float32 or not out.is_contiguous()):
            raise ValueError("out must be a contiguous float32 [B, out_ld] tensor")
        need_scratch = plan.use_fm and len(plan.slots) > NRX_MAX_FEATURES          # FM over > 64 fields runs on the concat
        self.out = out if out is not None else (torch.empty((B, self.ld), dtype=torch.float32, device=dev) if (need_out or need_scratch) else None)
        self.wide = torch.empty((B, plan.wide_width), dtype=torch.float32, device=dev) if plan.wide_width else None
        self.fm = fm if fm is not None else (torch.empty((B,), dtype=torch.float32, device=dev) if plan.use_fm else None)
        self.status = torch.zeros(4, dtype=torch.int32, device=dev) if check_index else None
        n = len(plan.slots)
        self.single = n <= NRX_MAX_FEATURES
        self.calls = []
        for lo in range(0, n, NRX_MAX_FEATURES):
            hi = min(n, lo + NRX_MAX_FEATURES)
            self.calls.append((_fill_features(plan, lo, hi, self.tables, self.ins, self.ws, fm=self.single), hi - lo))
        if fm_sums is not None and (not self.single or not plan.use_fm or fm_sums.dtype != torch.float32 or fm_sums.dim() != 2
                                    or fm_sums.shape[0] != B or not fm_sums.is_contiguous()):
            raise ValueError("fm_sums must be a contiguous float32 [B, dim] tensor, for an FM plan of <= 64 features")
        self.fm_sums = fm_sums
        self._args = (_ptr(self.out), self.ld, _ptr(self.wide), plan.wide_width,
                      _ptr(self.fm) if self.single else None, _ptr(fm_sums), fm_sums.shape[1] if fm_sums is not None else 0,
                      _ptr(self.status))
        self.device = dev

    def run(self):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fwd = self.lib.nrx_embed_fwd_train
        a = self._args
        for arr, n in self.calls:
            rc = fwd(arr, n, self.B, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], stream)
            if rc:
                check(rc, "nrx_embed_fwd_train")
        if self.plan.use_fm and not self.single:
            check(self.lib.nrx_fm_fwd(self.out.data_ptr(), self.ld, len(self.plan.slots), self.plan.slots[0].dim,
                                      self.B, self.fm.data_ptr(), stream), "nrx_fm_fwd")
        return self.out, self.wide, self.fm

    def check(self):
        if self.status is not None:
            _raise_if_oob(self.status, self.plan.names)


class PreparedSparseBackward:
    """A bound, re-launchable row-sparse backward of a PreparedEmbed launch (benchmarking / hand-written training
    loops): planning (nrx_sparse_plan) + the deterministic segmented reduction (nrx_embed_bwd_sorted, FM gradient
    folded in) with every buffer allocated once; `run()` only enqueues.  Results stay on the device: per embedding dim
    a dict(dim, uniq, values, counts, cap) exactly like an ops.SparseGradSink entry.  No autograd, no host reads."""

    def __init__(self, fwd: PreparedEmbed, g_out: Optional[torch.Tensor], g_fm: Optional[torch.Tensor] = None,
                 g_wide: Optional[torch.Tensor] = None, place_feats: Optional[int] = None, post_plan=None, payload: Optional[torch.Tensor] = None,
                 values: Optional[torch.Tensor] = None):
        """place_feats (optional): the placement mask instead of place_mask()'s -- 0 = a placement plan in which nothing is placeable: every row
        is listed and walked, two rows x four entries per pass (the form for launches whose rows are all looked up many times), no placement
        pass, and the plan always comes from the sorted planner.  post_plan (optional): callable(stream) enqueued between the plan and the
        reduction (the sharded step's pooled channel rewrites order[] there: nrx_pool_order_remap)."""
        self.lib, self.fwd, plan = fwd.lib, fwd, fwd.plan
        self.post_plan = post_plan
        # payload (int32 / uint32 [lookups], with place_feats = 0): nrx_sparse_plan_ex(NRX_PLAN_PAYLOAD) -- order[] lists payload[p] instead of p
        if payload is not None and (place_feats != 0 or payload.dtype not in (torch.int32, torch.uint32) or not payload.is_contiguous()):
            raise ValueError("PreparedSparseBackward: payload needs place_feats=0 and a contiguous 32-bit tensor")
        self.payload = payload
        # values (optional, plans of ONE embedding dim): the [lookups, dim] buffer the unique rows' gradients go to instead of one allocated here
        # (the sharded step keeps it at the head of an arena the requesters write placed rows into: plan_only() + run_walk())
        self._values_arg = values
        if not fwd.single:
            raise ValueError("PreparedSparseBackward covers plans of <= 64 features")
        self.g_out = None if g_out is None else _f32c(g_out, "g_out")
        self.g_wide = None if g_wide is None else _f32c(g_wide, "g_wide")
        self.fmg = None
        if g_fm is not None:
            if fwd.fm_sums is None or fwd.out is None:
                raise ValueError("an FM gradient needs the forward's fm_sums and concat (PreparedEmbed(..., fm_sums=...))")
            self.g_fm = _f32c(g_fm, "g_fm")
            self.fmg = NrxFmGrad(self.g_fm.data_ptr(), fwd.fm_sums.data_ptr(), fwd.fm_sums.shape[1], fwd.out.data_ptr(), fwd.ld)
        n_tables = len(fwd.tables)
        by_dim = {}
        for i, s in enumerate(plan.slots):
            if s.kind != NRX_DENSE:
                by_dim.setdefault(s.dim, []).append(i)
        self.groups = []
        dev = fwd.device
        for D, fs in by_dim.items():
            ids = [fwd.ins[i] for i in fs]
            if any(x.dtype != ids[0].dtype for x in ids):
                raise TypeError("PreparedSparseBackward: the ids of one embedding dim must share a dtype")
            n = len(fs)
            total = sum(x.numel() for x in ids)
            tabs = [plan.slots[i].table for i in fs]
            sub = EmbedPlan([plan.slots[i] for i in fs], out_width=plan.out_width, wide_width=plan.wide_width)
            arr = _fill_features(sub, 0, n, [None] * n_tables, ids, [fwd.ws[i] for i in fs], table_ptrs=[0] * n_tables,
                                 fm=self.fmg is not None)
            g = dict(dim=D, n=n, total=total, cap=total,
                     order=torch.empty(total, dtype=torch.int64, device=dev), uniq=torch.empty(total, dtype=torch.int64, device=dev),
                     seg=torch.empty(total + 1, dtype=torch.int64, device=dev),
                     counts=torch.empty(n_tables + 2, dtype=torch.int64, device=dev),
                     ws=torch.empty(max(1, self.lib.nrx_sparse_plan_workspace(total)), dtype=torch.uint8, device=dev),
                     values=values if values is not None else torch.empty((total, D), dtype=torch.float32, device=dev),
                     pmask=(place_feats if place_feats is not None else
                            place_mask([plan.slots[i].kind for i in fs], [plan.slots[i].bag_len for i in fs]) if SPARSE_PLACE else None),
                     dest=torch.empty(max(total, 1), dtype=torch.int32, device=dev),
                     walk=torch.empty(max(total, 1), dtype=torch.int32, device=dev),
                     n_walk=torch.empty(2, dtype=torch.int64, device=dev),                 # [0] walk rows  [1] pair records (one-kernel planner)
                     pair_recs=None,
                     lws=torch.empty(self.lib.nrx_embed_bwd_workspace_for(arr, n, fwd.B, D), dtype=torch.uint8, device=dev),
                     pairs=False, lds_ws=None,
                     ptrs=(C.c_void_p * n)(*[x.data_ptr() for x in ids]), lens=(C.c_int64 * n)(*[x.numel() for x in ids]),
                     tof=(C.c_int32 * n)(*tabs), rws=(C.c_int64 * n)(*[fwd.tables[t].shape[0] for t in tabs]),
                     bits=ids[0].element_size() * 8, n_tables=n_tables,
                     arr=arr)
            g["policy"] = None
            if g["pmask"] is not None and place_feats is None and D in (16, 32, 64) and all(plan.slots[i].kind == NRX_SPARSE for i in fs):
                g["policy"] = PlanPolicy(g["lens"], g["tof"], g["rws"], n, n_tables, total)
                g["pair_recs"] = torch.empty((total // 2 + 1, 4), dtype=torch.int32, device=dev)      # (either planner leaves pair records)
                if g["policy"].eligible:
                    g["lds_ws"] = torch.empty(self.lib.nrx_sparse_plan_lds_workspace(total), dtype=torch.uint8, device=dev)
            # multi-valued features: the padding lookups are set aside when the previous plan counted enough of them (PadPolicy)
            g["pad"] = PadPolicy(total) if PAD_SPLIT != "0" and total >= PAD_SPLIT_MIN and any(plan.slots[i].kind != NRX_SPARSE for i in fs) else None
            self.groups.append(g)
        if values is not None and (len(self.groups) != 1 or tuple(values.shape) != (self.groups[0]["total"], self.groups[0]["dim"])
                                   or values.dtype != torch.float32 or not values.is_contiguous()):
            raise ValueError("PreparedSparseBackward: `values` must be a contiguous float32 [lookups, dim] tensor for a plan of one embedding dim")

    def plan_only(self):
        """Enqueue the planning alone on the current stream (the following run_walk() / run() uses it)."""
        dev = self.fwd.device
        self._plan(torch.cuda.current_stream(dev).cuda_stream)
        if self.post_plan is not None:
            self.post_plan(torch.cuda.current_stream(dev).cuda_stream)
        self._planned = True

    def run_walk(self):
        """The reduction WITHOUT its placement pass (nrx_embed_bwd_walk), after plan_only(): the rows the plan places are in values[] already --
        the sharded step's requesters wrote them there (nrx_embed_bwd_scatter_multi with the plan's dest[]) -- and only the listed rows and the
        pair records are reduced from g_out.  Same (keys, values, counts) as run(), bit for bit."""
        lib, f = self.lib, self.fwd
        stream = torch.cuda.current_stream(f.device).cuda_stream
        if not getattr(self, "_planned", False):
            raise RuntimeError("run_walk() needs plan_only() first")
        self._planned = False
        for g in self.groups:
            if g["pmask"] is None:
                raise RuntimeError("run_walk(): the plan has no placement")
            pr = g["pairs"]
            rc = lib.nrx_embed_bwd_walk(g["arr"], g["n"], f.B, g["dim"], _ptr(self.g_out), f.ld, g["order"].data_ptr(), g["seg"].data_ptr(),
                                        g["uniq"].data_ptr(), g["total"], g["counts"].data_ptr(), self.fmg, g["values"].data_ptr(), g["pmask"],
                                        g["dest"].data_ptr(), g["walk"].data_ptr(), g["n_walk"].data_ptr(),
                                        g["pair_recs"].data_ptr() if pr else None, g["n_walk"].data_ptr() + 8 if pr else None,
                                        g["lws"].data_ptr(), g["lws"].numel(), stream)
            if rc:
                check(rc, "nrx_embed_bwd_walk")
        return self.groups

    def _plan(self, stream):
        lib = self.lib
        for g in self.groups:
            g["pairs"] = False
            pol = g["policy"]
            if pol is not None and pol.choose():
                state = _lds_state(self.fwd.device, stream)
                if state is not None:
                    rc = lib.nrx_sparse_plan_lds(g["ptrs"], g["lens"], g["tof"], g["rws"], g["n"], g["bits"], g["n_tables"], g["order"].data_ptr(),
                                                 g["uniq"].data_ptr(), g["seg"].data_ptr(), g["counts"].data_ptr(), g["dest"].data_ptr(),
                                                 g["walk"].data_ptr(), g["n_walk"].data_ptr(), g["pair_recs"].data_ptr(), g["n_walk"].data_ptr() + 8,
                                                 pol.stats_ptr, state.data_ptr(), g["lds_ws"].data_ptr(), stream)
                    if rc == 0:
                        g["pairs"] = True
                        continue
                    if rc != NRX_ERR_UNSUPPORTED:
                        check(rc, "nrx_sparse_plan_lds")
            if self.payload is not None:
                rc = lib.nrx_sparse_plan_ex(g["ptrs"], g["lens"], g["tof"], g["rws"], g["n"], g["bits"], g["n_tables"], 0, _lib.NRX_PLAN_PAYLOAD,
                                            g["order"].data_ptr(), g["uniq"].data_ptr(), g["seg"].data_ptr(), g["counts"].data_ptr(),
                                            g["dest"].data_ptr(), g["walk"].data_ptr(), g["n_walk"].data_ptr(), self.payload.data_ptr(), None, None,
                                            g["ws"].data_ptr(), stream)
            elif PLAN_PAIRS and g.get("pair_recs") is not None and g.get("sorted_pairs", True) and g["pmask"] == (1 << g["n"]) - 1:
                rc = lib.nrx_sparse_plan_ex(g["ptrs"], g["lens"], g["tof"], g["rws"], g["n"], g["bits"], g["n_tables"], g["pmask"], NRX_PLAN_PAIRS,
                                            g["order"].data_ptr(), g["uniq"].data_ptr(), g["seg"].data_ptr(), g["counts"].data_ptr(),
                                            g["dest"].data_ptr(), g["walk"].data_ptr(), g["n_walk"].data_ptr(), g["pair_recs"].data_ptr(),
                                            g["n_walk"].data_ptr() + 8, None, g["ws"].data_ptr(), stream)
                g["pairs"] = rc == 0
            elif g["pad"] is not None:
                pm = g["pmask"] is not None
                rc = lib.nrx_sparse_plan_ex(g["ptrs"], g["lens"], g["tof"], g["rws"], g["n"], g["bits"], g["n_tables"], g["pmask"] or 0,
                                            NRX_PLAN_SPLIT_PADDING if g["pad"].choose() else 0, g["order"].data_ptr(), g["uniq"].data_ptr(),
                                            g["seg"].data_ptr(), g["counts"].data_ptr(), g["dest"].data_ptr() if pm else None,
                                            g["walk"].data_ptr() if pm else None, g["n_walk"].data_ptr() if pm else None, None, None,
                                            g["pad"].stats_arg(), g["ws"].data_ptr(), stream)
            elif g["pmask"] is not None:
                rc = lib.nrx_sparse_plan_place(g["ptrs"], g["lens"], g["tof"], g["rws"], g["n"], g["bits"], g["n_tables"], g["pmask"],
                                               g["order"].data_ptr(), g["uniq"].data_ptr(), g["seg"].data_ptr(), g["counts"].data_ptr(),
                                               g["dest"].data_ptr(), g["walk"].data_ptr(), g["n_walk"].data_ptr(), g["ws"].data_ptr(), stream)
            else:
                rc = lib.nrx_sparse_plan(g["ptrs"], g["lens"], g["tof"], g["rws"], g["n"], g["bits"], g["n_tables"], g["order"].data_ptr(),
                                         g["uniq"].data_ptr(), g["seg"].data_ptr(), g["counts"].data_ptr(), g["ws"].data_ptr(), stream)
            if rc:
                check(rc, "nrx_sparse_plan")
            if pol is not None and pol.eligible and PLAN_LDS == "auto" and g["pmask"] is not None:
                lib.nrx_sparse_plan_stats(g["counts"].data_ptr(), g["n_walk"].data_ptr(), g["total"], pol.stats_ptr, stream)

    def plan_ahead(self):
        """Enqueue the planning (sort, unique rows, segments: it depends only on the ids) on a side stream NOW -- call it
        next to the forward launch; the following run() then waits for it instead of planning inline.  In a training step
        the dense model's forward and backward sit between the two calls and the planning is hidden behind them."""
        dev = self.fwd.device
        cur, side = torch.cuda.current_stream(dev), _plan_stream(dev)
        side.wait_stream(cur)              # ids ready; and the previous run()'s reduction is done with the plan buffers
        with torch.cuda.stream(side):
            self._plan(side.cuda_stream)
            self._plan_ev = torch.cuda.Event()
            self._plan_ev.record(side)

    def run(self):
        lib, f = self.lib, self.fwd
        cur = torch.cuda.current_stream(f.device)
        stream = cur.cuda_stream
        ev = getattr(self, "_plan_ev", None)
        if ev is not None:
            cur.wait_event(ev)
            self._plan_ev = None
        else:
            self._plan(stream)
        if self.post_plan is not None:
            self.post_plan(stream)
        for g in self.groups:
            pm = g["pmask"] is not None
            if g["pairs"]:
                rc = lib.nrx_embed_bwd_placed_pairs(g["arr"], g["n"], f.B, g["dim"], _ptr(self.g_out), f.ld, _ptr(self.g_wide), f.plan.wide_width,
                                                    g["order"].data_ptr(), g["seg"].data_ptr(), g["uniq"].data_ptr(), g["total"],
                                                    g["counts"].data_ptr(), self.fmg, g["values"].data_ptr(), None, 0, 0, g["pmask"],
                                                    g["dest"].data_ptr(), g["walk"].data_ptr(), g["n_walk"].data_ptr(), g["pair_recs"].data_ptr(),
                                                    g["n_walk"].data_ptr() + 8, g["lws"].data_ptr(), g["lws"].numel(), _aux_stream(f.device), stream)
                if rc == NRX_ERR_UNSUPPORTED:           # outside the pair pass's shapes: plan again with the sorted planner and no pair records, for good
                    g["policy"] = None
                    g["sorted_pairs"] = False
                    self._plan(stream)
                else:
                    if rc:
                        check(rc, "nrx_embed_bwd_placed_pairs")
                    continue
            rc = lib.nrx_embed_bwd_placed(g["arr"], g["n"], f.B, g["dim"], _ptr(self.g_out), f.ld, _ptr(self.g_wide), f.plan.wide_width,
                                          g["order"].data_ptr(), g["seg"].data_ptr(), g["uniq"].data_ptr(), g["total"],
                                          g["counts"].data_ptr(), self.fmg, g["values"].data_ptr(), g["pmask"] if pm else 0,
                                          g["dest"].data_ptr() if pm else None, g["walk"].data_ptr() if pm else None,
                                          g["n_walk"].data_ptr() if pm else None, g["lws"].data_ptr(), g["lws"].numel(), stream)
            if rc:
                check(rc, "nrx_embed_bwd_placed")
        return self.groups


# ------------------------------------------------------------------------------- pooling
class _BagPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, mask):
        lib = _lib.load()
        emb = _f32c(emb, "embedding")
        if emb.dim() != 3:
            raise ValueError("embedding must be [B, L, D]")
        B, L, D = emb.shape
        if mask is not None:
            mask = _f32c(mask.float() if mask.dtype != torch.float32 else mask, "mask")
            if tuple(mask.shape) != (B, L):
                raise ValueError("mask must be [B, L]")
        out = torch.empty((B, D), dtype=torch.float32, device=emb.device)
        check(lib.nrx_bag_pool_fwd(emb.data_ptr(), _ptr(mask), B, L, D, out.data_ptr(), _stream_ptr(emb)), "nrx_bag_pool_fwd")
        ctx.mask, ctx.shape = mask, (B, L, D)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, L, D = ctx.shape
        g = _f32c(g, "grad")
        gemb = torch.empty((B, L, D), dtype=torch.float32, device=g.device)
        check(lib.nrx_bag_pool_bwd(g.data_ptr(), _ptr(ctx.mask), B, L, D, gemb.data_ptr(), _stream_ptr(g)), "nrx_bag_pool_bwd")
        return gemb, None


def bag_pool(embedding: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """array_feature_pooling (base_model.py:273-282) on a materialised [B, L, D] tensor."""
    return _BagPoolFn.apply(embedding, mask)


# ------------------------------------------------------------------------------- FM
class _FmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, n_fields, dim):
        lib = _lib.load()
        feat = _f32c(feat, "features")
        B, W = feat.shape
        if W != n_fields * dim:
            raise RuntimeError(f"FM expects {n_fields} fields of equal dim {dim}; got width {W} "
                               "(the reference's torch.stack would fail on unequal field dims)")
        out = torch.empty((B,), dtype=torch.float32, device=feat.device)
        check(lib.nrx_fm_fwd(feat.data_ptr(), W, n_fields, dim, B, out.data_ptr(), _stream_ptr(feat)), "nrx_fm_fwd")
        ctx.save_for_backward(feat)
        ctx.nf, ctx.dim = n_fields, dim
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (feat,) = ctx.saved_tensors
        g = _f32c(g, "grad")
        B, W = feat.shape
        gfeat = torch.empty_like(feat)
        check(lib.nrx_fm_bwd(feat.data_ptr(), W, ctx.nf, ctx.dim, B, g.data_ptr(), None, 0, gfeat.data_ptr(), W,
                             _stream_ptr(feat)), "nrx_fm_bwd")
        return gfeat, None, None


_fm_head_states = {}


def _fm_head_state(dev: torch.device, stream: int) -> Optional[torch.Tensor]:
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(stream))
    st = _fm_head_states.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        st = torch.zeros(_lib.load().nrx_fm_head_state_bytes(), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        _fm_head_states[key] = st
    return st


class _FmHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, bias):
        lib = _lib.load()
        logit = _f32c(_dev(logit, "fm logit"), "fm logit").reshape(-1)
        bias = _f32c(_dev(bias, "fm bias"), "fm bias")
        out = torch.empty((logit.shape[0], 1), dtype=torch.float32, device=logit.device)
        check(lib.nrx_fm_head_fwd(logit.data_ptr(), bias.data_ptr(), out.data_ptr(), logit.shape[0], _raw_stream(logit.device)), "nrx_fm_head_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        B = out.shape[0]
        dev = out.device
        stream = _raw_stream(dev)
        if g.stride(0) == 0 and (g.dim() < 2 or g.shape[1] == 1):
            gs, stride = g, 0                        # an expanded scalar (the gradient of .sum() / .mean()): read once, not materialised
        else:
            gs, stride = _f32c(g, "grad of the fm head"), 1
        if gs.dtype != torch.float32:
            gs, stride = _f32c(g.contiguous(), "grad of the fm head"), 1
        g_logit = torch.empty((B,), dtype=torch.float32, device=dev)
        want_b = ctx.needs_input_grad[1]
        state = _fm_head_state(dev, stream) if want_b else None
        if want_b and state is None:                 # (captured before any eager step made the state words: the torch form)
            gl = (gs.reshape(-1)[:1].expand(B) if stride == 0 else gs.reshape(-1)) * (out.view(-1) * (1.0 - out.view(-1)))
            return gl, gl.sum().reshape(1)
        g_bias = torch.empty((1,), dtype=torch.float32, device=dev) if want_b else None
        check(lib.nrx_fm_head_bwd(gs.data_ptr(), stride, out.data_ptr(), g_logit.data_ptr(), _ptr(g_bias), _ptr(state), B, stream), "nrx_fm_head_bwd")
        return g_logit, g_bias


def fm_head(fm_logit: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """sigmoid(bias + fm_logit) as [B, 1] -- the last line of FMModel.forward (src/model/sort/fm/model.py:25-26) -- one launch forward, one backward
    (the bias gradient summed inside it, in a fixed order)."""
    return _FmHeadFn.apply(fm_logit, bias)


def fm_interaction(features: torch.Tensor, n_fields: int, dim: int) -> torch.Tensor:
    """[B, n_fields*dim] -> [B]: sum_f w_f + 0.5*sum_k[(sum_f v)^2 - sum_f v^2] (no bias, no sigmoid)."""
    return _FmFn.apply(features, n_fields, dim)


# ------------------------------------------------------------------------------- DCN v1
def _dcn_params(w: torch.Tensor, b: torch.Tensor):
    w = _f32c(w, "cross w")
    b = _f32c(b, "cross b")
    if w.dim() != 2 or w.shape != b.shape:
        raise ValueError("cross w/b must both be [n_layers, dim]")
    return w, b



def _dcn_v1_grads(w: torch.Tensor, b: torch.Tensor):
    """g_w / g_b buffers of a DCN-v1 backward: the ordered mode overwrites them, the atomic one adds into zeros (see _dcn_v1_bwd)."""
    if (WGRAD_ORDERED or not WGRAD_ATOMIC) and w.shape[0] > 0:
        return torch.empty_like(w), torch.empty_like(b)
    return torch.zeros_like(w), torch.zeros_like(b)


def _dcn_v1_bwd(lib, args, dim: int, n_layers: int, dev, stream, gw: torch.Tensor, gb: torch.Tensor) -> None:
    """nrx_dcn_v1_bwd_ordered(*args, workspace, stream) -- the cross weights' and biases' gradients (sums over the batch) added block by block in
    block order: as fast as or faster than the atomic launch with its two fills (B = 65 536, forward + backward: D = 112 x 3 layers 46.8 vs 51.2 us,
    320 x 2 81.5 vs 82.6, 640 x 3 473.1 vs 469.4; profiles/r05_ordered_wgrad.txt) and bit-reproducible -- unless NRX_WGRAD=atomic, or the stack's
    n_layers x dim is beyond the mode's LDS slabs
    (NRX_ERR_UNSUPPORTED, nothing enqueued): then nrx_dcn_v1_bwd(*args, stream) into zero-filled g_w / g_b."""
    if (WGRAD_ORDERED or not WGRAD_ATOMIC) and n_layers > 0:
        ws = torch.empty(lib.nrx_dcn_v1_bwd_ordered_workspace(dim, n_layers), dtype=torch.uint8, device=dev)
        rc = lib.nrx_dcn_v1_bwd_ordered(*args, ws.data_ptr(), stream)
        if rc != NRX_ERR_UNSUPPORTED:
            check(rc, "nrx_dcn_v1_bwd_ordered")
            return
        gw.zero_()
        gb.zero_()
        if WGRAD_ORDERED:
            # asked for explicitly: not silently -- the eager step warns (once per shape), a deterministic capture refuses (GraphedStep reads the counter)
            dense_bwd_paths["atomic"] += 1
            if ("dcn_v1", n_layers, dim) not in _atomic_warned:
                _atomic_warned.add(("dcn_v1", n_layers, dim))
                import warnings
                warnings.warn(f"ordered weight gradients asked for, but a DCN stack of {n_layers} layers x {dim} columns is beyond the ordered mode of "
                              "nrx_dcn_v1_bwd: its cross weights' gradients are summed with float atomics (not bit-reproducible)", RuntimeWarning, stacklevel=3)
    check(lib.nrx_dcn_v1_bwd(*args, stream), "nrx_dcn_v1_bwd")


class _DcnV1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, x0):
        lib = _lib.load()
        x = _f32c(x, "x")
        w, b = _dcn_params(w, b)
        B, D = x.shape
        if w.shape[1] != D:
            raise ValueError(f"cross weights are for dim {w.shape[1]}, input has dim {D}")
        if x0 is not None:
            x0 = _f32c(x0, "x0")
            if x0.shape != x.shape:
                raise ValueError(f"x0 {tuple(x0.shape)} and x_l {tuple(x.shape)} must have the same shape")
        out = torch.empty_like(x)
        check(lib.nrx_dcn_v1_fwd(x.data_ptr(), D, _ptr(x0), D, B, D, w.shape[0], w.data_ptr(), b.data_ptr(), out.data_ptr(), D,
                                 _stream_ptr(x)), "nrx_dcn_v1_fwd")
        ctx.sep = x0 is not None
        ctx.save_for_backward(x, w, b, *((x0,) if ctx.sep else ()))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w, b, *rest = ctx.saved_tensors
        x0 = rest[0] if ctx.sep else None
        g = _f32c(g, "grad")
        B, D = x.shape
        gx = torch.empty_like(x)
        gx0 = torch.empty_like(x) if ctx.sep else None
        gw, gb = _dcn_v1_grads(w, b)
        _dcn_v1_bwd(lib, (x.data_ptr(), D, _ptr(x0), D, B, D, w.shape[0], w.data_ptr(), b.data_ptr(), g.data_ptr(), D,
                          gx.data_ptr(), D, _ptr(gx0), D, gw.data_ptr(), gb.data_ptr()), D, w.shape[0], x.device, _stream_ptr(x), gw, gb)
        return gx, gw, gb, gx0


class _DcnV1LayersFn(torch.autograd.Function):
    """The v1 stack fed with the layers' OWN parameters (w_l, b_l of shape [dim, 1], what DCNLayer holds, dcn_arch.py:5-12): packed into
    the [n_layers, dim] arrays the kernel reads inside forward -- outside autograd, so no stack / unbind nodes and no stacked gradient to
    split: the per-layer gradients are views of the kernel's [n_layers, dim] outputs."""

    @staticmethod
    def forward(ctx, x, n, *params):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, D = x.shape
        w = torch.cat([p.reshape(1, -1) for p in params[:n]]).float()
        b = torch.cat([p.reshape(1, -1) for p in params[n:]]).float()
        if tuple(w.shape) != (n, D) or tuple(b.shape) != (n, D):
            raise ValueError(f"cross weights are for dim {w.shape[1]}, input has dim {D}")
        out = torch.empty_like(x)
        check(lib.nrx_dcn_v1_fwd(x.data_ptr(), D, None, D, B, D, n, w.data_ptr(), b.data_ptr(), out.data_ptr(), D, _stream_ptr(x)),
              "nrx_dcn_v1_fwd")
        ctx.save_for_backward(x, w, b)
        ctx.n, ctx.shapes = n, [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w, b = ctx.saved_tensors
        g = _f32c(g, "grad")
        B, D = x.shape
        n = ctx.n
        gx = torch.empty_like(x)
        gw, gb = _dcn_v1_grads(w, b)
        _dcn_v1_bwd(lib, (x.data_ptr(), D, None, D, B, D, n, w.data_ptr(), b.data_ptr(), g.data_ptr(), D, gx.data_ptr(), D, None, D,
                          gw.data_ptr(), gb.data_ptr()), D, n, x.device, _stream_ptr(x), gw, gb)
        grads = [gw[l].reshape(ctx.shapes[l]) for l in range(n)] + [gb[l].reshape(ctx.shapes[n + l]) for l in range(n)]
        return (gx, None, *grads)


class _PackRowsFn(torch.autograd.Function):
    """n parameter tensors of `dim` elements each ([dim, 1] as DCNLayer holds them, dcn_arch.py:9-10) -> ONE [n, dim] tensor, as a single autograd
    node: one cat kernel forward, and the per-layer gradients are views of the incoming [n, dim] gradient (torch.stack over `p[:, 0]` selects is
    2 n + 1 nodes per call and an unbind + n copies in the backward)."""

    @staticmethod
    def forward(ctx, *params):
        ctx.shapes = [p.shape for p in params]
        return torch.cat([p.reshape(1, -1) for p in params]).float()

    @staticmethod
    def backward(ctx, g):
        return tuple(g[i].reshape(s) for i, s in enumerate(ctx.shapes))


def pack_rows(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """[n, dim] from n per-layer parameter tensors (see _PackRowsFn)."""
    return _PackRowsFn.apply(*params)


def dcn_v1_layers(x: torch.Tensor, ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor]) -> torch.Tensor:
    """DCNNet.forward (dcn_arch.py:63-70) from the layers' own parameter tensors (each [dim, 1] or [dim])."""
    if len(ws) != len(bs) or not ws:
        raise ValueError("dcn_v1_layers: one w and one b per layer")
    return _DcnV1LayersFn.apply(x, len(ws), *ws, *bs)


def dcn_v1(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, x0: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DCNNet.forward (dcn_arch.py:63-70): all cross layers fused; w, b are [n_layers, dim].  With `x0` the stack
    starts from a later layer's input x (= x_l) and x0 is the cross network's layer-0 input -- the per-layer call
    DCNLayer.forward(x_l, x_0) (dcn_arch.py:14-30) is n_layers = 1 of this."""
    if x0 is x:
        x0 = None
    return _DcnV1Fn.apply(x, w, b, x0)


class _DcnV1CatFn(torch.autograd.Function):
    """In place: buf[:, :D] holds x (written by embed_apply with out_ld = 2D); fills buf[:, D:] with
    cross(x), giving torch.cat([x, cross], dim=1) of dcn/model.py:29 without the extra pass."""

    @staticmethod
    def forward(ctx, buf, w, b):
        lib = _lib.load()
        _dev(buf, "buf")
        w, b = _dcn_params(w, b)
        B, W2 = buf.shape
        D = W2 // 2
        if not buf.is_contiguous() or W2 != 2 * D or w.shape[1] != D:
            raise ValueError("buf must be contiguous [B, 2*dim]")
        check(lib.nrx_dcn_v1_fwd(buf.data_ptr(), W2, None, 0, B, D, w.shape[0], w.data_ptr(), b.data_ptr(),
                                 buf.data_ptr() + 4 * D, W2, _stream_ptr(buf)), "nrx_dcn_v1_fwd")
        ctx.mark_dirty(buf)
        ctx.save_for_backward(buf, w, b)
        return buf

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        buf, w, b = ctx.saved_tensors
        g = _f32c(g, "grad")
        B, W2 = buf.shape
        D = W2 // 2
        gbuf = torch.zeros_like(buf)          # right half of the input buffer was never read
        gw, gb = _dcn_v1_grads(w, b)
        _dcn_v1_bwd(lib, (buf.data_ptr(), W2, None, 0, B, D, w.shape[0], w.data_ptr(), b.data_ptr(),
                          g.data_ptr() + 4 * D, W2, gbuf.data_ptr(), W2, None, 0, gw.data_ptr(), gb.data_ptr()), D, w.shape[0], buf.device,
                    _stream_ptr(buf), gw, gb)
        gbuf[:, :D] += g[:, :D]
        return gbuf, gw, gb


def dcn_v1_cat_(buf: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _DcnV1CatFn.apply(buf, w, b)


class FusedUnsupported(Exception):
    """The fused gather+cross kernel does not cover this feature mix; use the two-launch path."""


class PreparedEmbedDcn:
    """Bound forward of gather -> concat -> DCN-v1 cross in ONE launch (inference / benchmarking):
    out[:, :W] = x, out[:, W:] = cross(x).  Raises FusedUnsupported when the plan is not eligible
    (bags, dense features, dims not multiples of 4, ...)."""

    def __init__(self, plan: EmbedPlan, tables, inputs, w: torch.Tensor, b: torch.Tensor,
                 out: Optional[torch.Tensor] = None, check_index: bool = False, status: Optional[torch.Tensor] = None):
        self.lib = _lib.load()
        if any(s.kind != NRX_SPARSE or s.wide_col >= 0 for s in plan.slots) or plan.use_fm or len(plan.slots) > NRX_MAX_FEATURES:
            raise FusedUnsupported("only plain single-valued features are fused")
        self.tables = [t.detach() for t in tables]
        self.B, self.ins, _ = _prep_inputs(plan, self.tables, list(inputs), [None] * len(plan.slots))
        self.w, self.b = _dcn_params(w.detach(), b.detach())
        W = plan.out_width
        if self.w.shape[1] != W:
            raise ValueError(f"cross weights are for dim {self.w.shape[1]}, the concat has {W}")
        dev = self.ins[0].device
        self.out = out if out is not None else torch.empty((self.B, 2 * W), dtype=torch.float32, device=dev)
        self.status = status if status is not None else (torch.zeros(4, dtype=torch.int32, device=dev) if check_index else None)
        self.arr = _fill_features(plan, 0, len(plan.slots), self.tables, self.ins, [None] * len(plan.slots), fm=False)
        self.plan, self.W, self.device = plan, W, dev
        self.run()          # eligibility is decided by the library: surface NRX_ERR_UNSUPPORTED now

    def run(self):
        rc = self.lib.nrx_embed_dcn_v1_fwd(self.arr, len(self.plan.slots), self.B, self.W, self.out.data_ptr(),
                                           self.out.shape[1], self.w.shape[0], self.w.data_ptr(), self.b.data_ptr(),
                                           _ptr(self.status), torch.cuda.current_stream(self.device).cuda_stream)
        if rc == -3:
            raise FusedUnsupported(self.lib.nrx_last_error().decode())
        if rc:
            check(rc, "nrx_embed_dcn_v1_fwd")
        return self.out

    def check(self):
        if self.status is not None:
            _raise_if_oob(self.status, self.plan.names)


def fused_cross_is_fast(plan: EmbedPlan) -> bool:
    """True when nrx_embed_dcn_v1_fwd runs its grouped kernel (mirrors the rule in csrc/nrx_interact.hip): plain
    single-valued features, all with the same 32- or 64-wide rows, 2..8 of them, gap-free concat.  Measured at the
    C3 shape: 47.9 us for the one launch vs 60.0 us for gather + cross; other eligible mixes run the
    one-wave-per-sample kernel, which is slower than two launches."""
    sl = plan.slots
    if not (2 <= len(sl) <= 8) or plan.use_fm or plan.wide_width:
        return False
    d0 = sl[0].dim
    col = 0
    for s_ in sl:
        if s_.kind != NRX_SPARSE or s_.wide_col >= 0 or s_.dim != d0 or s_.out_col != col:
            return False
        col += s_.dim
    return d0 in (32, 64) and col == plan.out_width


class _EmbedDcnFn(torch.autograd.Function):
    """Fused gather -> cat[x, cross(x)] with autograd: backward = nrx_dcn_v1_bwd on the saved buffer's x half, then the
    embedding backward (dense, COO or sink -- the same three modes as embed_apply)."""

    @staticmethod
    def forward(ctx, plan: EmbedPlan, inputs, sparse_grad, w, b, *tables):
        if _INDEX_CHECK == "deferred":
            call = PreparedEmbedDcn(plan, tables, inputs, w, b, status=_deferred_status(plan.names))
        else:
            call = PreparedEmbedDcn(plan, tables, inputs, w, b, check_index=_INDEX_CHECK != "off")
            if _INDEX_CHECK == "sync":
                call.check()
            elif _INDEX_CHECK == "lazy":
                flush_index_checks()
                call.check()
        ctx.plan, ctx.B, ctx.ld = plan, call.B, plan.out_width
        ctx.ins, ctx.ws = call.ins, [None] * len(plan.slots)
        ctx.tables_ref = tables          # leaves (the module's parameters): shapes / devices are read from them in the backward
        ctx.has_fm_feat = False
        ctx.sink = sparse_grad if isinstance(sparse_grad, SparseGradSink) else None
        ctx.sparse_grad = bool(sparse_grad)
        ctx.dense_sorted = _dense_sorted_ok(plan, tables, ctx.sparse_grad, call.B) and any(ctx.needs_input_grad[5:])
        ctx.tables = list(tables) if ctx.sink is not None else None
        ctx.save_for_backward(call.out, call.w, call.b)
        ctx.w_shape, ctx.b_shape = tuple(w.shape), tuple(b.shape)
        return call.out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        buf, w, b = ctx.saved_tensors
        g = _f32c(g, "grad")
        B, W2 = buf.shape
        D = W2 // 2
        gx = torch.empty((B, D), dtype=torch.float32, device=buf.device)
        gw, gb = _dcn_v1_grads(w, b)
        _dcn_v1_bwd(lib, (buf.data_ptr(), W2, None, 0, B, D, w.shape[0], w.data_ptr(), b.data_ptr(), g.data_ptr() + 4 * D, W2,
                          gx.data_ptr(), D, None, 0, gw.data_ptr(), gb.data_ptr()), D, w.shape[0], buf.device, _stream_ptr(buf), gw, gb)
        gx += g[:, :D]
        table_grads = _EmbedFn.backward(ctx, gx, None, None)[7:]
        return (None, None, None, gw.view(ctx.w_shape), gb.view(ctx.b_shape), *table_grads)


def embed_dcn_v1(plan: EmbedPlan, tables, inputs, w: torch.Tensor, b: torch.Tensor, sparse_grad=False) -> torch.Tensor:
    """Fused gather -> cat[x, cross(x)] in one launch.  Differentiable w.r.t. tables, w and b when gradients are
    enabled (sparse_grad as in embed_apply); raises FusedUnsupported for feature mixes the kernel does not cover."""
    if torch.is_grad_enabled() and (w.requires_grad or b.requires_grad or any(t.requires_grad for t in tables)):
        return _EmbedDcnFn.apply(plan, list(inputs), sparse_grad, w, b, *tables)
    if _INDEX_CHECK == "deferred":
        return PreparedEmbedDcn(plan, tables, inputs, w, b, status=_deferred_status(plan.names)).out
    call = PreparedEmbedDcn(plan, tables, inputs, w, b, check_index=_INDEX_CHECK != "off")
    if _INDEX_CHECK == "sync":
        call.check()
    return call.out


# ------------------------------------------------------------------------------- DCN v2 (MFMA)
def _dcn_v2_layer_backward(lib, x0, xl, lin, out, relu, W, g, g_x0, accumulate, gW=None, gb=None, math="fp32"):
    """One layer of the hand-written backward (nrx_dcn_v2_layer_bwd: elementwise prep + MFMA dgrad + MFMA wgrad).
    Returns (g_xl, g_W, g_b); g_x0 is written (accumulate bit 0 clear) or accumulated (set) in place, and folded into g_xl when bit 1 is set.
    `gW` / `gb`: contiguous destinations (a layer's slice of the stack's gradient tensors) written in place of fresh tensors."""
    B, D = xl.shape
    g = _f32c(g, "grad")
    g_xl = torch.empty_like(xl)
    if gW is None and gb is None:
        # one allocation, g_b right behind g_W: the library then clears both with ONE fill launch (two ~5 us launches per layer otherwise)
        both = torch.empty((D * D + D,), dtype=torch.float32, device=xl.device)
        gW, gb = both[:D * D].view(D, D), both[D * D:]
    if gW is None:
        gW = torch.empty((D, D), dtype=torch.float32, device=xl.device)
    if gb is None:
        gb = torch.empty((D,), dtype=torch.float32, device=xl.device)
    ws = torch.empty(max(1, lib.nrx_dcn_v2_layer_bwd_workspace(B, D)), dtype=torch.uint8, device=xl.device)
    check(lib.nrx_dcn_v2_layer_bwd(x0.data_ptr(), xl.data_ptr(), D, lin.data_ptr(), _ptr(out), _dcn2_flags(relu, math, WGRAD_ORDERED), B, D, W.data_ptr(),
                                   g.data_ptr(), D, g_xl.data_ptr(), D, g_x0.data_ptr(), D, int(accumulate), gW.data_ptr(),
                                   gb.data_ptr(), ws.data_ptr(), _stream_ptr(xl)), "nrx_dcn_v2_layer_bwd")
    return g_xl, gW, gb


DCN2_MATH = ("fp32", "bf16x3")
# Weight gradients that contract over the batch (DCN cross layers, ops.linear), NRX_WGRAD: "atomic" = every batch slice's / block's partial sums are
# added with float atomics; "ordered" = they are stored and a second launch adds them in a fixed order -- the same bits run to run
# (nrx_linear_wgrad_ordered, nrx_dcn_v1_bwd_ordered, flags bit 2 of nrx_dcn_v2_layer_bwd); "auto" (default) = ordered wherever it is as fast.
# GraphedStep(deterministic=True) captures its step in the ordered mode.
WGRAD_ORDERED = os.environ.get("NRX_WGRAD", "auto") == "ordered"          # everywhere (what GraphedStep(deterministic=True) sets)
WGRAD_ATOMIC = os.environ.get("NRX_WGRAD", "auto") == "atomic"            # nowhere (A/B); the default "auto": ordered where it costs nothing --
                                                                          # ops.linear with a small scratch, every DCN-v1 stack the mode takes, DCN-v2
                                                                          # layers up to 128 wide (decided in the library)
LINEAR_ORDERED_MAX = int(os.environ.get("NRX_LINEAR_ORDERED_MAX", 64 << 20))      # ops.linear: ordered by default while its scratch is at most this many bytes


def _dcn2_flags(relu: bool, math: str, ordered: bool = False) -> int:
    if math not in DCN2_MATH:
        raise ValueError(f"dcn_v2 math must be one of {DCN2_MATH}")
    return (1 if relu else 0) | (2 if math == "bf16x3" else 0) | (4 if ordered else 0)


class _DcnV2Fn(torch.autograd.Function):
    """The whole cross stack.  The per-layer parameters arrive as SEPARATE tensors (W_0 .. W_{n-1}, b_0 .. b_{n-1}: the modules'
    own nn.Linear weights) -- no torch.stack per call, no stacked gradient to un-stack."""

    @staticmethod
    def forward(ctx, x, relu, math, n, *params):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, D = x.shape
        Ws = [_f32c(w, "W") for w in params[:n]]
        bs = [_f32c(v, "b") for v in params[n:]]
        if len(bs) != n or any(tuple(w.shape) != (D, D) for w in Ws) or any(tuple(v.shape) != (D,) for v in bs):
            raise ValueError("dcn_v2: every layer needs W [dim, dim] and b [dim]")
        train = any(ctx.needs_input_grad)       # (grad mode is off inside forward: ask the node, not torch.is_grad_enabled)
        flags = _dcn2_flags(relu, math)
        xs, lins = [x], []
        stream = _stream_ptr(x)
        for l in range(n):
            out = torch.empty_like(x)
            lin = torch.empty_like(x) if train else None      # x_l W^T + b, saved for the backward (one extra write, no GEMM later)
            check(lib.nrx_dcn_v2_layer_fwd(x.data_ptr(), xs[-1].data_ptr(), D, B, D, Ws[l].data_ptr(), bs[l].data_ptr(),
                                           flags, out.data_ptr(), D, _ptr(lin), stream), "nrx_dcn_v2_layer_fwd")
            xs.append(out)
            lins.append(lin)
        ctx.save_for_backward(*Ws, *xs, *[t for t in lins if t is not None])
        ctx.relu, ctx.n, ctx.train, ctx.math = relu, n, train, math
        return xs[-1]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        n = ctx.n
        saved = ctx.saved_tensors
        Ws, xs, lins = saved[:n], saved[n:2 * n + 1], saved[2 * n + 1:]
        if len(lins) != n:
            raise RuntimeError("dcn_v2 backward: the forward ran without gradients enabled")
        x0 = xs[0]
        gx0 = torch.empty_like(x0)
        g = g.contiguous()
        gWs, gbs = [None] * n, [None] * n
        for l in reversed(range(n)):
            # accumulate bit 0: layers before the last add to g_x0 (the last one writes it); bit 1: layer 0 folds the total into
            # its g_xl (x_0 IS the stack's input there): no separate `g + gx0` pass over [B, D]
            acc = (0 if l == n - 1 else 1) | (2 if l == 0 else 0)
            g, gWs[l], gbs[l] = _dcn_v2_layer_backward(lib, x0, xs[l], lins[l], xs[l + 1], ctx.relu, Ws[l], g, gx0, accumulate=acc,
                                                       math=ctx.math)
        return (g, None, None, None, *gWs, *gbs)


class _DcnV2LayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, xl, W, b, relu, math="fp32"):
        lib = _lib.load()
        x0, xl, W, b = _f32c(x0, "x0"), _f32c(xl, "xl"), _f32c(W, "W"), _f32c(b, "b")
        B, D = xl.shape
        if x0.shape != xl.shape or tuple(W.shape) != (D, D) or tuple(b.shape) != (D,):
            raise ValueError("x0/xl must be [B, dim], W [dim, dim], b [dim]")
        out = torch.empty_like(xl)
        train = any(ctx.needs_input_grad[:4])
        lin = torch.empty_like(xl) if train else None
        check(lib.nrx_dcn_v2_layer_fwd(x0.data_ptr(), xl.data_ptr(), D, B, D, W.data_ptr(), b.data_ptr(),
                                       _dcn2_flags(relu, math), out.data_ptr(), D, _ptr(lin), _stream_ptr(xl)), "nrx_dcn_v2_layer_fwd")
        ctx.save_for_backward(x0, xl, W, out, *((lin,) if train else ()))
        ctx.relu, ctx.math = relu, math
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x0, xl, W, out, *rest = ctx.saved_tensors
        if not rest:
            raise RuntimeError("dcn_v2_layer backward: the forward ran without gradients enabled")
        gx0 = torch.empty_like(x0)
        g_xl, gW, gb = _dcn_v2_layer_backward(lib, x0, xl, rest[0], out, ctx.relu, W, g, gx0, accumulate=False, math=ctx.math)
        return gx0, g_xl, gW, gb, None, None


def dcn_v2_layer(x0: torch.Tensor, xl: torch.Tensor, W: torch.Tensor, b: torch.Tensor, relu: bool = False, math: str = "fp32") -> torch.Tensor:
    """One DCNv2Layer (dcn_arch.py:39-50): act(x0 * (xl W^T + b) + xl) on MFMA.  math: "fp32" (default: the fp32 fma chain, value-
    exact against the C oracle) or "bf16x3" (each operand split into two bfloat16 parts, three bf16 MFMAs, fp32 accumulate)."""
    return _DcnV2LayerFn.apply(x0, xl, W, b, relu, math)


def dcn_v2(x: torch.Tensor, W, b, relu: bool = True, math: str = "fp32") -> torch.Tensor:
    """DCNv2Net.forward (dcn_arch.py:83-91): x <- relu(x0 * (x W_l^T + b_l) + x) per layer, on MFMA.
    W / b: stacked tensors [n, dim, dim] / [n, dim], or sequences of the per-layer tensors (what the module passes: no stack)."""
    Ws = list(W.unbind(0)) if isinstance(W, torch.Tensor) else list(W)
    bs = list(b.unbind(0)) if isinstance(b, torch.Tensor) else list(b)
    if len(Ws) != len(bs):
        raise ValueError("W must be [n_layers, dim, dim] and b [n_layers, dim]")
    if isinstance(W, torch.Tensor) and (W.dim() != 3 or W.shape[1] != W.shape[2] or tuple(b.shape) != tuple(W.shape[:2])):
        raise ValueError("W must be [n_layers, dim, dim] and b [n_layers, dim]")
    if not Ws:
        return x
    return _DcnV2Fn.apply(x, relu, math, len(Ws), *Ws, *bs)


# ------------------------------------------------------------------------------- integer utilities
class _LinearFn(torch.autograd.Function):
    """y = a W^T + b with the weight and bias gradients on nrx_linear_wgrad (the batch is the contraction there: at B = 65 536 the vendor
    GEMM takes 0.2-0.3 ms per MLP layer, the split-over-the-batch MFMA kernel of the DCN-v2 backward a fraction of that).  The
    forward and the input gradient stay on the vendor GEMM.  Reference layer: src/model/model_utils/utils.py:6-17."""

    @staticmethod
    def forward(ctx, a, W, b):
        ctx.save_for_backward(a, W)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(a, W, b)

    @staticmethod
    def backward(ctx, g):
        a, W = ctx.saved_tensors
        lib = _lib.load()
        g2 = _f32c(g.reshape(-1, g.shape[-1]), "grad")
        a2 = a.reshape(-1, a.shape[-1])
        if a2.stride(-1) != 1:
            a2 = a2.contiguous()
        ga = gW = gb = None
        if ctx.needs_input_grad[0]:
            ga = (g2 @ W).reshape(a.shape)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gW = torch.empty_like(W, memory_format=torch.contiguous_format)
            if want_b:                        # the bias gradient comes out of the same pass over g
                gb = torch.empty((W.shape[0],), dtype=torch.float32, device=W.device)
            # ordered (bit-reproducible) whenever its scratch is small: measured as fast or faster than the atomic form at every MLP shape tried
            # (B = 65 536: [128, 128] 32.5 vs 37.9 us, [64, 128] 23.9 vs 27.8, [128, 416] 95.4 vs 95.5, [512, 416] 298.8 vs 292.0 -- the atomic form
            # needs two fill launches, the ordered one a reduction; tools/ab_linear_wgrad.py)
            wsz = lib.nrx_linear_wgrad_ordered_workspace(g2.shape[0], W.shape[0], W.shape[1])
            if WGRAD_ORDERED or (not WGRAD_ATOMIC and wsz <= LINEAR_ORDERED_MAX):
                ws = torch.empty(max(1, wsz), dtype=torch.uint8, device=W.device)
                check(lib.nrx_linear_wgrad_ordered(g2.data_ptr(), g2.stride(0), a2.data_ptr(), a2.stride(0), g2.shape[0], W.shape[0], W.shape[1],
                                                   gW.data_ptr(), _ptr(gb), ws.data_ptr(), _stream_ptr(g2)), "nrx_linear_wgrad_ordered")
            else:
                check(lib.nrx_linear_wgrad(g2.data_ptr(), g2.stride(0), a2.data_ptr(), a2.stride(0), g2.shape[0], W.shape[0], W.shape[1],
                                           gW.data_ptr(), _ptr(gb), _stream_ptr(g2)), "nrx_linear_wgrad")
        elif want_b:
            gb = g2.sum(0)
        return ga, gW, gb


def linear(a: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.nn.functional.linear whose weight gradient runs on this package's split-over-the-batch MFMA kernel (fp32, CUDA only)."""
    _dev(a, "linear input")
    if a.dtype != torch.float32 or W.dtype != torch.float32:
        raise TypeError("ops.linear computes in fp32")
    return _LinearFn.apply(a, W, b)


def bucketize_by_owner(ids: torch.Tensor, world: int):
    """Stable bucketing of a flat id tensor by owner rank (id % world).
    Returns (counts[world] int64, local_rows[n] int64 (= id // world in send order), slot[n] int64)."""
    lib = _lib.load()
    _dev(ids, "ids")
    ids = ids.contiguous().view(-1)
    if ids.dtype not in (torch.int64, torch.int32):
        ids = ids.long()
    n = ids.numel()
    dev = ids.device
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    local_rows = torch.empty(n, dtype=torch.int64, device=dev)
    slot = torch.empty(n, dtype=torch.int64, device=dev)
    ws = torch.empty(max(1, lib.nrx_bucketize_workspace(n, world)), dtype=torch.int64, device=dev)
    check(lib.nrx_bucketize_by_owner(ids.data_ptr(), ids.element_size() * 8, n, world, counts.data_ptr(),
                                     local_rows.data_ptr(), slot.data_ptr(), ws.data_ptr(), _stream_ptr(ids)),
          "nrx_bucketize_by_owner")
    return counts, local_rows, slot


def gather_rows_segmented(tables: Sequence[torch.Tensor], seg_start: torch.Tensor, seg_table: torch.Tensor,
                          local_rows: torch.Tensor, n_rows: int, check_index: bool = True, defer_check: bool = False):
    """Owner-side gather: rows of tables[seg_table[s]] for local_rows[seg_start[s]:seg_start[s+1]].
    defer_check=True returns (rows, status) without reading the status back (the sharded exchange
    checks it collectively after the return all-to-all); status is None when index checks are off."""
    lib = _lib.load()
    dim = tables[0].shape[1]
    for t in tables:
        _f32c(t, "table")
        if t.shape[1] != dim or not t.is_contiguous():
            raise ValueError("segmented gather needs contiguous tables of one common dim")
    dev = tables[0].device
    out = torch.empty((n_rows, dim), dtype=torch.float32, device=dev)
    if defer_check:
        check_index = _INDEX_CHECK != "off"
    if n_rows == 0:
        if defer_check:
            return out, (torch.zeros(4, dtype=torch.int32, device=dev) if check_index else None)
        return out
    seg_start = _dev(seg_start, "seg_start").to(torch.int64).contiguous()
    seg_table = _dev(seg_table, "seg_table").to(torch.int32).contiguous()
    local_rows = _dev(local_rows, "local_rows").to(torch.int64).contiguous()
    n_seg = seg_table.numel()
    tp = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
    tr = (C.c_int64 * len(tables))(*[t.shape[0] for t in tables])
    status = torch.zeros(4, dtype=torch.int32, device=dev) if check_index else None
    check(lib.nrx_gather_rows_segmented(tp, tr, len(tables), seg_start.data_ptr(), seg_table.data_ptr(), n_seg,
                                        n_rows, dim, local_rows.data_ptr(), out.data_ptr(), _ptr(status),
                                        _stream_ptr(out)), "nrx_gather_rows_segmented")
    if defer_check:
        return out, status
    if status is not None:
        st = status.tolist()
        if st[0] != 0:
            raise IndexError(f"index out of range in self: {st[0]} routed lookup(s); first: table #{st[1]}, "
                             f"position {st[2]}, local row {st[3]}")
    return out


def scatter_add_rows_segmented(grad_tables: Sequence[torch.Tensor], seg_start: torch.Tensor, seg_table: torch.Tensor,
                               local_rows: torch.Tensor, g_rows: torch.Tensor, skip_row0: bool) -> None:
    """Owner-side backward of gather_rows_segmented: grad_tables[seg_table[s]][local_rows[p]] += g_rows[p]."""
    lib = _lib.load()
    n_rows, dim = g_rows.shape
    if n_rows == 0:
        return
    g_rows = _f32c(g_rows, "g_rows")
    for t in grad_tables:
        _f32c(t, "grad table")
        if t.shape[1] != dim or not t.is_contiguous():
            raise ValueError("segmented scatter needs contiguous grad tables of one common dim")
    seg_start = _dev(seg_start, "seg_start").to(torch.int64).contiguous()
    seg_table = _dev(seg_table, "seg_table").to(torch.int32).contiguous()
    local_rows = _dev(local_rows, "local_rows").to(torch.int64).contiguous()
    tp = (C.c_void_p * len(grad_tables))(*[t.data_ptr() for t in grad_tables])
    tr = (C.c_int64 * len(grad_tables))(*[t.shape[0] for t in grad_tables])
    check(lib.nrx_scatter_add_rows_segmented(tp, tr, len(grad_tables), seg_start.data_ptr(), seg_table.data_ptr(),
                                             seg_table.numel(), n_rows, dim, local_rows.data_ptr(), g_rows.data_ptr(),
                                             1 if skip_row0 else 0, _stream_ptr(g_rows)), "nrx_scatter_add_rows_segmented")


def route_ids(id_tensors: Sequence[torch.Tensor], world: int, cap: int, want_pos: bool = False):
    """Fixed-capacity routing of the ids of one exchange (no host synchronisation).
    Returns (send_rows [world*cap] int32, slot [N] int32, counts2d [world, F] int64, overflow [1] int64); with want_pos
    (nrx_route_ids_pos, one-sided placement) also send_pos [world*cap] int32: the position of every sent id inside its feature."""
    lib = _lib.load()
    n = len(id_tensors)
    if not 1 <= n <= NRX_MAX_FEATURES:
        raise ValueError(f"route_ids takes 1..{NRX_MAX_FEATURES} id tensors per exchange")
    dt = id_tensors[0].dtype
    if dt not in (torch.int64, torch.int32):
        raise TypeError("ids must be int64 or int32")
    xs = []
    for x in id_tensors:
        _dev(x, "ids")
        if x.dtype != dt:
            raise TypeError("all id tensors of one exchange must share a dtype")
        xs.append(x if x.is_contiguous() else x.contiguous())
    dev = xs[0].device
    total = sum(x.numel() for x in xs)
    send_rows = torch.empty(world * cap, dtype=torch.int32, device=dev)
    slot = torch.empty(total, dtype=torch.int32, device=dev)
    counts2d = torch.empty((world, n), dtype=torch.int64, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(1, lib.nrx_route_workspace(total, world)), dtype=torch.int64, device=dev)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    lens = (C.c_int64 * n)(*[x.numel() for x in xs])
    if want_pos:
        send_pos = torch.empty(world * cap, dtype=torch.int32, device=dev)
        check(lib.nrx_route_ids_pos(ptrs, lens, n, xs[0].element_size() * 8, world, cap, send_rows.data_ptr(), send_pos.data_ptr(),
                                    slot.data_ptr(), counts2d.data_ptr(), overflow.data_ptr(), ws.data_ptr(), _stream_ptr(xs[0])),
              "nrx_route_ids_pos")
        return send_rows, slot, counts2d, overflow, send_pos
    check(lib.nrx_route_ids(ptrs, lens, n, xs[0].element_size() * 8, world, cap, send_rows.data_ptr(), slot.data_ptr(),
                            counts2d.data_ptr(), overflow.data_ptr(), ws.data_ptr(), _stream_ptr(xs[0])), "nrx_route_ids")
    return send_rows, slot, counts2d, overflow


def route_ids_dedup(id_tensors: Sequence[torch.Tensor], table_of: Sequence[int], table_local_rows: Sequence[int], world: int, cap: int):
    """De-duplicated fixed-capacity routing (nrx_route_ids_dedup): every distinct (owner, table, row) is sent once.
    Returns (send_rows int32 [world*cap], slot int32 [N], counts2d int64 [world, n_tables], overflow int64 [1])."""
    lib = _lib.load()
    n, nt = len(id_tensors), len(table_local_rows)
    dt = id_tensors[0].dtype
    if dt not in (torch.int64, torch.int32) or any(x.dtype != dt for x in id_tensors):
        raise TypeError("route_ids_dedup: ids must share one integer dtype (int64 or int32)")
    xs = [_dev(x, "ids") if x.is_contiguous() else x.contiguous() for x in id_tensors]
    dev = xs[0].device
    total = sum(x.numel() for x in xs)
    send_rows = torch.empty(world * cap, dtype=torch.int32, device=dev)
    slot = torch.empty(total, dtype=torch.int32, device=dev)
    counts2d = torch.empty((world, nt), dtype=torch.int64, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(1, lib.nrx_route_dedup_workspace(total, world)), dtype=torch.uint8, device=dev)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    lens = (C.c_int64 * n)(*[x.numel() for x in xs])
    tof = (C.c_int32 * n)(*[int(t) for t in table_of])
    tlr = (C.c_int64 * nt)(*[int(r) for r in table_local_rows])
    check(lib.nrx_route_ids_dedup(ptrs, lens, tof, tlr, n, nt, xs[0].element_size() * 8, world, cap, send_rows.data_ptr(),
                                  slot.data_ptr(), counts2d.data_ptr(), overflow.data_ptr(), ws.data_ptr(), _stream_ptr(xs[0])),
          "nrx_route_ids_dedup")
    return send_rows, slot, counts2d, overflow


def unique_inverse(ids: torch.Tensor):
    """np.unique(ids, return_inverse=True) on the device (nrx_unique_inverse).  Returns (unique [n_unique] int64 ascending,
    inverse [ids.shape] int64).  One host read (the count) to size the result."""
    lib = _lib.load()
    _dev(ids, "ids")
    if ids.dtype not in (torch.int64, torch.int32):
        ids = ids.long()
    flat = ids.contiguous().view(-1)
    n = flat.numel()
    dev = flat.device
    uniq = torch.empty(n, dtype=torch.int64, device=dev)
    inv = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(1, lib.nrx_unique_inverse_workspace(n)), dtype=torch.uint8, device=dev)
    check(lib.nrx_unique_inverse(_ptr(flat) if n else None, flat.element_size() * 8, n, uniq.data_ptr(), inv.data_ptr(), cnt.data_ptr(),
                                 ws.data_ptr(), _stream_ptr(flat)), "nrx_unique_inverse")
    return uniq[:int(cnt.item())], inv.view(ids.shape)


def _inbox_common(tables, feat_table):
    dim = tables[0].shape[1]
    for t in tables:
        _f32c(t, "table")
        if t.shape[1] != dim or not t.is_contiguous():
            raise ValueError("inbox gather/scatter needs contiguous tables of one common dim")
    tp = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
    tr = (C.c_int64 * len(tables))(*[t.shape[0] for t in tables])
    ft = (C.c_int32 * len(feat_table))(*feat_table)
    return dim, tp, tr, ft


def gather_inbox(tables: Sequence[torch.Tensor], feat_table: Sequence[int], world: int, cap: int,
                 recv2d: torch.Tensor, inbox_rows: torch.Tensor, status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Owner side of the fixed-capacity exchange: rows for the valid prefix of every source block.
    Returns [world*cap, dim] (slots past a block's count are left uninitialised)."""
    lib = _lib.load()
    dim, tp, tr, ft = _inbox_common(tables, feat_table)
    if inbox_rows.dtype != torch.int32:
        inbox_rows = inbox_rows.to(torch.int32)        # the wire format is int32 local rows
    out = torch.empty((world * cap, dim), dtype=torch.float32, device=tables[0].device)
    check(lib.nrx_gather_inbox(tp, tr, len(tables), ft, len(feat_table), world, cap, recv2d.data_ptr(),
                               inbox_rows.data_ptr(), dim, out.data_ptr(), _ptr(status), _stream_ptr(out)),
          "nrx_gather_inbox")
    return out


def scatter_add_inbox(grad_tables: Sequence[torch.Tensor], feat_table: Sequence[int], world: int, cap: int,
                      recv2d: torch.Tensor, inbox_rows: torch.Tensor, g_rows: torch.Tensor, skip_row0: bool) -> None:
    lib = _lib.load()
    dim, tp, tr, ft = _inbox_common(grad_tables, feat_table)
    if inbox_rows.dtype != torch.int32:
        inbox_rows = inbox_rows.to(torch.int32)
    g_rows = _f32c(g_rows, "g_rows")
    check(lib.nrx_scatter_add_inbox(tp, tr, len(grad_tables), ft, len(feat_table), world, cap, recv2d.data_ptr(),
                                    inbox_rows.data_ptr(), dim, g_rows.data_ptr(), 1 if skip_row0 else 0,
                                    _stream_ptr(g_rows)), "nrx_scatter_add_inbox")


def bag_norm_weights(mask: Optional[torch.Tensor], batch: int, bag_len: int, kind: int, device=None) -> torch.Tensor:
    """Per-lookup pooling weights with the normalisation folded in (see nrx_bag_norm_weights): [batch, bag_len] float32."""
    lib = _lib.load()
    if mask is not None:
        mask = _f32c(mask, "mask")
        device = mask.device
    out = torch.empty((batch, bag_len), dtype=torch.float32, device=device)
    check(lib.nrx_bag_norm_weights(_ptr(mask), batch, bag_len, kind, out.data_ptr(), torch.cuda.current_stream(out.device).cuda_stream),
          "nrx_bag_norm_weights")
    return out


def route_bags(id_tensors: Sequence[torch.Tensor], weights: Sequence[Optional[torch.Tensor]], world: int, cap: int):
    """Pooled-bag routing (nrx_route_bags).  id_tensors: [B, L_f] each, one dtype; weights: normalised [B, L_f] or None.
    Returns (send_rows int32 [world*cap], send_tag int32, send_w float32, counts2d int64 [world, F], overflow int64 [1])."""
    lib = _lib.load()
    n = len(id_tensors)
    dt = id_tensors[0].dtype
    if dt not in (torch.int64, torch.int32) or any(x.dtype != dt for x in id_tensors):
        raise TypeError("route_bags: ids must share one integer dtype (int64 or int32)")
    xs = [_dev(x, "ids") if x.is_contiguous() else x.contiguous() for x in id_tensors]
    B = xs[0].shape[0]
    ws = [None if w is None else _f32c(w, "weights") for w in weights]
    dev = xs[0].device
    total = sum(x.numel() for x in xs)
    send_rows = torch.empty(world * cap, dtype=torch.int32, device=dev)
    send_tag = torch.empty(world * cap, dtype=torch.int32, device=dev)
    send_w = torch.empty(world * cap, dtype=torch.float32, device=dev)
    counts2d = torch.empty((world, n), dtype=torch.int64, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = torch.empty(max(1, lib.nrx_route_workspace(total, world)), dtype=torch.int64, device=dev)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    wptrs = (C.c_void_p * n)(*[(_ptr(w) or 0) for w in ws])
    bl = (C.c_int32 * n)(*[x.shape[1] for x in xs])
    check(lib.nrx_route_bags(ptrs, wptrs, bl, n, xs[0].element_size() * 8, B, world, cap, send_rows.data_ptr(), send_tag.data_ptr(),
                             send_w.data_ptr(), counts2d.data_ptr(), overflow.data_ptr(), wsb.data_ptr(), _stream_ptr(xs[0])),
          "nrx_route_bags")
    return send_rows, send_tag, send_w, counts2d, overflow


def pool_inbox(tables: Sequence[torch.Tensor], feat_table: Sequence[int], batch: int, world: int, cap: int, recv2d, inbox_rows,
               inbox_tag, inbox_w, status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Owner-side partial pooling (nrx_pool_inbox_fwd): returns partial [world, n_feats*batch, dim] (zeros where this rank
    holds nothing of a (feature, sample))."""
    lib = _lib.load()
    dim, tp, tr, ft = _inbox_common(tables, feat_table)
    nf = len(feat_table)
    partial = torch.empty((world, nf * batch, dim), dtype=torch.float32, device=tables[0].device)
    ws = torch.empty(max(1, lib.nrx_pool_inbox_workspace(nf, batch, world)), dtype=torch.uint8, device=tables[0].device)
    check(lib.nrx_pool_inbox_fwd(tp, tr, len(tables), ft, nf, batch, world, cap, recv2d.data_ptr(), inbox_rows.data_ptr(),
                                 inbox_tag.data_ptr(), inbox_w.data_ptr(), dim, partial.data_ptr(), ws.data_ptr(), _ptr(status),
                                 _stream_ptr(partial)), "nrx_pool_inbox_fwd")
    return partial


def pool_inbox_bwd(grad_tables: Sequence[torch.Tensor], feat_table: Sequence[int], batch: int, world: int, cap: int, recv2d,
                   inbox_rows, inbox_tag, inbox_w, g_partial: torch.Tensor, skip_row0: bool) -> None:
    lib = _lib.load()
    dim, tp, tr, ft = _inbox_common(grad_tables, feat_table)
    g_partial = _f32c(g_partial, "g_partial")
    check(lib.nrx_pool_inbox_bwd(tp, tr, len(grad_tables), ft, len(feat_table), batch, world, cap, recv2d.data_ptr(),
                                 inbox_rows.data_ptr(), inbox_tag.data_ptr(), inbox_w.data_ptr(), dim, g_partial.data_ptr(),
                                 1 if skip_row0 else 0, _stream_ptr(g_partial)), "nrx_pool_inbox_bwd")


def csr_to_padded(values: torch.Tensor, offsets: torch.Tensor, bag_len: int, rows: Optional[torch.Tensor] = None):
    """CSR batch of an array feature -> (ids [B, bag_len] 0-padded, mask float32 [B, bag_len]) on the device:
    the padded form DataReader builds per sample on the host (data_reader.py:96-109).  With `rows` (device
    int64 [B]) the CSR is a whole dataset resident in HBM and batch row b is its row rows[b]."""
    lib = _lib.load()
    _dev(values, "values")
    _dev(offsets, "offsets")
    if values.dtype not in (torch.int64, torch.int32):
        raise TypeError("values must be int64 or int32")
    offsets = offsets.to(torch.int64).contiguous()
    values = values.contiguous()
    if rows is not None:
        rows = _dev(rows, "rows").to(torch.int64).contiguous()
        B = rows.numel()
    else:
        B = offsets.numel() - 1
    ids = torch.empty((B, bag_len), dtype=values.dtype, device=values.device)
    mask = torch.empty((B, bag_len), dtype=torch.float32, device=values.device)
    check(lib.nrx_csr_to_padded(_ptr(values) if values.numel() else None, values.element_size() * 8, offsets.data_ptr(), _ptr(rows), B,
                                bag_len, ids.data_ptr(), mask.data_ptr(), _stream_ptr(offsets)), "nrx_csr_to_padded")
    return ids, mask


def mask_lengths(mask: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    mask = _f32c(mask, "mask")
    B, L = mask.shape
    lens = torch.empty(B, dtype=torch.int64, device=mask.device)
    check(lib.nrx_mask_lengths(mask.data_ptr(), B, L, lens.data_ptr(), _stream_ptr(mask)), "nrx_mask_lengths")
    return lens


TOPK_KMAX = 32          # list length one launch keeps in registers (nrx_topk.hip)


def _topk_ip_once(items, queries, k, exclude):
    lib = _lib.load()
    N, d = items.shape
    Q = queries.shape[0]
    out_idx = torch.empty((Q, k), dtype=torch.int64, device=queries.device)
    out_score = torch.empty((Q, k), dtype=torch.float32, device=queries.device)
    nbytes = lib.nrx_topk_workspace(N, Q, k)
    if nbytes < 0:
        raise ValueError("topk_ip: bad sizes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=queries.device)
    eo = ei = None
    if exclude is not None:
        eo, ei = exclude
        eo = eo.to(device=queries.device, dtype=torch.int64).contiguous()
        ei = ei.to(device=queries.device, dtype=torch.int64).contiguous()
        if eo.numel() != Q + 1:
            raise ValueError("topk_ip: exclude offsets must have n_queries + 1 entries")
    check(lib.nrx_topk_ip(items.data_ptr(), N, d, queries.data_ptr(), Q, k,
                          eo.data_ptr() if eo is not None else None,
                          ei.data_ptr() if ei is not None and ei.numel() else (ws.data_ptr() if eo is not None else None),
                          out_idx.data_ptr(), out_score.data_ptr(), ws.data_ptr(), _stream_ptr(queries)), "nrx_topk_ip")
    return out_idx, out_score


def topk_ip(items: torch.Tensor, queries: torch.Tensor, k: int, exclude=None):
    """Exact inner-product top-k (faiss.IndexFlatIP.search semantics; TopKSearcher.py:50-84).
    items [N, d], queries [Q, d] fp32 on the GPU.  `exclude` = (offsets [Q+1], item_idx) device int64
    CSR of per-query item positions to skip (each list ascending).  Returns (idx [Q, k] int64, score
    [Q, k] fp32): scores descending, ties toward the lower index, empty slots -1 / -FLT_MAX.
    k > 32 runs ceil(k / 32) passes, each excluding what the earlier ones returned (same result as one
    pass: the order (score desc, index asc) is total)."""
    items = _f32c(items, "items")
    queries = _f32c(queries, "queries")
    if items.dim() != 2 or queries.dim() != 2 or items.shape[1] != queries.shape[1]:
        raise ValueError(f"topk_ip: items {tuple(items.shape)} vs queries {tuple(queries.shape)}")
    if k < 1:
        raise ValueError("topk_ip: k must be >= 1")
    if k <= TOPK_KMAX or queries.shape[0] == 0:
        return _topk_ip_once(items, queries, k, exclude)
    Q = queries.shape[0]
    dev = queries.device
    BIG = torch.iinfo(torch.int64).max
    if exclude is not None:                                  # CSR -> padded [Q, L] with BIG as filler
        eo = exclude[0].to(device=dev, dtype=torch.int64)
        ei = exclude[1].to(device=dev, dtype=torch.int64)
        lens = eo[1:] - eo[:-1]
        L = int(lens.max().item()) if Q else 0
        acc = torch.full((Q, L), BIG, dtype=torch.int64, device=dev)
        if ei.numel():
            row = torch.repeat_interleave(torch.arange(Q, device=dev), lens)
            col = torch.arange(ei.numel(), device=dev) - torch.repeat_interleave(eo[:-1], lens)
            acc[row, col] = ei
    else:
        acc = torch.empty((Q, 0), dtype=torch.int64, device=dev)
    idxs, scs = [], []
    done = 0
    excl = exclude
    while done < k:
        kk = min(TOPK_KMAX, k - done)
        idx, sc = _topk_ip_once(items, queries, kk, excl)
        idxs.append(idx)
        scs.append(sc)
        done += kk
        if done < k:
            acc = torch.sort(torch.cat([acc, torch.where(idx >= 0, idx, torch.full_like(idx, BIG))], dim=1), dim=1).values
            valid = acc != BIG
            offs = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
            torch.cumsum(valid.sum(dim=1), 0, out=offs[1:])
            excl = (offs, acc[valid])
    return torch.cat(idxs, dim=1), torch.cat(scs, dim=1)


def device_info(device: int = 0):
    lib = _lib.load()
    info = (C.c_int64 * 6)()
    check(lib.nrx_device_info(device, info), "nrx_device_info")
    return {"compute_units": info[0], "wavefront": info[1], "clock_khz": info[2], "global_mem_bytes": info[3],
            "mem_clock_khz": info[4], "mem_bus_bits": info[5]}
