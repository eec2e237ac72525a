"""ctypes wrapper of oracle/nrx_oracle.c (OpenMP C restatement) -- TEST INFRASTRUCTURE ONLY.
Used by bench.py's cpu_baseline leg (all host cores) and tests/test_oracle_golden.py."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
SPARSE, DENSE, BAG_MASKED_MEAN, BAG_MEAN = 0, 1, 2, 3


class OFeature(C.Structure):
    _fields_ = [("table", C.c_void_p), ("index", C.c_void_p), ("weight", C.c_void_p), ("rows", C.c_int64),
                ("dim", C.c_int32), ("bag_len", C.c_int32), ("kind", C.c_int32), ("out_col", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(LIB)
        _lib.oracle_embed_concat.restype = C.c_int64
        _lib.oracle_embed_concat.argtypes = [C.POINTER(OFeature), C.c_int32, C.c_int64, C.c_void_p, C.c_int64]
        _lib.oracle_fm_logit.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
        _lib.oracle_dcn_v1.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64]
        _lib.oracle_threads.restype = C.c_int
        _lib.oracle_dcn_v2_layer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_int32, C.c_void_p, C.c_int64]
        _lib.oracle_topk_ip.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def threads():
    return load().oracle_threads()


def set_threads(n):
    load().oracle_set_threads(int(n))


class EmbedCall:
    """Bound call: features = list of dict(kind, table (np f32 [rows,D]) | None, index (np), weight (np|None)), in
    output order (the caller sorts like base_model.py:286)."""

    def __init__(self, feats, B):
        lib = load()
        self.keep = []
        self.arr = (OFeature * len(feats))()
        col = 0
        self.dims = []
        for a, f in zip(self.arr, feats):
            kind = f["kind"]
            if kind == DENSE:
                idx = np.ascontiguousarray(f["index"], np.float64)
                a.table, a.rows, a.dim, a.bag_len = None, 0, 1, 0
            else:
                t = np.ascontiguousarray(f["table"], np.float32)
                idx = np.ascontiguousarray(f["index"], np.int64)
                self.keep.append(t)
                a.table, a.rows, a.dim = t.ctypes.data, t.shape[0], t.shape[1]
                a.bag_len = idx.shape[1] if idx.ndim == 2 else 0
            w = f.get("weight")
            if w is not None:
                w = np.ascontiguousarray(w, np.float32)
                self.keep.append(w)
                a.weight = w.ctypes.data
            self.keep.append(idx)
            a.index, a.kind, a.out_col = idx.ctypes.data, kind, col
            col += a.dim
            self.dims.append(a.dim)
        self.B, self.width = B, col
        self.out = np.empty((B, col), np.float32)
        self.lib = lib

    def run(self):
        bad = self.lib.oracle_embed_concat(self.arr, len(self.arr), self.B, self.out.ctypes.data, self.width)
        if bad:
            raise IndexError("index out of range in self")
        return self.out


def fm_logit(feat, n_fields, dim):
    feat = np.ascontiguousarray(feat, np.float32)
    out = np.empty(feat.shape[0], np.float32)
    load().oracle_fm_logit(feat.ctypes.data, feat.shape[1], n_fields, dim, feat.shape[0], out.ctypes.data)
    return out


def dcn_v1(x, w, b):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty_like(x)
    load().oracle_dcn_v1(x.ctypes.data, x.shape[1], x.shape[0], x.shape[1], w.shape[0], w.ctypes.data, b.ctypes.data,
                         out.ctypes.data, x.shape[1])
    return out


def topk_ip(items, queries, k, exclude=None):
    """oracle_topk_ip: (idx [Q,k] int64, score [Q,k] f32); exclude = (offsets [Q+1], item_idx) int64 CSR."""
    lib = load()
    items = np.ascontiguousarray(items, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    Q, d = queries.shape
    idx = np.empty((Q, k), np.int64)
    score = np.empty((Q, k), np.float32)
    eo = ei = None
    if exclude is not None:
        eo = np.ascontiguousarray(exclude[0], np.int64)
        ei = np.ascontiguousarray(exclude[1], np.int64)
        if ei.size == 0:
            ei = np.zeros(1, np.int64)
    lib.oracle_topk_ip(items.ctypes.data, items.shape[0], d, queries.ctypes.data, Q, k,
                       eo.ctypes.data if eo is not None else None, ei.ctypes.data if ei is not None else None,
                       idx.ctypes.data, score.ctypes.data)
    return idx, score


def dcn_v2(x, W, b, relu=True):
    """All layers of DCNv2Net through oracle_dcn_v2_layer (the kernel's exact fp32 order).  x [B, D], W [n, D, D], b [n, D]."""
    lib = load()
    x0 = np.ascontiguousarray(x, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    B, D = x0.shape
    xl = x0
    for l in range(W.shape[0]):
        out = np.empty_like(x0)
        lib.oracle_dcn_v2_layer(x0.ctypes.data, xl.ctypes.data, D, B, D, W[l].ctypes.data, b[l].ctypes.data, int(relu),
                                out.ctypes.data, D)
        xl = out
    return xl
