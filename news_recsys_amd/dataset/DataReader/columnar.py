"""Columnar binary form of a feature file + a batch loader that feeds the GPU path.

`convert_features_txt` parses a reference-format text file ONCE (same grammar as DataReader /
data_reader.py:54-115) into one flat array per column under a directory:

    meta.json                       names, dtypes, max lengths, sample count, label width
    sparse/<name>.npy               int32 (or int64 when an id needs it)   [N]
    dense/<name>.npy                float64                               [N]
    array/<name>.values.npy         int32/int64, truncated to max_len      [sum len]   (CSR)
    array/<name>.offsets.npy        int64                                  [N + 1]
    label.npy                       float32                                [N, n_labels]

`ColumnarLoader` memory-maps them and yields the batch dict the models consume (the collated form of
DataReader items: sparse int[B], dense float64[B], array int[B, L] + `<name>_mask` float32[B, L],
`label` float32[B, n_labels]) already on the device: host batches are assembled into pinned buffers,
copied with an async H2D on a side stream (double-buffered), and array features travel as compact
CSR and are expanded to padded ids + mask on the GPU by `nrx_csr_to_padded` (less PCIe traffic than
the padded form: only the valid ids cross the link).  Narrow (int32) ids are kept narrow end to end
-- the kernels accept them (`index_bits = 32`), like the reference accepts any integer dtype through
`.long()` (base_model.py:271)."""
from __future__ import annotations

import json
import os
from typing import Optional, Dict, Iterator, List

import numpy as np
import torch

from ...config import load_config
from .data_reader import parse_feature_line

FORMAT_VERSION = 1


def convert_features_txt(config_path: str, txt_path: str, out_dir: str, index_dtype: str = "auto") -> Dict:
    cfg = load_config(config_path)
    sparse = list(cfg.features.sparse_feature_names or [])
    dense = list(cfg.features.dense_feature_names or [])
    array = list(cfg.features.array_feature_names or [])
    max_len = dict(cfg.features.array_max_length or {})
    s_set, d_set, a_set = set(sparse), set(dense), set(array)
    cols_s = {n: [] for n in sparse}
    cols_d = {n: [] for n in dense}
    vals_a = {n: [] for n in array}
    offs_a = {n: [0] for n in array}
    labels: List[List[float]] = []
    n = 0
    with open(txt_path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            raw = parse_feature_line(line, n, s_set, d_set, a_set, max_len)
            for k in sparse:
                if k not in raw:
                    raise ValueError(f"Line {n}: sparse feature '{k}' missing (columnar files need every column)")
                cols_s[k].append(raw[k])
            for k in dense:
                if k not in raw:
                    raise ValueError(f"Line {n}: dense feature '{k}' missing (columnar files need every column)")
                cols_d[k].append(raw[k])
            for k in array:
                if k not in raw:
                    raise ValueError(f"Line {n}: array feature '{k}' missing (columnar files need every column)")
                ids = raw[k][: max_len[k]]
                vals_a[k].extend(ids)
                offs_a[k].append(len(vals_a[k]))
            labels.append(raw["label"])
            n += 1
    if n and len({len(l) for l in labels}) != 1:
        raise ValueError("all lines must carry the same number of labels")

    def idt(values) -> str:
        if index_dtype in ("int32", "int64"):
            return index_dtype
        mx = max(values) if len(values) else 0
        mn = min(values) if len(values) else 0
        return "int32" if (mx < 2 ** 31 and mn >= -2 ** 31) else "int64"

    for sub in ("sparse", "dense", "array"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    meta = {"version": FORMAT_VERSION, "n": n, "sparse": {}, "dense": dense, "array": {},
            "n_labels": len(labels[0]) if n else 0}
    for k in sparse:
        dt = idt(cols_s[k])
        np.save(os.path.join(out_dir, "sparse", k + ".npy"), np.asarray(cols_s[k], dtype=dt))
        meta["sparse"][k] = dt
    for k in dense:
        np.save(os.path.join(out_dir, "dense", k + ".npy"), np.asarray(cols_d[k], dtype=np.float64))
    for k in array:
        dt = idt(vals_a[k])
        np.save(os.path.join(out_dir, "array", k + ".values.npy"), np.asarray(vals_a[k], dtype=dt))
        np.save(os.path.join(out_dir, "array", k + ".offsets.npy"), np.asarray(offs_a[k], dtype=np.int64))
        meta["array"][k] = {"dtype": dt, "max_len": int(max_len[k])}
    np.save(os.path.join(out_dir, "label.npy"), np.asarray(labels, dtype=np.float32).reshape(n, -1))
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    return meta


class ColumnarDataset:
    """Memory-mapped view of a converted directory."""

    def __init__(self, col_dir: str):
        with open(os.path.join(col_dir, "meta.json")) as f:
            self.meta = json.load(f)
        if self.meta.get("version") != FORMAT_VERSION:
            raise ValueError(f"unsupported columnar format version {self.meta.get('version')}")
        mm = lambda *p: np.load(os.path.join(col_dir, *p), mmap_mode="r")
        self.sparse = {k: mm("sparse", k + ".npy") for k in self.meta["sparse"]}
        self.dense = {k: mm("dense", k + ".npy") for k in self.meta["dense"]}
        self.values = {k: mm("array", k + ".values.npy") for k in self.meta["array"]}
        self.offsets = {k: mm("array", k + ".offsets.npy") for k in self.meta["array"]}
        self.max_len = {k: v["max_len"] for k, v in self.meta["array"].items()}
        self.label = mm("label.npy")
        self.n = int(self.meta["n"])

    def __len__(self) -> int:
        return self.n

    @classmethod
    def from_arrays(cls, sparse: Dict[str, np.ndarray], label: np.ndarray, dense: Optional[Dict[str, np.ndarray]] = None,
                    arrays: Optional[Dict[str, tuple]] = None) -> "ColumnarDataset":
        """The same object over in-memory columns (synthetic data, tests): sparse / dense name -> [n]; arrays name -> (values [nnz],
        offsets int64 [n + 1], max_len); label [n, n_labels]."""
        self = cls.__new__(cls)
        self.sparse = dict(sparse)
        self.dense = dict(dense or {})
        self.values = {k: v[0] for k, v in (arrays or {}).items()}
        self.offsets = {k: v[1] for k, v in (arrays or {}).items()}
        self.max_len = {k: int(v[2]) for k, v in (arrays or {}).items()}
        self.label = label
        self.n = int(len(label))
        self.meta = {"version": FORMAT_VERSION, "n": self.n, "sparse": list(self.sparse), "dense": list(self.dense),
                     "array": {k: {"max_len": m} for k, m in self.max_len.items()}}
        return self


class ColumnarLoader:
    """Iterates device batches.  `shuffle` draws a fresh permutation per epoch (numpy Generator seeded
    with seed + epoch); contiguous epochs (shuffle=False) slice the mmaps without a gather."""

    def __init__(self, dataset: ColumnarDataset, batch_size: int, device, shuffle: bool = False,
                 drop_last: bool = False, seed: int = 0, expand_on_device: bool = True, resident: bool = False,
                 csr_bags: bool = False, pinned: bool = False):
        """csr_bags=True (streaming mode, cuda): array features are delivered as they are stored -- `name` = the
        concatenated ids [nnz] and `name_offsets` = int64 [B + 1] -- instead of DataReader's padded ids + mask; the
        models' fused launch reads that form directly (NRX_FEAT_BAG_CSR: same pooled values, bit for bit).
        resident=True: every column is uploaded ONCE and stays in HBM (MIND's training split is a few GB of
        integer columns; one MI355X has 288 GB); a batch is then a device-side row gather -- no host work, no
        PCIe traffic per batch, and a shuffled epoch costs the same as a sequential one.  Same batches, bit
        for bit, as the streaming mode."""
        """pinned=True (streaming mode, cuda): every column is copied ONCE into page-locked host memory; a sequential batch is then a set of
        asynchronous host -> device copies straight from slices of those columns -- no per-batch host copy at all (the mmap form copies every
        batch twice on the host: out of the page cache, then into a pinned staging buffer -- 10 M samples/s, 3 % of the host link; this form
        runs at the link's rate).  A shuffled epoch gathers on the host as before."""
        self.ds, self.B, self.device = dataset, int(batch_size), torch.device(device)
        self.pinned = bool(pinned)
        if self.pinned and (resident or torch.device(device).type != "cuda"):
            raise ValueError("pinned=True is a streaming-mode option for cuda devices")
        self._pin = None
        self.shuffle, self.drop_last, self.seed, self.epoch = shuffle, drop_last, seed, 0
        self.resident = bool(resident)
        if self.resident and self.device.type != "cuda":
            raise ValueError("resident=True keeps the dataset in GPU memory: it needs a cuda device")
        self._res = None
        self.expand_on_device = expand_on_device and self.device.type == "cuda"
        self.csr_bags = bool(csr_bags)
        if self.csr_bags and (self.resident or self.device.type != "cuda"):
            raise ValueError("csr_bags=True is a streaming-mode option for cuda devices")
        self._side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        # ring of grow-only pinned staging buffers: batch-dependent sizes (CSR value counts under
        # shuffle) would otherwise miss torch's pinned-memory cache and pay a hipHostMalloc per batch
        self._ring = [dict() for _ in range(3)]
        self._ring_ev = [None, None, None]
        self._ring_i = 0

    def _stage_pinned(self, slot: dict, key: str, a: np.ndarray) -> torch.Tensor:
        n = a.size
        buf = slot.get(key)
        if buf is None or buf.numel() < n or buf.dtype != torch.from_numpy(a[:0]).dtype:
            cap = max(1024, 1 << (max(n, 1) - 1).bit_length())
            buf = torch.empty(cap, dtype=torch.from_numpy(a[:0]).dtype).pin_memory()
            slot[key] = buf
        view = buf[:n]
        view.numpy()[...] = a.reshape(-1)
        return view.view(a.shape)

    def __len__(self) -> int:
        return self.ds.n // self.B if self.drop_last else (self.ds.n + self.B - 1) // self.B

    # ---- host side: one batch as numpy views/gathers (arrays stay CSR)
    def _host_batch(self, sel):
        ds = self.ds
        take = (lambda a: a[sel]) if isinstance(sel, slice) else (lambda a: a[sel])
        # np.array(...) copies out of the read-only mmap into a fresh writable buffer (pinned next)
        hb = {"sparse": {k: np.array(take(a)) for k, a in ds.sparse.items()},
              "dense": {k: np.array(take(a)) for k, a in ds.dense.items()},
              "label": np.array(take(ds.label)), "array": {}}
        for k in ds.values:
            off = ds.offsets[k]
            if isinstance(sel, slice):
                lo, hi = sel.start, sel.stop
                o = np.asarray(off[lo:hi + 1])
                vals = np.array(ds.values[k][o[0]:o[-1]])
                rel = (o - o[0]).astype(np.int64)
            else:
                starts, ends = np.asarray(off[sel]), np.asarray(off[sel + 1])
                lens = ends - starts
                rel = np.zeros(len(sel) + 1, np.int64)
                np.cumsum(lens, out=rel[1:])
                idx = np.repeat(starts - rel[:-1], lens) + np.arange(rel[-1])
                vals = np.array(ds.values[k][idx])
            hb["array"][k] = (vals, rel)
        return hb

    def _to_device(self, hb) -> Dict[str, torch.Tensor]:
        from ... import ops
        dev = self.device
        if dev.type == "cuda":
            i = self._ring_i
            self._ring_i = (i + 1) % len(self._ring)
            if self._ring_ev[i] is not None:
                self._ring_ev[i].synchronize()          # the H2D copies that last used this slot are done
            slot = self._ring[i]
            keyed = [0]

            def pin(a):
                keyed[0] += 1
                return self._stage_pinned(slot, f"k{keyed[0]}", a)
        else:
            pin = torch.from_numpy
        out: Dict[str, torch.Tensor] = {}
        for k, a in hb["sparse"].items():
            out[k] = pin(a).to(dev, non_blocking=True)
        for k, a in hb["dense"].items():
            out[k] = pin(a).to(dev, non_blocking=True)
        out["label"] = pin(hb["label"]).to(dev, non_blocking=True)
        for k, (vals, rel) in hb["array"].items():
            L = self.ds.max_len[k]
            if self.expand_on_device or self.csr_bags:
                v = pin(vals).to(dev, non_blocking=True) if vals.size else torch.zeros(0, dtype=torch.from_numpy(vals).dtype, device=dev)
                o = pin(rel).to(dev, non_blocking=True)
                if self.csr_bags:
                    out[k], out[f"{k}_offsets"] = v, o
                else:
                    out[k], out[f"{k}_mask"] = ops.csr_to_padded(v, o, L)
            else:
                Bn = len(rel) - 1
                ids = np.zeros((Bn, L), vals.dtype)
                mask = np.zeros((Bn, L), np.float32)
                lens = np.diff(rel)
                col = np.arange(rel[-1]) - np.repeat(rel[:-1], lens)
                row = np.repeat(np.arange(Bn), lens)
                ids[row, col] = vals
                mask[row, col] = 1.0
                out[k] = pin(ids).to(dev, non_blocking=True)
                out[f"{k}_mask"] = pin(mask).to(dev, non_blocking=True)
        if dev.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._ring_ev[i] = ev
        return out

    # ---- device-resident mode
    def _upload(self):
        ds, dev = self.ds, self.device
        up = lambda a: torch.from_numpy(np.array(a)).to(dev)      # np.array: writable copy out of the read-only mmap
        self._res = {"sparse": {k: up(a) for k, a in ds.sparse.items()},
                     "dense": {k: up(a) for k, a in ds.dense.items()},
                     "label": up(ds.label),
                     "values": {k: up(a) for k, a in ds.values.items()},
                     "offsets": {k: up(np.asarray(a, np.int64)) for k, a in ds.offsets.items()}}

    def resident_bytes(self) -> int:
        if self._res is None:
            return 0
        tot = self._res["label"].numel() * self._res["label"].element_size()
        for grp in ("sparse", "dense", "values", "offsets"):
            tot += sum(t.numel() * t.element_size() for t in self._res[grp].values())
        return tot

    def _resident_batch(self, sel: torch.Tensor) -> Dict[str, torch.Tensor]:
        from ... import ops
        r = self._res
        out: Dict[str, torch.Tensor] = {}
        for k, t in r["sparse"].items():
            out[k] = t.index_select(0, sel)
        for k, t in r["dense"].items():
            out[k] = t.index_select(0, sel)
        out["label"] = r["label"].index_select(0, sel)
        for k in r["values"]:
            out[k], out[f"{k}_mask"] = ops.csr_to_padded(r["values"][k], r["offsets"][k], self.ds.max_len[k], rows=sel)
        return out

    def _iter_resident(self) -> Iterator[Dict[str, torch.Tensor]]:
        if self._res is None:
            self._upload()
        n, B, dev = self.ds.n, self.B, self.device
        nb = len(self)
        if self.shuffle:     # the same permutation (and the same in-batch order) as the streaming mode
            perm = torch.from_numpy(np.random.default_rng(self.seed + self.epoch).permutation(n)).to(dev)
        self.epoch += 1
        for i in range(nb):
            lo, hi = i * B, min(n, (i + 1) * B)
            sel = torch.sort(perm[lo:hi]).values if self.shuffle else torch.arange(lo, hi, device=dev)
            yield self._resident_batch(sel)

    # ---- page-locked host columns
    def _pin_columns(self):
        """One-time: the single-valued columns of one dtype are laid out BATCH-BLOCKED in page-locked memory -- [batch][column][B] -- so that a
        sequential batch crosses the host link as ONE contiguous copy (27 copies of 512 KB per C2 batch ran at 36 % of the link: the per-copy
        cost; one 13.6 MB copy runs at its rate), and the device batch's columns are rows of that chunk.  The rows past the last whole batch,
        dense / label columns and array features are pinned as they are."""
        ds, B = self.ds, self.B
        nfull = ds.n // B
        pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
        groups: Dict[str, List[str]] = {}
        for k, a in ds.sparse.items():
            groups.setdefault(str(a.dtype), []).append(k)
        self._blocks = []
        blocked = set()
        for dt, names in groups.items():
            if len(names) < 2 or nfull == 0:
                continue
            blk = torch.empty((nfull, len(names), B), dtype=torch.from_numpy(np.zeros(0, dt)).dtype).pin_memory()
            view = blk.numpy()
            for j, k in enumerate(names):
                view[:, j, :] = np.asarray(ds.sparse[k][:nfull * B]).reshape(nfull, B)
            self._blocks.append((names, blk))
            blocked.update(names)
        tail = slice(nfull * B, ds.n)
        self._pin = {"sparse": {k: pin(a if k not in blocked else a[tail]) for k, a in ds.sparse.items()}, "blocked": blocked, "tail0": nfull * B,
                     "dense": {k: pin(a) for k, a in ds.dense.items()},
                     "label": pin(ds.label), "values": {k: pin(a) for k, a in ds.values.items()},
                     "offsets": {k: pin(np.asarray(a, np.int64)) for k, a in ds.offsets.items()}}

    def _pinned_batch(self, lo: int, hi: int) -> Dict[str, torch.Tensor]:
        """Rows lo .. hi of every column, host -> device, asynchronously, from the page-locked columns themselves (enqueued on the current stream)."""
        from ... import ops
        p, dev, B = self._pin, self.device, self.B
        out: Dict[str, torch.Tensor] = {}
        whole = hi - lo == B and lo % B == 0 and hi <= p["tail0"]
        if whole:
            for names, blk in self._blocks:
                chunk = blk[lo // B].to(dev, non_blocking=True)          # [columns, B]: one copy
                for j, k in enumerate(names):
                    out[k] = chunk[j]
        for k, t in p["sparse"].items():
            if k in p["blocked"]:
                if not whole:
                    out[k] = t[lo - p["tail0"]: hi - p["tail0"]].to(dev, non_blocking=True)
            else:
                out[k] = t[lo:hi].to(dev, non_blocking=True)
        out = {k: out[k] for k in self.ds.sparse}                         # (the streaming loader's key order)
        for k, t in p["dense"].items():
            out[k] = t[lo:hi].to(dev, non_blocking=True)
        out["label"] = p["label"][lo:hi].to(dev, non_blocking=True)
        for k, off in p["offsets"].items():
            o0, o1 = int(off[lo]), int(off[hi])
            v = p["values"][k][o0:o1].to(dev, non_blocking=True)
            o = off[lo:hi + 1].to(dev, non_blocking=True) - o0
            if self.csr_bags:
                out[k], out[f"{k}_offsets"] = v, o
            else:
                out[k], out[f"{k}_mask"] = ops.csr_to_padded(v, o, self.ds.max_len[k])
        return out

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        if self.resident:
            yield from self._iter_resident()
            return
        if self.pinned and not self.shuffle:
            if self._pin is None:
                self._pin_columns()
            n, B = self.ds.n, self.B
            nb = len(self)
            self.epoch += 1
            cur_s = torch.cuda.current_stream(self.device)

            def stage_p(i):
                self._side.wait_stream(cur_s)             # (allocations of this batch's tensors may reuse blocks the consumer has just freed)
                with torch.cuda.stream(self._side):
                    batch = self._pinned_batch(i * B, min(n, (i + 1) * B))
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                return batch, ev

            nxt = stage_p(0) if nb else None
            for i in range(nb):
                batch, ev = nxt
                nxt = stage_p(i + 1) if i + 1 < nb else None      # the next batch crosses the link while this one is consumed
                cur_s.wait_event(ev)
                for t in batch.values():
                    t.record_stream(cur_s)
                yield batch
            return
        n, B = self.ds.n, self.B
        nb = len(self)
        if self.shuffle:
            perm = np.random.default_rng(self.seed + self.epoch).permutation(n)
        self.epoch += 1
        sels = []
        for i in range(nb):
            lo, hi = i * B, min(n, (i + 1) * B)
            sels.append(np.sort(perm[lo:hi]) if self.shuffle else slice(lo, hi))

        def stage(sel):
            hb = self._host_batch(sel)
            if self._side is None:
                return self._to_device(hb), None
            with torch.cuda.stream(self._side):
                batch = self._to_device(hb)
                ev = torch.cuda.Event()
                ev.record(self._side)
            return batch, ev

        nxt = stage(sels[0]) if nb else None
        for i in range(nb):
            cur = nxt
            nxt = stage(sels[i + 1]) if i + 1 < nb else None      # next batch's H2D overlaps this batch's compute
            batch, ev = cur
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in batch.values():
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield batch
