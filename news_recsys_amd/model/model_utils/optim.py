"""Optimizer for the row-sparse gradient mode (`embeddings.sparse_grad: true`).

The reference trains everything with one dense `AdamW(self.parameters())` (sort/deep/model.py:55): on a
100M-row table that is a full-table read-modify-write of weights and both moments every step.  With
row-sparse table grads the tables are updated by `torch.optim.SparseAdam` (moments touched only for the
looked-up rows; no weight decay) and the dense parameters keep AdamW.  This wrapper presents both as
one `Optimizer` so `configure_optimizers()` keeps its reference shape (one optimizer + one scheduler)."""
import ctypes as C
import math

import torch

from ... import _lib, ops


def dense_adamw(params, **kw):
    """torch.optim.AdamW as the reference builds it (sort/deep/model.py:55, recall/DSSM/model.py) -- with torch's ONE-PASS multi-tensor kernel
    (`fused=True`) when every parameter lives on the GPU: same update rule, one read and one write of (p, grad, m, v) instead of the ~10
    elementwise passes of the default foreach form.  On the 26 x 100 k-row tables of a C2-shaped model the step (embedding forward, backward,
    optimizer) goes 1110 -> 396 us at B = 512 (tools/probe_small_train_modes.py); NRX_ADAMW_FUSED=0 keeps torch's default."""
    import os
    params = list(params)
    if os.environ.get("NRX_ADAMW_FUSED", "1") != "0" and params and all(torch.is_tensor(p) and p.is_cuda and p.is_floating_point() for p in params):
        try:
            return torch.optim.AdamW(params, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.AdamW(params, **kw)


class FusedSparseAdam:
    """Adam(W) for the embedding tables, fused with the row-sparse backward.  The backward
    leaves (unique (table,row) keys, summed row gradients, counts) on the device in an ops.SparseGradSink; step()
    updates exactly those rows of weights and moments with one `nrx_sparse_adam_step` launch per group -- no
    COO tensors, no host synchronisation, no traffic proportional to the table size.  Update rule =
    torch.optim.SparseAdam (tested against it) + optional decoupled weight decay on the touched rows.
    A table that received gradients from several backward groups in one step (DSSM's towers share the news
    table) gets them merged first, so the step is still ONE Adam update per row.  Tables are identified by tensor
    identity; their moments are created (zeros) the first time a table shows up in the sink."""

    def __init__(self, sink: "ops.SparseGradSink", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False,
                 params=None):
        """capturable=True keeps the step counter and the bias-corrected step size on the device (like
        torch.optim.Adam(capturable=True)) so step() can be captured in a HIP graph (graph.GraphedStep); lr is
        then fixed at capture time.  params (optional): the table tensors in a stable order -- state_dict() then
        keys the moments by position in that list, so a checkpoint restores into a freshly built model."""
        self.sink, self.lr, self.betas, self.eps, self.weight_decay = sink, lr, betas, eps, weight_decay
        self.capturable = bool(capturable)
        self.params = list(params) if params is not None else None
        self._t_dev = None
        self.t = 0
        self.tables = []         # every table seen so far; position = the table's index in the optimizer's key space
        self._index = {}         # id(tensor) -> position
        self.moments = []        # (exp_avg, exp_avg_sq) per table
        self._gmaps = {}         # table-list identity -> device map (launch-local table index -> position)
        self._identity = {}      # table-list identity -> that map is the identity
        self._maps = []          # per-table slot maps of the two-list merge (made on first use)
        self.pair_merge = True   # two backward groups of one width: merge by marking (False: the sort-based _merge)

    def _register(self, t: torch.Tensor) -> int:
        i = self._index.get(id(t))
        if i is None:
            i = len(self.tables)
            self._index[id(t)] = i
            self.tables.append(t)
            # both moments of a row side by side ([rows, 2, D]; exp_avg / exp_avg_sq are its two views): the update is a random
            # read-modify-write of (w, m, v) and every 64-byte access costs a 128-byte fetch -- adjacent, m and v share one
            mv = torch.zeros((t.shape[0], 2, t.shape[1]), dtype=t.dtype, device=t.device)
            self.moments.append((mv[:, 0], mv[:, 1]))
        return i

    def _global_keys(self, e):
        """Re-express an entry's keys (table index = position in that launch's table list) in the optimizer's own
        table numbering; filler past the device-side count becomes INT64_MAX (ignored by the kernel)."""
        BIG = torch.iinfo(torch.int64).max
        MASK = (1 << 40) - 1
        dev = e["uniq"].device
        ck = tuple(id(t) for t in e["tables"])
        gmap = self._gmaps.get(ck)
        if gmap is None:        # built once per table list (a host-to-device copy: must not happen inside a graph capture)
            pos = [self._register(t) for t in e["tables"]]
            gmap = torch.tensor(pos, dtype=torch.int64, device=dev)
            self._gmaps[ck] = gmap
            self._identity[ck] = pos == list(range(len(pos)))
        k = e["uniq"]
        if e.get("filler"):             # the one-launch small form: unused slots are keyed -1 wherever they are
            if self._identity.get(ck):
                return k                # launch-local table numbers ARE the optimizer's: the kernel skips negative keys itself
            valid = k >= 0
        else:
            valid = torch.arange(e["cap"], device=dev) < e["counts"][0]
        local = torch.where(valid, k >> 40, torch.zeros_like(k))
        return torch.where(valid, (gmap[local] << 40) | (k & MASK), torch.full_like(k, BIG))

    @staticmethod
    def _merge(keys, vals):
        """(keys, values) lists with possibly repeated (table,row) keys -> one entry per key.  Device-only torch
        ops; INT64_MAX filler sorts last and stays filler."""
        BIG = torch.iinfo(torch.int64).max
        skeys, order = torch.sort(keys, stable=True)
        head = torch.ones_like(skeys, dtype=torch.bool)
        head[1:] = skeys[1:] != skeys[:-1]
        seg = torch.cumsum(head, 0) - 1
        merged_vals = torch.zeros_like(vals).index_add_(0, seg, vals[order])
        merged_keys = torch.full_like(skeys, BIG)
        merged_keys[seg] = skeys
        return merged_keys, merged_vals

    @torch.no_grad()
    def step(self):
        if not self.sink.pending:
            return
        lib = _lib.load()
        self.t += 1
        b1, b2 = self.betas
        step_size = self.lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        ss_dev = None
        if self.capturable:
            dev = self.sink.pending[0]["uniq"].device
            if self._t_dev is None:
                self._t_dev = torch.zeros((), dtype=torch.float64, device=dev)
            self._t_dev += 1
            ss_dev = (self.lr * torch.sqrt(1.0 - b2 ** self._t_dev) / (1.0 - b1 ** self._t_dev)).to(torch.float32).reshape(1)
        by_dim = {}
        for e in self.sink.pending:
            by_dim.setdefault(e["dim"], []).append((self._global_keys(e), e["values"]))
        n = len(self.tables)
        if n > _lib.NRX_MAX_FEATURES:
            raise NotImplementedError("FusedSparseAdam: more than 64 distinct tables")
        tp = (C.c_void_p * n)(*[t.data_ptr() for t in self.tables])
        mp = (C.c_void_p * n)(*[m.data_ptr() for m, _ in self.moments])
        vp = (C.c_void_p * n)(*[v.data_ptr() for _, v in self.moments])
        def adam(keys, vals):
            ops.check(lib.nrx_sparse_adam_step(tp, mp, vp, n, dim, keys.data_ptr(), vals.data_ptr(), keys.numel(), None,
                                               step_size, ss_dev.data_ptr() if ss_dev is not None else None, b1, b2, self.eps,
                                               self.lr * self.weight_decay,
                                               torch.cuda.current_stream(keys.device).cuda_stream), "nrx_sparse_adam_step")

        for dim, lst in by_dim.items():
            if len(lst) == 1:
                adam(*lst[0])
            elif len(lst) == 2 and self.pair_merge:
                # one table fed by two backward groups (DSSM's towers share the news table): ONE update per row.  List A is marked in per-table
                # slot maps, the pairs of B that A also holds are added into A's rows and blanked (nrx_rows_merge), A is unmarked; the two lists
                # are then disjoint: three small launches instead of a device sort + segment sums over the concatenation
                (ka, va), (kb, vb) = lst
                stream = torch.cuda.current_stream(ka.device).cuda_stream
                maps = self._slot_maps()
                rows = (C.c_int64 * n)(*[t.shape[0] for t in self.tables])
                ops.check(lib.nrx_rows_mark(ka.data_ptr(), ka.numel(), None, maps, rows, n, 0, stream), "nrx_rows_mark")
                ops.check(lib.nrx_rows_merge(kb.data_ptr(), vb.data_ptr(), kb.numel(), None, va.data_ptr(), maps, rows, n, dim, stream), "nrx_rows_merge")
                ops.check(lib.nrx_rows_mark(ka.data_ptr(), ka.numel(), None, maps, rows, n, 1, stream), "nrx_rows_mark")
                adam(ka, va)
                adam(kb, vb)
            else:
                adam(*self._merge(torch.cat([k for k, _ in lst]), torch.cat([v for _, v in lst])))
        self.sink.clear()

    def _slot_maps(self):
        """int32 [rows] per table, all -1 between uses (nrx_rows_mark / nrx_rows_merge): made when a step first needs them."""
        while len(self._maps) < len(self.tables):
            t = self.tables[len(self._maps)]
            self._maps.append(torch.full((t.shape[0],), -1, dtype=torch.int32, device=t.device))
        return (C.c_void_p * len(self._maps))(*[m.data_ptr() for m in self._maps])

    def zero_grad(self, set_to_none: bool = True):
        self.sink.clear()

    # ---- checkpointing: step count + both moments of every table that has been updated so far
    def _stable_index(self, t):
        if self.params is not None:
            for i, p in enumerate(self.params):
                if p is t:
                    return i
        return None

    def state_dict(self):
        tables = {}
        for pos, t in enumerate(self.tables):
            key = self._stable_index(t)
            m, v = self.moments[pos]
            tables[key if key is not None else f"unlisted:{pos}"] = {"exp_avg": m, "exp_avg_sq": v}
        steps = {}
        for pos, c in getattr(self, "_steps", {}).items():          # (ExactDenseAdamW: a table's own step count)
            key = self._stable_index(self.tables[pos])
            steps[key if key is not None else f"unlisted:{pos}"] = c
        return {"t": self.t, "t_dev": None if self._t_dev is None else float(self._t_dev.item()), "tables": tables, "steps": steps}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        self._t_dev = None
        if sd.get("t_dev") is not None and self.capturable and self.params:
            self._t_dev = torch.tensor(sd["t_dev"], dtype=torch.float64, device=self.params[0].device)
        for key, mv in sd["tables"].items():
            if isinstance(key, str):
                raise ValueError("FusedSparseAdam.load_state_dict: the checkpoint holds moments of a table that was not in "
                                 "`params` when it was saved; construct the optimizer with params=<the table list>")
            if self.params is None or not 0 <= key < len(self.params):
                raise ValueError("FusedSparseAdam.load_state_dict needs params=<the same table list as at save time>")
            pos = self._register(self.params[key])
            m, v = self.moments[pos]
            m.copy_(mv["exp_avg"])
            v.copy_(mv["exp_avg_sq"])
        unlisted = [key for key in (sd.get("steps") or {}) if isinstance(key, str)]
        if unlisted:
            raise ValueError("load_state_dict: the checkpoint holds step counts of tables that were not in `params` when it was saved "
                             f"({unlisted}); construct the optimizer with params=<the table list>")
        if sd.get("steps"):
            self._steps = {self._register(self.params[key]): int(c) for key, c in sd["steps"].items()}
        elif hasattr(self, "maps"):
            # (ExactDenseAdamW) a checkpoint written before the per-table step counts existed: every table had moved on every one of the `t`
            # steps -- restarting the tables at step 1 would apply lr / (1 - beta1) to the restored moments and leave torch.optim.AdamW's path
            self._steps = {pos: self.t for pos in range(len(self.tables))}


class ExactDenseAdamW(FusedSparseAdam):
    """The reference's optimizer for the tables -- ONE dense torch.optim.AdamW over model.parameters() (sort/deep/model.py:54-65: every row of
    every table moves every step: decoupled weight decay, decaying moments) -- fed from the row-sparse sink instead of dense .grad tensors:
    `nrx_rows_mark` notes which rows have a gradient this step, `nrx_dense_adamw_rows` streams over every row of every table once (SURVEY 8f
    row 2, "exact-dense mode").  Same numbers as torch.optim.AdamW on the dense gradients (tests/test_fused_sparse_adam_gpu.py); no dense
    gradient is formed, zero-filled or read.  As torch.optim.AdamW skips a parameter whose .grad is None, a step() updates the tables that a
    backward launch of the step looked up (the sink entries name them) -- all of their rows -- and leaves the others alone, each table with
    its own step count (capturable=True: one device-side count for all tables, every registered table is streamed every step).
    exp_avg / exp_avg_sq are plain [rows, dim] tensors (torch's layout)."""

    def __init__(self, sink, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        super().__init__(sink, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable, params=list(params))
        self.maps = []
        for t in self.params:
            self._register(t)

    def _register(self, t: torch.Tensor) -> int:
        i = self._index.get(id(t))
        if i is None:
            i = len(self.tables)
            self._index[id(t)] = i
            self.tables.append(t)
            self.moments.append((torch.zeros_like(t), torch.zeros_like(t)))
            self.maps.append(torch.full((t.shape[0],), -1, dtype=torch.int32, device=t.device))
        return i

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        self.t += 1
        by_dim = {}
        touched = set()
        for e in self.sink.pending:
            by_dim.setdefault(e["dim"], []).append((self._global_keys(e), e["values"]))
            for tid in e.get("table_ids", range(len(e["tables"]))):      # the tables this backward launch looked up
                i = self._index.get(id(e["tables"][tid]))
                if i is not None:
                    touched.add(i)
        if self.capturable:
            touched = set(range(len(self.tables)))       # (one device-side step count: every table moves every step)
        steps = getattr(self, "_steps", None)
        if steps is None:
            steps = self._steps = {}
        for i in touched:
            steps[i] = steps.get(i, 0) + 1
        n = len(self.tables)
        if n > _lib.NRX_MAX_FEATURES:
            raise NotImplementedError("ExactDenseAdamW: more than 64 distinct tables")
        b1, b2 = self.betas
        hyper = None
        if self.capturable:             # the step count lives on the device: a captured loop advances the bias corrections between replays
            dev0 = self.tables[0].device
            if self._t_dev is None:
                self._t_dev = torch.zeros((), dtype=torch.float64, device=dev0)
            self._t_dev += 1
            hyper = torch.stack([self.lr / (1.0 - b1 ** self._t_dev), 1.0 / torch.sqrt(1.0 - b2 ** self._t_dev)]).to(torch.float32)
        elif torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ExactDenseAdamW: construct with capturable=True to capture step() in a graph (the step count is baked in otherwise)")
        marked = {}
        for dim, tstep in sorted({(self.tables[i].shape[1], steps[i]) for i in touched}):
            idx = [i for i in sorted(touched) if self.tables[i].shape[1] == dim and steps[i] == tstep]
            dev = self.tables[idx[0]].device
            stream = torch.cuda.current_stream(dev).cuda_stream
            k = len(idx)
            rows = (C.c_int64 * n)(*[t.shape[0] for t in self.tables])
            maps_all = (C.c_void_p * n)(*[m.data_ptr() for m in self.maps])
            if dim not in marked:                    # a dim's rows are marked once; tables of the dim at another step count read the same marks
                vals = None
                lst = by_dim.get(dim)
                if lst:
                    if len(lst) == 1:
                        keys, vals = lst[0]
                    else:       # one table fed by several backward groups: ONE gradient per row
                        keys, vals = self._merge(torch.cat([kk for kk, _ in lst]), torch.cat([v for _, v in lst]))
                    ops.check(lib.nrx_rows_mark(keys.data_ptr(), keys.numel(), None, maps_all, rows, n, 0, stream), "nrx_rows_mark")
                marked[dim] = vals
            vals = marked[dim]
            tp = (C.c_void_p * k)(*[self.tables[i].data_ptr() for i in idx])
            mp = (C.c_void_p * k)(*[self.moments[i][0].data_ptr() for i in idx])
            vp = (C.c_void_p * k)(*[self.moments[i][1].data_ptr() for i in idx])
            sp = (C.c_void_p * k)(*[self.maps[i].data_ptr() for i in idx])
            rk = (C.c_int64 * k)(*[self.tables[i].shape[0] for i in idx])
            ops.check(lib.nrx_dense_adamw_rows(tp, mp, vp, sp, rk, k, dim, vals.data_ptr() if vals is not None else None, tstep, self.lr, b1, b2,
                                               self.eps, self.weight_decay, hyper.data_ptr() if hyper is not None else None, stream),
                      "nrx_dense_adamw_rows")
        self.sink.clear()


class SparseDenseAdam(torch.optim.Optimizer):
    def __init__(self, sparse_params, dense_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, fused_sink=None,
                 capturable=False, exact=False):
        """fused_sink: an ops.SparseGradSink -> the tables are updated by FusedSparseAdam from the sink instead of
        torch.optim.SparseAdam from COO .grad tensors.  exact (with fused_sink): by ExactDenseAdamW -- the reference's dense AdamW over every
        row, weight decay included, fed from the sink."""
        sparse_params, dense_params = list(sparse_params), list(dense_params)
        groups = [{"params": sparse_params, "sparse": True}]
        if dense_params:
            groups.append({"params": dense_params, "sparse": False})
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if exact and fused_sink is None:
            raise ValueError("SparseDenseAdam(exact=True) needs fused_sink")
        self._sparse = (ExactDenseAdamW(fused_sink, sparse_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable) if exact else
                        FusedSparseAdam(fused_sink, lr=lr, betas=betas, eps=eps, capturable=capturable, params=sparse_params)
                        if fused_sink is not None
                        else torch.optim.SparseAdam(sparse_params, lr=lr, betas=betas, eps=eps))
        self._dense = (torch.optim.AdamW(dense_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable)
                       if dense_params else None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():         # the closure runs forward + backward
                loss = closure()
        for g in self.param_groups:           # a scheduler edits self.param_groups: forward the lr
            if g["sparse"] and isinstance(self._sparse, FusedSparseAdam):
                self._sparse.lr = g["lr"]
                continue
            for inner in ((self._sparse,) if g["sparse"] else ((self._dense,) if self._dense else ())):
                for ig in inner.param_groups:
                    ig["lr"] = g["lr"]
        self._sparse.step()
        if self._dense is not None:
            self._dense.step()
        return loss

    def zero_grad(self, set_to_none: bool = True):
        self._sparse.zero_grad(set_to_none)
        if self._dense is not None:
            self._dense.zero_grad(set_to_none)

    # ---- checkpointing (Lightning saves optimizer.state_dict()): all Adam state lives in the inner optimizers
    def state_dict(self):
        return {"sparse_dense_adam": 1,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "sparse": self._sparse.state_dict(),
                "dense": self._dense.state_dict() if self._dense is not None else None}

    def load_state_dict(self, sd):
        if "sparse_dense_adam" not in sd:
            raise ValueError("not a SparseDenseAdam state dict")
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        self._sparse.load_state_dict(sd["sparse"])
        if self._dense is not None and sd.get("dense") is not None:
            self._dense.load_state_dict(sd["dense"])
