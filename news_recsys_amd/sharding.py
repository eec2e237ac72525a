"""Row-sharded embedding tables over the GPUs of one node (new in this build; the reference is
single-device -- every trainer is `devices=1`, e.g. src/model/sort/deep/train.py:38-44).

Partitioning: data-parallel batch (every rank owns B impressions, dense params are
replicated), tables row-wise round-robin: global row r lives on rank r % G at local row r // G.
The padding row 0 is rank 0's local row 0.

One forward, per group of features that share an embedding dim (one exchange each):
  1. flat id list of the group (feature-major; bag features contribute B*L ids)
  2. stable bucketing by owner (HIP, deterministic)  -> send buffer of local rows + slot[] un-permute map
  3. per-(owner, feature) counts  -> all_to_all_single  (tells each owner how its inbox is segmented)
  4. local rows                   -> all_to_all_single  (RCCL over xGMI; variable splits)
  5. owner-side segmented gather from the local shards (HIP)
  6. rows back                    -> all_to_all_single
  7. ONE fused launch over the returned rows addressed by slot[]: bag pooling + concat
     (+ wide split / FM epilogue) -- the same kernel as the single-GPU path, with the returned-row
     buffer as its "table".
Backward mirrors it: slot-scatter of the upstream grad (a permutation, so collision-free), one
all_to_all_single back to the owners, segmented scatter-add into the local dense grads (row 0 of
rank 0 -- the global padding row -- excluded).  Dense parameters are all-reduced by
`allreduce_dense_grads`.

Two exchange modes.  "capacity" (default) is host-sync free: every (source, owner) pair gets a fixed
block of `cap = ceil(N/G * (1 + slack))` slots, so all three all-to-alls use EQUAL splits, the owner
learns its inbox segmentation from a device-resident counts matrix, and nothing is read back to the
host inside a step (`nrx_route_ids` / `nrx_gather_inbox`).  If a block overflows (heavily skewed
ids) the step is redone in "exact" mode: variable splits sized from counts read back to the host.

The local kernels come from a backend object.  The product backend is `HipBackend` (the C-ABI);
tests inject a CPU checker backend so that the routing + collectives can run under gloo without a
GPU -- no CPU implementation ships in this package.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import ctypes

import torch
import torch.distributed as dist

from . import ops
from ._lib import (NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_DENSE, NRX_FEAT_BAG_CSR, NRX_FEAT_ROW0_IS_DATA, NRX_SPARSE)


# --------------------------------------------------------------------------------- partition helpers
def local_row_count(rows: int, rank: int, world: int) -> int:
    """Number of global rows r in [0, rows) with r % world == rank."""
    return (rows - rank + world - 1) // world if rows > rank else 0


def shard_table(full: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows rank, rank+world, ... of a full [rows, D] table (scatter-on-load)."""
    return full[rank::world].contiguous()


def unshard_tables(shards: Sequence[torch.Tensor]) -> torch.Tensor:
    """Inverse of shard_table over all ranks (gather-on-save): shards[r] holds rows r::world."""
    world = len(shards)
    rows = sum(s.shape[0] for s in shards)
    full = shards[0].new_empty((rows, shards[0].shape[1]))
    for r, s in enumerate(shards):
        full[r::world] = s
    return full


# --------------------------------------------------------------------------------- local backend
class HipBackend:
    """The product backend: every local step is a HIP kernel behind the C-ABI."""

    def bucketize(self, ids: torch.Tensor, world: int):
        return ops.bucketize_by_owner(ids, world)

    def gather_segmented(self, tables, seg_start, seg_table, local_rows, n_rows):
        """Returns (rows, status | None).  status is the device int32[4] out-of-range record; it is NOT
        checked here: raising on one rank before the return all-to-all would strand its peers."""
        return ops.gather_rows_segmented(tables, seg_start, seg_table, local_rows, n_rows, defer_check=True)

    def scatter_add_segmented(self, grad_tables, seg_start, seg_table, local_rows, g_rows, skip_row0):
        ops.scatter_add_rows_segmented(grad_tables, seg_start, seg_table, local_rows, g_rows, skip_row0)

    def embed(self, plan, tables, inputs, weights, out_ld=None, need_out=True):
        return ops.embed_apply(plan, tables, inputs, weights, out_ld=out_ld, need_out=need_out)

    # fixed-capacity (sync-free) exchange
    def route(self, id_tensors, world, cap):
        return ops.route_ids(id_tensors, world, cap)

    def route_dedup(self, id_tensors, table_of, table_local_rows, world, cap):
        return ops.route_ids_dedup(id_tensors, table_of, table_local_rows, world, cap)

    def gather_inbox(self, tables, feat_table, world, cap, recv2d, inbox, want_status):
        status = torch.zeros(4, dtype=torch.int32, device=inbox.device) if want_status else None
        return ops.gather_inbox(tables, feat_table, world, cap, recv2d, inbox, status), status

    def scatter_add_inbox(self, grad_tables, feat_table, world, cap, recv2d, inbox, g_rows, skip_row0):
        ops.scatter_add_inbox(grad_tables, feat_table, world, cap, recv2d, inbox, g_rows, skip_row0)

    # pooled-bag channel (owner-side partial pooling)
    def bag_norm_weights(self, mask, batch, bag_len, kind, device):
        return ops.bag_norm_weights(mask, batch, bag_len, kind, device)

    def route_bags(self, id_tensors, weights, world, cap):
        return ops.route_bags(id_tensors, weights, world, cap)

    def pool_inbox(self, tables, feat_table, batch, world, cap, recv2d, inbox_rows, inbox_tag, inbox_w, want_status):
        status = torch.zeros(4, dtype=torch.int32, device=inbox_rows.device) if want_status else None
        return ops.pool_inbox(tables, feat_table, batch, world, cap, recv2d, inbox_rows, inbox_tag, inbox_w, status), status

    def pool_inbox_bwd(self, grad_tables, feat_table, batch, world, cap, recv2d, inbox_rows, inbox_tag, inbox_w, g_partial, skip_row0):
        ops.pool_inbox_bwd(grad_tables, feat_table, batch, world, cap, recv2d, inbox_rows, inbox_tag, inbox_w, g_partial, skip_row0)

    def index_checks_on(self) -> bool:
        return ops._INDEX_CHECK != "off"


# --------------------------------------------------------------------------------- feature description
class PeerMappingError(RuntimeError):
    """Raised on EVERY rank when some rank could not map its peers' buffers (PreparedShardedForward._map_peer_buffers)."""


@dataclass
class ShardedFeature:
    name: str
    kind: int            # NRX_SPARSE / NRX_DENSE / NRX_BAG_*
    table: str           # table name ('' for dense)
    dim: int
    bag_len: int = 0
    wide: bool = False
    fm: bool = False
    replicated: bool = False   # planner: this feature's table is held in full on every rank (no exchange)


@dataclass
class _Route:
    """What one exchange leaves behind for the backward."""
    feats: List[int]                # indices into the feature list, in group order
    table_names: List[str]
    dim: int
    n_send: int
    send_counts: List[int]          # ids sent to each rank
    recv_counts: List[int]          # ids received from each rank
    seg_start: torch.Tensor         # [world*F + 1] segment starts of the inbox (src-major, feature-minor)
    seg_table: torch.Tensor         # [world*F] table index of each segment
    recv_rows: torch.Tensor         # [n_recv] local rows asked of this rank
    slot: torch.Tensor              # [n_send] position of each source id in the send buffer
    feat_off: List[int]             # start of each feature inside the flat source list
    cap: int = 0                    # > 0: fixed-capacity route (recv2d / feat_table used instead of seg_*)
    recv2d: Optional[torch.Tensor] = None
    feat_table: Optional[List[int]] = None
    pooled: bool = False            # pooled-bag channel: the owner returned partial sums, not rows
    inbox_tag: Optional[torch.Tensor] = None
    inbox_w: Optional[torch.Tensor] = None
    batch: int = 0


class RowShardedEmbedding:
    """Exchange engine for one process.  `tables` maps table name -> LOCAL shard tensor
    [local_row_count(rows), dim] (a leaf requiring grad when training)."""

    def __init__(self, rank: int, world: int, group=None, backend=None, mode: str = "capacity",
                 slack: float = 0.05, overflow_policy: str = "check", host_staged: bool = False, pool_bags: bool = True,
                 dedup: bool = False, grad_average: bool = True):
        """mode: "capacity" (sync-free, default) or "exact".  overflow_policy (capacity mode):
        "check" = agree on overflow across ranks after each forward (one small all-reduce + host read)
        and transparently redo the step in exact mode; "defer" = never read back inside the step --
        the caller polls `overflowed()` (bench / inference pipelines)."""
        if mode not in ("capacity", "exact"):
            raise ValueError("mode must be 'capacity' or 'exact'")
        self.rank, self.world, self.group = rank, world, group
        # host_staged: collectives bounce device buffers through host memory (a gloo group works then).  A TEST transport:
        # it lets several ranks share one GPU, so the HIP routing / owner kernels can be run with world > 1 on a
        # single-GPU box (RCCL refuses two ranks on one device); the product transport is RCCL on device buffers.
        self.host_staged = bool(host_staged)
        self.backend = backend if backend is not None else HipBackend()
        self.mode, self.slack, self.overflow_policy = mode, slack, overflow_policy
        # pool_bags: row-sharded bag features are pooled AT THE OWNER (one partial vector per (sample, owner) comes back
        # instead of L rows per sample -- SURVEY 8e step 2).  Off: bags travel as rows like single-valued features.
        self.pool_bags = bool(pool_bags) and mode == "capacity"
        # dedup: every distinct (owner, table, row) of an exchange travels once (nrx_route_ids_dedup; a sort per step: pays
        # on skewed click-log ids, costs on uniform ones -- off by default)
        self.dedup = bool(dedup) and mode == "capacity"
        # grad_average: every rank's loss is a mean over ITS batch; a row-sharded table receives the SUM of all ranks'
        # gradient rows.  True (default) scales them by 1 / world so that tables see the gradient of the global-batch mean
        # -- the same convention as allreduce_dense_grads for the replicated dense parameters (and what a single-GPU run
        # over the concatenated batch computes).  False: the plain sum.
        self.grad_average = bool(grad_average)
        self._overflow_marks: List[Tuple[torch.Tensor, int]] = []

    def plan_groups(self, feats: Sequence["ShardedFeature"]):
        """Exchange groups: features that share an embedding dim travel together (<= 64 per exchange); row-sharded bag
        features form their own groups when they are pooled at the owner.  Returns (groups, indices of the pooled ones)."""
        by_key: Dict[tuple, List[int]] = {}
        for i, f in enumerate(feats):
            if f.kind != NRX_DENSE and not f.replicated:
                pooled = self.pool_bags and f.kind in (NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM) and not f.wide
                by_key.setdefault((f.dim, pooled), []).append(i)
        groups: List[List[int]] = []
        pooled_set = set()
        for key in sorted(by_key):
            idxs = by_key[key]
            for k in range(0, len(idxs), 64):
                if key[1]:
                    pooled_set.add(len(groups))
                groups.append(idxs[k:k + 64])
        return groups, pooled_set

    def capacity_for(self, n_total: int) -> int:
        if self.world == 1:
            return max(64, n_total)
        cap = int(n_total / self.world * (1.0 + self.slack)) + 256
        return (cap + 63) // 64 * 64

    def overflowed(self) -> bool:
        """True if any fixed-capacity exchange since the last call exceeded its block capacity."""
        bad = False
        for t, cap in self._overflow_marks:
            bad |= int(t.item()) > cap
        self._overflow_marks.clear()
        return bad

    # ---- collectives (RCCL when the tensors are on GPUs: backend "nccl" is RCCL on ROCm)
    def _a2a(self, out: torch.Tensor, inp: torch.Tensor, out_split=None, in_split=None) -> torch.Tensor:
        if self.world == 1:
            out.copy_(inp)
        elif self.host_staged and out.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_split, in_split, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_split, in_split, group=self.group)
        return out

    def _all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            if self.host_staged and t.is_cuda:
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
                t.copy_(c)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def _exchange(self, feats: Sequence[ShardedFeature], idxs: List[int], inputs, tables: Dict[str, torch.Tensor]):
        """Steps 1-6 for one dim-group.  Returns (returned rows [n_send, D], _Route)."""
        W = self.world
        dev = inputs[idxs[0]].device
        flat, feat_off, off = [], [], 0
        for i in idxs:
            x = inputs[i].reshape(-1)
            x = x if x.dtype == torch.int64 else x.long()
            flat.append(x)
            feat_off.append(off)
            off += x.numel()
        ids = torch.cat(flat) if len(flat) > 1 else flat[0].contiguous()
        n_send = ids.numel()
        F_ = len(idxs)
        table_names: List[str] = []
        for i in idxs:
            if feats[i].table not in table_names:
                table_names.append(feats[i].table)
        D = feats[idxs[0]].dim

        counts, send_rows, slot = self.backend.bucketize(ids, W)                       # step 2
        # step 3: per-(owner, feature) counts.  feature id of every source position:
        lens = torch.tensor([flat[k].numel() for k in range(F_)], device=dev)
        fid = torch.repeat_interleave(torch.arange(F_, device=dev), lens)
        owner = torch.where(ids < 0, torch.zeros_like(ids), torch.remainder(ids, W))   # same rule as the HIP bucketing
        counts2d = torch.bincount(owner * F_ + fid, minlength=W * F_).view(W, F_)
        recv2d = torch.empty_like(counts2d)
        self._a2a(recv2d.view(-1), counts2d.contiguous().view(-1))                    # equal splits: F_ each
        host = torch.stack([counts2d.sum(1), recv2d.sum(1)]).cpu()                     # the one host sync per group
        send_counts, recv_counts = host[0].tolist(), host[1].tolist()
        n_recv = int(sum(recv_counts))

        recv_rows = torch.empty(n_recv, dtype=torch.int64, device=dev)                 # step 4
        self._a2a(recv_rows, send_rows, recv_counts, send_counts)

        seg_len = recv2d.reshape(-1)                                                   # src-major, feature-minor
        seg_start = torch.zeros(W * F_ + 1, dtype=torch.int64, device=dev)
        seg_start[1:] = torch.cumsum(seg_len, 0)
        tix = torch.tensor([table_names.index(feats[i].table) for i in idxs], dtype=torch.int32, device=dev)
        seg_table = tix.repeat(W)
        local_tables = [tables[t] for t in table_names]
        rows_out, status = self.backend.gather_segmented(local_tables, seg_start, seg_table, recv_rows, n_recv)   # step 5

        ret = torch.empty((n_send, D), dtype=torch.float32, device=dev)                # step 6
        self._a2a(ret.view(-1), rows_out.view(-1), [c * D for c in send_counts], [c * D for c in recv_counts])
        if status is not None:
            # every rank learns about an out-of-range id on ANY owner and raises together (reference:
            # IndexError from nn.Embedding on CPU), instead of one rank leaving the collective sequence
            bad = status[:1].to(torch.int64)
            self._all_reduce_max(bad)
            if int(bad.item()) != 0:
                raise IndexError("index out of range in self: a routed lookup exceeded its table on some rank "
                                 f"(this rank's record: {status.tolist()})")
        route = _Route(list(idxs), table_names, D, n_send, send_counts, recv_counts, seg_start, seg_table,
                       recv_rows, slot, feat_off)
        return ret, route

    def _exchange_capacity(self, feats: Sequence[ShardedFeature], idxs: List[int], inputs, tables: Dict[str, torch.Tensor]):
        """Sync-free steps 1-6 for one group (<= 64 features of one dim).  Returns (rows, _Route, overflow)."""
        W = self.world
        ids = []
        dt = torch.int32 if all(inputs[i].dtype == torch.int32 for i in idxs) else torch.int64
        feat_off, off = [], 0
        for i in idxs:
            x = inputs[i]
            ids.append(x if x.dtype == dt else x.to(dt))
            feat_off.append(off)
            off += x.numel()
        n_send = off
        cap = self.capacity_for(n_send)
        table_names: List[str] = []
        for i in idxs:
            if feats[i].table not in table_names:
                table_names.append(feats[i].table)
        feat_table = [table_names.index(feats[i].table) for i in idxs]
        D = feats[idxs[0]].dim
        dev = inputs[idxs[0]].device

        if self.dedup:            # segments of an owner's block = tables (unique rows, ordered by table then row)
            lrows = [tables[t].shape[0] + 1 for t in table_names]
            send_rows, slot, counts2d, overflow = self.backend.route_dedup(ids, feat_table, lrows, W, cap)
            feat_table = list(range(len(table_names)))
        else:
            send_rows, slot, counts2d, overflow = self.backend.route(ids, W, cap)        # steps 1-3
        recv2d = torch.empty_like(counts2d)
        self._a2a(recv2d.view(-1), counts2d.view(-1))                                   # equal splits
        inbox = torch.empty(W * cap, dtype=send_rows.dtype, device=dev)
        self._a2a(inbox, send_rows)                                                     # step 4, equal splits (int32 local rows)
        want_status = self.backend.index_checks_on() and self.overflow_policy == "check"
        rows_out, status = self.backend.gather_inbox([tables[t] for t in table_names], feat_table, W, cap,
                                                     recv2d, inbox, want_status)        # step 5
        ret = torch.empty((W * cap, D), dtype=torch.float32, device=dev)
        self._a2a(ret.view(-1), rows_out.view(-1))                                      # step 6, equal splits
        route = _Route(list(idxs), table_names, D, n_send, [], [], None, None, inbox, slot, feat_off,
                       cap=cap, recv2d=recv2d, feat_table=feat_table)
        return ret, route, overflow, status

    _POOL_KIND = {NRX_BAG_MASKED_MEAN: NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN: NRX_BAG_MEAN, NRX_BAG_SUM: NRX_BAG_SUM}

    def _exchange_pooled(self, feats: Sequence[ShardedFeature], idxs: List[int], inputs, weights, tables: Dict[str, torch.Tensor]):
        """Pooled-bag channel for one group (<= 64 bag features of one dim): normalised weights -> nrx_route_bags ->
        equal-split all-to-alls of {local rows, tags, weights} -> owner-side pooling -> all-to-all of the dense partials.
        Returns (partials [world, nf*B, D] as received from every owner, _Route, overflow, status)."""
        W = self.world
        dev = inputs[idxs[0]].device
        B = inputs[idxs[0]].shape[0]
        dt = torch.int32 if all(inputs[i].dtype == torch.int32 for i in idxs) else torch.int64
        ids = [inputs[i] if inputs[i].dtype == dt else inputs[i].to(dt) for i in idxs]
        wn = [self.backend.bag_norm_weights(weights[i] if feats[i].kind != NRX_BAG_MEAN else None, B, feats[i].bag_len,
                                            feats[i].kind, dev) for i in idxs]
        n_send = sum(x.numel() for x in ids)
        cap = self.capacity_for(n_send)
        table_names: List[str] = []
        for i in idxs:
            if feats[i].table not in table_names:
                table_names.append(feats[i].table)
        feat_table = [table_names.index(feats[i].table) for i in idxs]
        D = feats[idxs[0]].dim
        nf = len(idxs)
        send_rows, send_tag, send_w, counts2d, overflow = self.backend.route_bags(ids, wn, W, cap)
        recv2d = torch.empty_like(counts2d)
        self._a2a(recv2d.view(-1), counts2d.view(-1))
        inbox_rows, inbox_tag, inbox_w = torch.empty_like(send_rows), torch.empty_like(send_tag), torch.empty_like(send_w)
        self._a2a(inbox_rows, send_rows)
        self._a2a(inbox_tag, send_tag)
        self._a2a(inbox_w, send_w)
        want_status = self.backend.index_checks_on() and self.overflow_policy == "check"
        partial, status = self.backend.pool_inbox([tables[t] for t in table_names], feat_table, B, W, cap, recv2d, inbox_rows,
                                                  inbox_tag, inbox_w, want_status)
        ret = torch.empty_like(partial)                                                # [W, nf*B, D]: slab o came from owner o
        self._a2a(ret.view(-1), partial.view(-1))
        route = _Route(list(idxs), table_names, D, n_send, [], [], None, None, inbox_rows, None, [], cap=cap, recv2d=recv2d,
                       feat_table=feat_table, pooled=True, inbox_tag=inbox_tag, inbox_w=inbox_w, batch=B)
        return ret, route, overflow, status

    def _pooled_ids(self, B: int, nf: int, k: int, device) -> torch.Tensor:
        """ids [B, world] of feature k of a pooled group inside the returned slabs viewed as [world*nf*B, D] rows."""
        key = (B, nf, k, str(device))
        cache = self.__dict__.setdefault("_pooled_id_cache", {})
        t = cache.get(key)
        if t is None:
            o = torch.arange(self.world, device=device, dtype=torch.int32)[None, :] * (nf * B)
            t = (o + (k * B + torch.arange(B, device=device, dtype=torch.int32))[:, None]).contiguous()
            cache[key] = t
        return t

    @staticmethod
    def replicated_tables(feats: Sequence[ShardedFeature]) -> List[str]:
        names: List[str] = []
        for f in feats:
            if f.kind != NRX_DENSE and f.replicated and f.table not in names:
                names.append(f.table)
        return names

    def _final_plan(self, feats: Sequence[ShardedFeature], groups: List[List[int]], pooled_groups: Sequence[int] = ()):
        """Step 7 plan: tables = one returned-row buffer per exchange group, then the replicated tables;
        indices = slot[] segments (routed features) or the original ids (replicated features)."""
        slots, col, wcol = [], 0, 0
        gidx = {}
        for g, idxs in enumerate(groups):
            for i in idxs:
                gidx[i] = g
        rep = self.replicated_tables(feats)
        for i, f in enumerate(feats):
            if f.kind == NRX_DENSE:
                slots.append(ops.Slot(f.name, NRX_DENSE, -1, 1, 0, col))
                col += 1
                continue
            kind, blen = f.kind, f.bag_len
            if f.replicated:
                tix, flags = len(groups) + rep.index(f.table), 0
            else:
                tix, flags = gidx[i], NRX_FEAT_ROW0_IS_DATA
                if gidx[i] in pooled_groups:      # the owners pooled: add the `world` partials, rank order
                    kind, blen = NRX_BAG_SUM, self.world
            slots.append(ops.Slot(f.name, kind, tix, f.dim, blen, col, wide_col=wcol if f.wide else -1,
                                  fm_field=int(f.fm), flags=flags))
            if f.wide:
                wcol += 1
                col += f.dim - 1
            else:
                col += f.dim
        use_fm = any(f.fm for f in feats)
        return ops.EmbedPlan(slots, out_width=col, wide_width=wcol, use_fm=use_fm)

    def forward(self, feats: Sequence[ShardedFeature], inputs: Sequence[torch.Tensor],
                weights: Sequence[Optional[torch.Tensor]], tables: Dict[str, torch.Tensor],
                out_ld: Optional[int] = None, need_out: bool = True):
        """Returns (out, wide, fm) like ops.embed_apply; differentiable w.r.t. the local shards."""
        names = sorted(tables)
        return _ShardedEmbedFn.apply(self, list(feats), list(inputs), list(weights), names, out_ld, need_out,
                                     *[tables[n] for n in names])


class _ShardedEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: RowShardedEmbedding, feats, inputs, weights, names, out_ld, need_out, *shards):
        tables = dict(zip(names, shards))
        groups, pooled_set = eng.plan_groups(feats)

        def run(mode):
            rets, routes, flags = [], [], []
            for g, idxs in enumerate(groups):
                if mode == "capacity" and g in pooled_set:
                    ret, route, overflow, status = eng._exchange_pooled(feats, idxs, inputs, weights, tables)
                    flags.append((overflow, route.cap, status))
                    ret = ret.view(-1, route.dim)                       # [world*nf*B, D] pseudo-table of partial sums
                elif mode == "capacity":
                    ret, route, overflow, status = eng._exchange_capacity(feats, idxs, inputs, tables)
                    flags.append((overflow, route.cap, status))
                else:
                    ret, route = eng._exchange(feats, idxs, inputs, tables)
                rets.append(ret)
                routes.append(route)
            return rets, routes, flags

        rets, routes, flags = run(eng.mode)
        if eng.mode == "capacity":
            if eng.overflow_policy == "check":
                # one tiny all-reduce so that every rank takes the same branch, then one host read
                worst = torch.stack([torch.stack([o[0] - cap, (st[0].to(torch.int64) if st is not None else o[0] * 0)])
                                     for o, cap, st in flags]).max(dim=0).values
                eng._all_reduce_max(worst)
                over, bad = worst.tolist()
                if bad > 0:
                    raise IndexError("index out of range in self: a routed lookup exceeded its table on some rank")
                if over > 0:
                    pooled_set = set()                                   # exact mode returns rows for every feature
                    rets, routes, flags = run("exact")
            else:
                eng._overflow_marks += [(o, cap) for o, cap, _ in flags]

        plan = eng._final_plan(feats, groups, pooled_set)
        rep_names = eng.replicated_tables(feats)
        final_inputs = []
        final_weights = list(weights)
        for i, f in enumerate(feats):
            if f.kind == NRX_DENSE or f.replicated:
                final_inputs.append(inputs[i])
                continue
            g = next(k for k, idxs in enumerate(groups) if i in idxs)
            r = routes[g]
            k = r.feats.index(i)
            if r.pooled:
                final_inputs.append(eng._pooled_ids(r.batch, len(r.feats), k, inputs[i].device))
                final_weights[i] = None
                continue
            n = inputs[i].numel()
            final_inputs.append(r.slot[r.feat_off[k]: r.feat_off[k] + n].view(inputs[i].shape))
        train = any(ctx.needs_input_grad[7:])
        with torch.set_grad_enabled(train):
            leaves = [r.detach().requires_grad_(train) for r in rets]
            leaves += [tables[n].detach().requires_grad_(train) for n in rep_names]   # replicated: local full tables
            out, wide, fm = eng.backend.embed(plan, leaves, final_inputs, final_weights, out_ld=out_ld, need_out=need_out)
        ctx.eng, ctx.routes, ctx.leaves, ctx.rep_names = eng, routes, leaves, rep_names
        ctx.outs = (out, wide, fm)
        ctx.names, ctx.shard_meta = names, [(s.shape, s.device) for s in shards]
        ctx.set_materialize_grads(False)
        det = tuple(None if t is None else t.detach() for t in (out, wide, fm))
        return det

    @staticmethod
    def backward(ctx, g_out, g_wide, g_fm):
        eng: RowShardedEmbedding = ctx.eng
        outs, grads_in = [], []
        for t, g in zip(ctx.outs, (g_out, g_wide, g_fm)):
            if t is not None and g is not None:
                outs.append(t)
                grads_in.append(g)
        n_lead = 7
        if not outs:
            return (None,) * (n_lead + len(ctx.names))
        if eng.grad_average and eng.world > 1:
            grads_in = [g / eng.world for g in grads_in]
        g_rets = torch.autograd.grad(outs, ctx.leaves, grads_in, allow_unused=True)     # slot-scatter (HIP bwd kernel)
        shard_grads = {n: torch.zeros(shape, dtype=torch.float32, device=dev) for n, (shape, dev) in zip(ctx.names, ctx.shard_meta)}
        # replicated tables: the LOCAL dense grad of a data-parallel parameter -- the caller all-reduces it together with the
        # other dense parameters (allreduce_dense_grads(data_parallel_params(model), world) averages, which undoes the
        # 1 / world applied above only once: the local grad is kept unscaled here)
        for n, g_rep in zip(ctx.rep_names, g_rets[len(ctx.routes):]):
            if g_rep is not None:
                shard_grads[n] = g_rep * eng.world if (eng.grad_average and eng.world > 1) else g_rep
        for route, leaf, g_ret in zip(ctx.routes, ctx.leaves, g_rets):
            if g_ret is None:
                g_ret = torch.zeros_like(leaf)
            D = route.dim
            gts = [shard_grads[t] for t in route.table_names]
            if route.pooled:
                g_part = torch.empty_like(g_ret)
                eng._a2a(g_part.view(-1), g_ret.contiguous().view(-1))                   # slab s goes back to owner... from source s
                eng.backend.pool_inbox_bwd(gts, route.feat_table, route.batch, eng.world, route.cap, route.recv2d, route.recv_rows,
                                           route.inbox_tag, route.inbox_w, g_part.view(eng.world, -1, D), skip_row0=(eng.rank == 0))
            elif route.cap:
                g_recv = torch.empty_like(g_ret)
                eng._a2a(g_recv.view(-1), g_ret.contiguous().view(-1))                    # equal splits
                eng.backend.scatter_add_inbox(gts, route.feat_table, eng.world, route.cap, route.recv2d,
                                              route.recv_rows, g_recv, skip_row0=(eng.rank == 0))
            else:
                n_recv = int(sum(route.recv_counts))
                g_recv = torch.empty((n_recv, D), dtype=torch.float32, device=leaf.device)
                eng._a2a(g_recv.view(-1), g_ret.contiguous().view(-1), [c * D for c in route.recv_counts],
                         [c * D for c in route.send_counts])
                eng.backend.scatter_add_segmented(gts, route.seg_start, route.seg_table, route.recv_rows, g_recv,
                                                  skip_row0=(eng.rank == 0))
        return (None,) * n_lead + tuple(shard_grads[n] for n in ctx.names)


# --------------------------------------------------------------------------------- bound forward
class PreparedShardedForward:
    """A bound, re-launchable sync-free sharded forward (inference / benchmarking; HIP backend only):
    all exchange buffers, C descriptor arrays and the final fused launch are built once; `run()` only
    enqueues 3 routing kernels + 3 equal-split all-to-alls + the owner gather per exchange group, then
    the final launch.  ids / masks are re-read from the SAME tensors on every run.  `overflowed()`
    reports (with one host read) whether any block exceeded its capacity since the last call."""

    def __init__(self, eng: RowShardedEmbedding, feats: Sequence[ShardedFeature], inputs, weights,
                 tables: Dict[str, torch.Tensor], out_ld: Optional[int] = None,
                 out: Optional[torch.Tensor] = None, fm: Optional[torch.Tensor] = None, overlap_local: bool = True,
                 one_sided: Optional[bool] = None):
        """one_sided (default: the NRX_SHARD_ONE_SIDED environment variable, else off): ONE-SIDED PLACEMENT for the exchange groups of plain
        single-valued features -- the owner's gather writes every row straight into its place in the REQUESTER's concat buffer, which every
        rank maps at construction (hipIpc through torch's CUDA-IPC tensor sharing; over xGMI a peer mapping), instead of into a row buffer
        that an all-to-all carries back and a final launch reads again.  Per step: ids + sample positions out (two int32 all-to-alls), the
        owner's launch, one small collective as the completion fence.  Bags, wide-routed and dense features keep the buffer path; an FM
        epilogue becomes a pass over the finished concat (nrx_fm_fwd)."""
        import ctypes as C
        import os
        from . import _lib
        self.lib = _lib.load()
        self.eng = eng
        W = eng.world
        groups, pooled_set = eng.plan_groups(feats)
        self.groups = []
        self.keep = [inputs, weights, tables]
        slot_of: Dict[int, torch.Tensor] = {}
        final_weights = list(weights)
        rets = []
        if one_sided is None:
            one_sided = os.environ.get("NRX_SHARD_ONE_SIDED", "0") == "1"
        plan0 = eng._final_plan(feats, groups, pooled_set)
        ld0 = int(out_ld) if out_ld else plan0.out_width
        self.placed: List[int] = []                  # feature indices written by the owners (not by this rank's final launch)
        self.peers = None
        if one_sided and not eng.dedup and (ld0 & 3) == 0:
            for gi, idxs in enumerate(groups):
                D = feats[idxs[0]].dim
                if gi in pooled_set or D % 4 or not (16 <= D <= 256):
                    continue
                # worth its two extra launches when rows cross the fabric (W > 1: it replaces the row all-to-all) or, at world 1, when the
                # group is large (C4's two id features, 131 k rows: 166 -> 178 us with it; C5's 40 features: 277 -> 160 us)
                if W == 1 and sum(inputs[i].numel() for i in idxs) < (1 << 20) and os.environ.get("NRX_SHARD_ONE_SIDED_MIN") is None:
                    continue
                if all(feats[i].kind == NRX_SPARSE and not feats[i].wide and plan0.slots[i].out_col % 4 == 0 and inputs[i].dim() == 1
                       for i in idxs):
                    self.placed += idxs
        if self.placed and plan0.use_fm and (len(self.placed) != len(feats) or any(not f.fm for f in feats) or len({f.dim for f in feats}) != 1):
            self.placed = []      # an FM epilogue over a mix of placed and locally written features: keep the buffer path (the fused launch does it)
        placed_set = set(self.placed)
        if placed_set:
            B0 = inputs[0].shape[0]
            if out is None:
                out = torch.empty((B0, ld0), dtype=torch.float32, device=inputs[0].device)
            self.peers = self._map_peer_buffers(eng, out)
            self._peer_ptrs = (C.c_void_p * W)(*[t.data_ptr() for t in self.peers])
            self._fence = (torch.zeros(W, dtype=torch.int32, device=out.device), torch.zeros(W, dtype=torch.int32, device=out.device))
        for gi, idxs in enumerate(groups):
            dev = inputs[idxs[0]].device
            if gi in pooled_set:
                g = self._bind_pooled(eng, feats, idxs, inputs, weights, tables, C)
                for k, i in enumerate(idxs):
                    slot_of[i] = eng._pooled_ids(g["B"], g["n"], k, dev)
                    final_weights[i] = None
                self.groups.append(g)
                rets.append(g["ret"].view(-1, g["D"]))
                continue
            dt = inputs[idxs[0]].dtype
            if dt not in (torch.int32, torch.int64) or any(inputs[i].dtype != dt or not inputs[i].is_contiguous() for i in idxs):
                # a converted copy would be a snapshot: the bound call promises to re-read the caller's tensors every run
                raise TypeError("PreparedShardedForward: the ids of one exchange group must be contiguous and share one dtype "
                                "(int32 or int64)")
            ids = [inputs[i] for i in idxs]
            n = len(ids)
            total = sum(x.numel() for x in ids)
            cap = eng.capacity_for(total)
            D = feats[idxs[0]].dim
            table_names: List[str] = []
            for i in idxs:
                if feats[i].table not in table_names:
                    table_names.append(feats[i].table)
            loc = [tables[t] for t in table_names]
            g = dict(
                n=n, cap=cap, D=D, bits=ids[0].element_size() * 8, ids=ids,
                ptrs=(C.c_void_p * n)(*[x.data_ptr() for x in ids]), lens=(C.c_int64 * n)(*[x.numel() for x in ids]),
                send=torch.empty(W * cap, dtype=torch.int32, device=dev), slot=torch.empty(total, dtype=torch.int32, device=dev),
                counts2d=torch.empty((W, n), dtype=torch.int64, device=dev), recv2d=torch.empty((W, n), dtype=torch.int64, device=dev),
                overflow=torch.zeros(1, dtype=torch.int64, device=dev),
                ws=torch.empty(max(1, self.lib.nrx_route_workspace(total, W)), dtype=torch.int64, device=dev),
                inbox=torch.empty(W * cap, dtype=torch.int32, device=dev),
                rows_out=torch.empty((W * cap, D), dtype=torch.float32, device=dev),
                ret=torch.empty((W * cap, D), dtype=torch.float32, device=dev),
                # (an empty shard -- fewer table rows than ranks -- has no address: a zero row stands in, the row count stays 0)
                stand_in=torch.zeros((1, D), dtype=torch.float32, device=dev),
                tr=(C.c_int64 * len(loc))(*[t.shape[0] for t in loc]),
                nt=len(loc), ft=(C.c_int32 * n)(*[table_names.index(feats[i].table) for i in idxs]), dev=dev)
            g["tp"] = (C.c_void_p * len(loc))(*[(t if t.shape[0] else g["stand_in"]).data_ptr() for t in loc])
            if idxs[0] in placed_set:     # one-sided placement: no row buffers at all; the sample positions travel with the local rows
                g["placed"] = True
                g["rows_out"] = g["ret"] = None
                g["send_pos"] = torch.empty(W * cap, dtype=torch.int32, device=dev)
                g["inbox_pos"] = torch.empty(W * cap, dtype=torch.int32, device=dev)
                g["cols"] = (C.c_int32 * n)(*[plan0.slots[i].out_col for i in idxs])
                g["ld"] = ld0
                if W == 1:
                    g["inbox"], g["recv2d"], g["inbox_pos"] = g["send"], g["counts2d"], g["send_pos"]
                self.groups.append(g)
                continue
            if W == 1:        # a one-rank group exchanges with itself: the "received" buffers ARE the sent ones (no copies)
                g["inbox"], g["recv2d"], g["ret"] = g["send"], g["counts2d"], g["rows_out"]
            off = 0
            for i, x in zip(idxs, ids):
                slot_of[i] = g["slot"][off: off + x.numel()].view(inputs[i].shape)
                off += x.numel()
            self.groups.append(g)
            rets.append(g["ret"])
        plan = plan0
        rets += [tables[n] for n in eng.replicated_tables(feats)]
        self.fm_pass = None
        if placed_set:
            # the final launch covers what the owners did not place; the table indices of its slots still name `rets` entries in the order
            # _final_plan assigned them, so placed groups keep a (never read) placeholder there
            rest = [i for i in range(len(feats)) if i not in placed_set]
            self.local = None
            self.final = None
            self.out = out
            if plan.use_fm:
                if fm is None:
                    fm = torch.empty((inputs[0].shape[0],), dtype=torch.float32, device=out.device)
                self.fm_pass = (len(feats), feats[0].dim, fm)
            self.fm_out = fm
            if rest:
                ph = torch.empty((1, 1), dtype=torch.float32, device=out.device)
                rets_full, k = [], 0
                for gi, idxs in enumerate(groups):
                    if idxs[0] in placed_set:
                        rets_full.append(ph)
                    else:
                        rets_full.append(rets[k])
                        k += 1
                rets_full += rets[k:]
                final_inputs = [inputs[i] if (f.kind == NRX_DENSE or f.replicated) else slot_of.get(i) for i, f in enumerate(feats)]
                sp = ops.EmbedPlan([plan.slots[i] for i in rest], out_width=plan.out_width,
                                   wide_width=plan.wide_width if any(feats[i].wide for i in rest) else 0)
                self.final = ops.PreparedEmbed(sp, rets_full, [final_inputs[i] for i in rest], [final_weights[i] for i in rest],
                                               out_ld=ld0, out=out)
            return
        final_inputs = [inputs[i] if (f.kind == NRX_DENSE or f.replicated) else slot_of[i] for i, f in enumerate(feats)]
        # Features that need no exchange (dense values, planner-replicated tables) do not wait for one: they get their own
        # launch on a side stream, concurrent with routing / all-to-alls / owner gather, and the routed features get a second
        # launch after the exchange -- both write their own columns of the same concat.  Not possible when one epilogue
        # spans all features (FM) or when both halves carry wide columns.
        local = [i for i, f in enumerate(feats) if f.kind == NRX_DENSE or f.replicated]
        routed = [i for i in range(len(feats)) if i not in set(local)]
        wide_l = any(feats[i].wide for i in local)
        wide_r = any(feats[i].wide for i in routed)
        self.local = None
        # worth two launches + two cross-stream waits only when the exchange is long: always with real peers (collective
        # latencies), at world 1 only for a large exchange (measured at world 1: C5 17 routed tables 275 -> 225 us, but C4 /
        # C3 with one or two routed id features 75 -> 88 / 57 -> 62 us: the step is then bound by the host's launch rate)
        n_routed = sum(inputs[i].numel() for i in routed)
        if overlap_local and local and routed and not plan.use_fm and not (wide_l and wide_r) and \
                len(feats) <= ops.NRX_MAX_FEATURES and (W > 1 or n_routed >= (1 << 20)):
            ld = int(out_ld) if out_ld else plan.out_width
            B = inputs[0].shape[0]
            dev0 = inputs[0].device
            if out is None:
                out = torch.empty((B, ld), dtype=torch.float32, device=dev0)

            def sub(idx):
                sp = ops.EmbedPlan([plan.slots[i] for i in idx], out_width=plan.out_width,
                                   wide_width=plan.wide_width if any(feats[i].wide for i in idx) else 0)
                return ops.PreparedEmbed(sp, rets, [final_inputs[i] for i in idx], [final_weights[i] for i in idx],
                                         out_ld=ld, out=out)
            self.local = sub(local)
            self.final = sub(routed)
            self._side = torch.cuda.Stream(device=dev0)
            self._wide_from_local = wide_l
        else:
            self.final = ops.PreparedEmbed(plan, rets, final_inputs, final_weights, out_ld=out_ld, out=out, fm=fm)

    @staticmethod
    def _map_peer_buffers(eng, out: torch.Tensor):
        """Every rank's concat buffer as THIS process addresses it: its own tensor, and for the peers a tensor rebuilt from the CUDA-IPC handle
        torch exports (hipIpcGetMemHandle / hipIpcOpenMemHandle underneath; two rank processes on one GPU -- the test layout -- map each other
        the same way).  With one process per GPU the mapping is a peer mapping over xGMI: a first tiny copy makes torch enable peer access."""
        W = eng.world
        if W == 1:
            return [out]
        import os
        from torch.multiprocessing.reductions import reduce_tensor
        # A mapping that cannot be made (no IPC between the processes, no peer access between the devices) must not strand the other ranks in a
        # collective: every rank tries, the ranks agree (one small all-reduce), and ALL raise PeerMappingError together -- the callers then take
        # the buffered (all-to-all) forms, which need no mapping.  NRX_DEBUG_FAIL_PEER_MAP=<rank>: that rank pretends to fail (tests).
        peers, err = [], None
        try:
            handles = [None] * W
            dist.all_gather_object(handles, reduce_tensor(out), group=eng.group)
            if os.environ.get("NRX_DEBUG_FAIL_PEER_MAP") == str(eng.rank):
                raise RuntimeError("NRX_DEBUG_FAIL_PEER_MAP")
            for s in range(W):
                if s == eng.rank:
                    peers.append(out)
                    continue
                fn, args = handles[s]
                t = fn(*args)
                if t.device != out.device:
                    out[:1, :1].copy_(t[:1, :1])           # (enables peer access between the two devices; the element is rewritten every step)
                peers.append(t)
            if out.is_cuda:
                torch.cuda.synchronize(out.device)
        except Exception as e:          # noqa: BLE001 -- whatever went wrong, the ranks must leave this function together
            err = e
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int64, device=out.device)
        eng._all_reduce_max(bad)
        if int(bad.item()):
            raise PeerMappingError("one-sided placement needs every rank's buffer mapped into every other rank (CUDA IPC / peer access): "
                                   + (f"this rank failed with {type(err).__name__}: {err}" if err is not None else "another rank could not map it"))
        dist.barrier(group=eng.group)
        return peers

    def _run_placed(self, g, stream):
        """One-sided group: route (rows + sample positions) -> counts / rows / positions to the owners -> the owner's gather writes into the
        requesters' buffers."""
        eng, lib = self.eng, self.lib
        W = eng.world
        rc = lib.nrx_route_ids_pos(g["ptrs"], g["lens"], g["n"], g["bits"], W, g["cap"], g["send"].data_ptr(), g["send_pos"].data_ptr(),
                                   g["slot"].data_ptr(), g["counts2d"].data_ptr(), g["overflow"].data_ptr(), g["ws"].data_ptr(), stream)
        if rc:
            ops.check(rc, "nrx_route_ids_pos")
        if W > 1:
            eng._a2a(g["recv2d"].view(-1), g["counts2d"].view(-1))
            eng._a2a(g["inbox"], g["send"])
            eng._a2a(g["inbox_pos"], g["send_pos"])
        rc = lib.nrx_gather_inbox_place(g["tp"], g["tr"], g["nt"], g["ft"], g["n"], W, g["cap"], g["recv2d"].data_ptr(),
                                        g["inbox"].data_ptr(), g["inbox_pos"].data_ptr(), g["D"], self._peer_ptrs, g["ld"], self.out.shape[0],
                                        g["cols"], None, stream)
        if rc:
            ops.check(rc, "nrx_gather_inbox_place")

    def _bind_pooled(self, eng, feats, idxs, inputs, weights, tables, C):
        """Buffers and descriptor arrays of one pooled-bag group (owner-side partial pooling)."""
        W = eng.world
        dev = inputs[idxs[0]].device
        dt = torch.int32 if all(inputs[i].dtype == torch.int32 for i in idxs) else torch.int64
        if any(inputs[i].dtype != dt for i in idxs):
            raise TypeError("PreparedShardedForward: the ids of one exchange group must share a dtype")
        ids = [inputs[i] if inputs[i].is_contiguous() else inputs[i].contiguous() for i in idxs]
        n = len(ids)
        B = ids[0].shape[0]
        total = sum(x.numel() for x in ids)
        cap = eng.capacity_for(total)
        D = feats[idxs[0]].dim
        table_names: List[str] = []
        for i in idxs:
            if feats[i].table not in table_names:
                table_names.append(feats[i].table)
        loc = [tables[t] for t in table_names]
        # a table with fewer rows than ranks leaves some ranks an EMPTY shard (a 5-row category table at world 8): its tensor has no address, and
        # the pooling launch -- which reads row 0 for the entries it masks out -- wants one: a zero row stands in (the row count stays 0, so
        # every entry that reaches this owner is reported as out of range and contributes nothing)
        stand_in = torch.zeros((1, D), dtype=torch.float32, device=dev)
        masks = [weights[i] if feats[i].kind != NRX_BAG_MEAN else None for i in idxs]
        wn = [torch.empty((B, feats[i].bag_len), dtype=torch.float32, device=dev) for i in idxs]
        g = dict(pooled=True, n=n, B=B, cap=cap, D=D, bits=ids[0].element_size() * 8, ids=ids, masks=masks, wn=wn,
                 kinds=[feats[i].kind for i in idxs], lens=[feats[i].bag_len for i in idxs],
                 ptrs=(C.c_void_p * n)(*[x.data_ptr() for x in ids]), wptrs=(C.c_void_p * n)(*[w.data_ptr() for w in wn]),
                 bl=(C.c_int32 * n)(*[feats[i].bag_len for i in idxs]),
                 send=torch.empty(W * cap, dtype=torch.int32, device=dev), send_tag=torch.empty(W * cap, dtype=torch.int32, device=dev),
                 send_w=torch.empty(W * cap, dtype=torch.float32, device=dev),
                 counts2d=torch.empty((W, n), dtype=torch.int64, device=dev), recv2d=torch.empty((W, n), dtype=torch.int64, device=dev),
                 overflow=torch.zeros(1, dtype=torch.int64, device=dev),
                 ws=torch.empty(max(1, self.lib.nrx_route_workspace(total, W)), dtype=torch.int64, device=dev),
                 inbox=torch.empty(W * cap, dtype=torch.int32, device=dev), inbox_tag=torch.empty(W * cap, dtype=torch.int32, device=dev),
                 inbox_w=torch.empty(W * cap, dtype=torch.float32, device=dev),
                 partial=torch.empty((W, n * B, D), dtype=torch.float32, device=dev),
                 pws=torch.empty(max(1, self.lib.nrx_pool_inbox_workspace(n, B, W)), dtype=torch.uint8, device=dev),
                 ret=torch.empty((W, n * B, D), dtype=torch.float32, device=dev),
                 tp=(C.c_void_p * len(loc))(*[(t if t.shape[0] else stand_in).data_ptr() for t in loc]), tr=(C.c_int64 * len(loc))(*[t.shape[0] for t in loc]),
                 nt=len(loc), ft=(C.c_int32 * n)(*[table_names.index(feats[i].table) for i in idxs]), dev=dev, stand_in=stand_in)
        if W == 1:
            g["inbox"], g["inbox_tag"], g["inbox_w"], g["recv2d"], g["ret"] = g["send"], g["send_tag"], g["send_w"], g["counts2d"], g["partial"]
        import os
        # routing: "runs" (default, round 6) = ONE launch that also normalises the weights and writes the (owner, tag) run bounds the owner's pooling
        # launch needs (they travel in the place of the per-entry tags); "one" = the one-launch routing behind nrx_bag_norm_weights, runs found by
        # the owner (memset + marking pass); "legacy" = histogram + scan + placement launches
        how = os.environ.get("NRX_ROUTE_BAGS", "runs")
        if how == "runs":
            nb = self.lib.nrx_route_bags_runs_state_bytes(g["bl"], n, B, W)
            if nb > 0:
                g["runs_state"] = torch.zeros(nb, dtype=torch.uint8, device=dev)
                g["send_run"] = torch.empty((W, n * B, 2), dtype=torch.int32, device=dev)
                g["run"] = g["send_run"] if W == 1 else torch.empty((W, n * B, 2), dtype=torch.int32, device=dev)
                g["mptrs"] = (C.c_void_p * n)(*[(0 if m is None else m.data_ptr()) for m in masks])
                g["kinds_c"] = (C.c_int32 * n)(*g["kinds"])
                if any(m is not None and (not m.is_contiguous() or m.dtype != torch.float32) for m in masks):
                    raise ValueError("PreparedShardedForward: the masks of a pooled group must be contiguous float32 (they are bound by address)")
            else:
                how = "one"
        if how == "one":
            nb = self.lib.nrx_route_bags_one_state_bytes(g["bl"], n, B, W)
            if nb > 0:
                g["rstate"] = torch.zeros(nb, dtype=torch.uint8, device=dev)
        return g

    def _run_pooled_runs(self, g, stream):
        """The pooled channel with the run bounds written by the routing launch (nrx_route_bags_runs -> nrx_pool_inbox_fwd_runs): two launches
        instead of five (+ a memset), the tags stay home."""
        eng, lib = self.eng, self.lib
        W = eng.world
        if "inv_c" not in g:
            inv = g.get("inv")
            g["inv_c"] = (ctypes.c_void_p * g["n"])(*[t.data_ptr() for t in inv]) if inv is not None else None
        ip = g["inv_c"]
        rc = lib.nrx_route_bags_runs(g["ptrs"], g["mptrs"], g["kinds_c"], g["bl"], g["n"], g["bits"], g["B"], W, g["cap"], g["send"].data_ptr(),
                                     g["send_w"].data_ptr(), g["send_run"].data_ptr(), ip, g["counts2d"].data_ptr(), g["overflow"].data_ptr(),
                                     g["runs_state"].data_ptr(), stream)
        if rc:
            ops.check(rc, "nrx_route_bags_runs")
        if W > 1:
            eng._a2a(g["recv2d"].view(-1), g["counts2d"].view(-1))
            eng._a2a(g["inbox"], g["send"])
            eng._a2a(g["run"].view(-1), g["send_run"].view(-1))
            eng._a2a(g["inbox_w"], g["send_w"])
        rc = lib.nrx_pool_inbox_fwd_runs(g["tp"], g["tr"], g["nt"], g["ft"], g["n"], g["B"], W, g["cap"], g["recv2d"].data_ptr(),
                                         g["inbox"].data_ptr(), g["inbox_w"].data_ptr(), g["run"].data_ptr(), g["D"], g["partial"].data_ptr(),
                                         None, stream)
        if rc:
            ops.check(rc, "nrx_pool_inbox_fwd_runs")
        if W > 1:
            eng._a2a(g["ret"].view(-1), g["partial"].view(-1))

    def _run_pooled(self, g, stream):
        eng, lib = self.eng, self.lib
        W = eng.world
        if g.get("runs_state") is not None:
            return PreparedShardedForward._run_pooled_runs(self, g, stream)
        for k, (m, w, kind, L) in enumerate(zip(g["masks"], g["wn"], g["kinds"], g["lens"])):
            if g.get("inv") is not None:       # (the bound training step, 0/1 masks: the per-sample weight rides along)
                rc = lib.nrx_bag_norm_weights_inv(None if m is None else m.data_ptr(), g["B"], L, kind, w.data_ptr(), g["inv"][k].data_ptr(), stream)
            else:
                rc = lib.nrx_bag_norm_weights(None if m is None else m.data_ptr(), g["B"], L, kind, w.data_ptr(), stream)
            if rc:
                ops.check(rc, "nrx_bag_norm_weights")
        if g.get("rstate") is not None:
            rc = lib.nrx_route_bags_one(g["ptrs"], g["wptrs"], g["bl"], g["n"], g["bits"], g["B"], W, g["cap"], g["send"].data_ptr(),
                                        g["send_tag"].data_ptr(), g["send_w"].data_ptr(), g["counts2d"].data_ptr(), g["overflow"].data_ptr(),
                                        g["rstate"].data_ptr(), stream)
        else:
            rc = lib.nrx_route_bags(g["ptrs"], g["wptrs"], g["bl"], g["n"], g["bits"], g["B"], W, g["cap"], g["send"].data_ptr(),
                                    g["send_tag"].data_ptr(), g["send_w"].data_ptr(), g["counts2d"].data_ptr(), g["overflow"].data_ptr(),
                                    g["ws"].data_ptr(), stream)
        if rc:
            ops.check(rc, "nrx_route_bags")
        if W > 1:
            eng._a2a(g["recv2d"].view(-1), g["counts2d"].view(-1))
            eng._a2a(g["inbox"], g["send"])
            eng._a2a(g["inbox_tag"], g["send_tag"])
            eng._a2a(g["inbox_w"], g["send_w"])
        rc = lib.nrx_pool_inbox_fwd(g["tp"], g["tr"], g["nt"], g["ft"], g["n"], g["B"], W, g["cap"], g["recv2d"].data_ptr(),
                                    g["inbox"].data_ptr(), g["inbox_tag"].data_ptr(), g["inbox_w"].data_ptr(), g["D"],
                                    g["partial"].data_ptr(), g["pws"].data_ptr(), None, stream)
        if rc:
            ops.check(rc, "nrx_pool_inbox_fwd")
        if W > 1:
            eng._a2a(g["ret"].view(-1), g["partial"].view(-1))

    def run(self):
        eng, lib = self.eng, self.lib
        W = eng.world
        if self.peers is not None:
            return self._run_one_sided()
        if self.local is not None:          # the exchange-free features start now, on the side stream
            cur = torch.cuda.current_stream(self._side.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                lres = self.local.run()
        for g in self.groups:
            stream = torch.cuda.current_stream(g["dev"]).cuda_stream
            if g.get("pooled"):
                self._run_pooled(g, stream)
                continue
            rc = lib.nrx_route_ids(g["ptrs"], g["lens"], g["n"], g["bits"], W, g["cap"], g["send"].data_ptr(),
                                   g["slot"].data_ptr(), g["counts2d"].data_ptr(), g["overflow"].data_ptr(),
                                   g["ws"].data_ptr(), stream)
            if rc:
                ops.check(rc, "nrx_route_ids")
            if W > 1:
                eng._a2a(g["recv2d"].view(-1), g["counts2d"].view(-1))
                eng._a2a(g["inbox"], g["send"])
            rc = lib.nrx_gather_inbox(g["tp"], g["tr"], g["nt"], g["ft"], g["n"], W, g["cap"], g["recv2d"].data_ptr(),
                                      g["inbox"].data_ptr(), g["D"], g["rows_out"].data_ptr(), None, stream)
            if rc:
                ops.check(rc, "nrx_gather_inbox")
            if W > 1:
                eng._a2a(g["ret"].view(-1), g["rows_out"].view(-1))
        if self.local is None:
            return self.final.run()
        res = self.final.run()
        torch.cuda.current_stream(self._side.device).wait_stream(self._side)      # both halves of the concat are in place
        return res[0], (lres[1] if self._wide_from_local else res[1]), None

    def _run_one_sided(self):
        eng, lib = self.eng, self.lib
        W = eng.world
        for g in self.groups:
            stream = torch.cuda.current_stream(g["dev"]).cuda_stream
            if g.get("pooled"):
                self._run_pooled(g, stream)
            elif g.get("placed"):
                self._run_placed(g, stream)
            else:
                rc = lib.nrx_route_ids(g["ptrs"], g["lens"], g["n"], g["bits"], W, g["cap"], g["send"].data_ptr(),
                                       g["slot"].data_ptr(), g["counts2d"].data_ptr(), g["overflow"].data_ptr(),
                                       g["ws"].data_ptr(), stream)
                if rc:
                    ops.check(rc, "nrx_route_ids")
                if W > 1:
                    eng._a2a(g["recv2d"].view(-1), g["counts2d"].view(-1))
                    eng._a2a(g["inbox"], g["send"])
                rc = lib.nrx_gather_inbox(g["tp"], g["tr"], g["nt"], g["ft"], g["n"], W, g["cap"], g["recv2d"].data_ptr(),
                                          g["inbox"].data_ptr(), g["D"], g["rows_out"].data_ptr(), None, stream)
                if rc:
                    ops.check(rc, "nrx_gather_inbox")
                if W > 1:
                    eng._a2a(g["ret"].view(-1), g["rows_out"].view(-1))
        res = self.final.run() if self.final is not None else (self.out, None, None)
        if W > 1:
            # completion fence: a collective enqueued behind every rank's placing launch -- when it has completed here, every owner's rows
            # are in this rank's buffer (and the next step's first all-to-all keeps the owners from overwriting it too early)
            eng._a2a(self._fence[0], self._fence[1])
        fmv = None
        if self.fm_pass is not None:
            n, D, fmv = self.fm_pass
            out = self.out
            rc = lib.nrx_fm_fwd(out.data_ptr(), out.shape[1], n, D, out.shape[0], fmv.data_ptr(),
                                torch.cuda.current_stream(out.device).cuda_stream)
            if rc:
                ops.check(rc, "nrx_fm_fwd")
        return self.out, (res[1] if res is not None else None), fmv

    def overflowed(self) -> bool:
        """True if any run since the last call exceeded a block capacity (the kernels keep a running maximum)."""
        bad = False
        for g in self.groups:
            bad |= int(g["overflow"].item()) > g["cap"]
            g["overflow"].zero_()
        return bad


# --------------------------------------------------------------------------------- dense params
def allreduce_dense_grads(params, world: int, group=None) -> None:
    """Data-parallel dense parameters (MLP / cross / FM bias): one flat bucketed all-reduce, averaged.
    They are tiny here (<= ~0.3 MB), so a single latency-bound collective is the right shape."""
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat.div_(world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


# --------------------------------------------------------------------------------- model integration
def data_parallel_params(model):
    """Parameters that are REPLICATED on every rank of a shard_model_'ed model and therefore need the gradient
    all-reduce: everything except the row-sharded tables (dense heads, cross / FM parameters, planner-replicated
    tables).  Use: allreduce_dense_grads(data_parallel_params(model), world)."""
    sharded = {id(e.weight) for n, e in model.embedding_tables.items() if n not in getattr(model, "_replicated_tables", ())}
    return [p for p in model.parameters() if id(p) not in sharded]


def shard_model_(model, rank: int, world: int, group=None, backend=None):
    """Convert a BaseModel in place to row-sharded tables: every `embedding_tables[name].weight` becomes
    the local shard (rows rank::world) and `_embed` is routed through the exchange engine.  The
    state_dict keys are unchanged; values are the local shards (use shard_table / unshard_tables to
    convert checkpoints: scatter-on-load / gather-on-save)."""
    import torch.nn as nn
    if getattr(model, "sparse_grad", False):
        # the routed backward delivers dense shard gradients; the row-sparse optimizers expect COO grads / the fused sink
        raise NotImplementedError("shard_model_: embeddings.sparse_grad is not supported together with row-sharded tables")
    model._replicated_tables = ()
    eng = RowShardedEmbedding(rank, world, group, backend)
    for name, emb in model.embedding_tables.items():
        local = shard_table(emb.weight.data, rank, world)
        new = nn.Embedding(max(1, local.shape[0]), local.shape[1])
        new.weight = nn.Parameter(local if local.shape[0] else local.new_zeros((1, local.shape[1])))
        new.global_rows = emb.num_embeddings
        model.embedding_tables[name] = new
    model._shard_engine = eng

    def _embed_sharded(batch, feature_names, fm=False, wide_names=(), out_ld=None, need_out=True):
        plan, table_names, dims, present = model._plan(batch, feature_names, fm, wide_names)
        if not present:
            return None, None, None, [], []
        if out_ld is not None and out_ld < 0:       # "pad the row stride" request of the single-GPU path (BaseModel._embed): not used here
            out_ld = None
        feats = []
        for s in plan.slots:
            tname = '' if s.kind == NRX_DENSE else table_names[s.table]
            feats.append(ShardedFeature(s.name, s.kind, tname, s.dim, s.bag_len, s.wide_col >= 0, bool(s.fm_field)))
        inputs = [batch[s.name] for s in plan.slots]
        weights = [batch.get(f"{s.name}_mask") if s.kind == NRX_BAG_MASKED_MEAN else None for s in plan.slots]
        for i, s in enumerate(plan.slots):          # CSR bags (name + "_offsets"): the exchange routes the padded form
            if s.flags & NRX_FEAT_BAG_CSR:
                inputs[i], weights[i] = ops.csr_to_padded(inputs[i], batch[f"{s.name}_offsets"], s.bag_len)
        tables = {t: model.embedding_tables[t].weight for t in table_names}
        out, wide, fmv = eng.forward(feats, inputs, weights, tables, out_ld=out_ld, need_out=need_out)
        return out, wide, fmv, list(dims), list(present)

    model._embed = _embed_sharded
    return model


def full_state_dict(model, group=None) -> Dict[str, torch.Tensor]:
    """Gather-on-save for a model converted by shard_model_: the REFERENCE's state_dict -- `embedding_tables.<name>.weight` as the full
    [rows, D] table (base_model.py:531-536 loads it with strict=True), every other entry as it is (dense parameters are replicated) --
    on every rank.  One all-gather per sharded table (shards padded to the longest, rank 0's); a checkpoint written from it loads into the
    reference, into the unsharded mirror, or back into any world size through load_full_state_dict_."""
    eng = getattr(model, "_shard_engine", None)
    world = eng.world if eng is not None else 1
    out: Dict[str, torch.Tensor] = {}
    for k, v in model.state_dict().items():
        name = k[len("embedding_tables."):-len(".weight")] if k.startswith("embedding_tables.") and k.endswith(".weight") else None
        emb = model.embedding_tables[name] if name is not None and name in model.embedding_tables else None
        rows = getattr(emb, "global_rows", None)
        if rows is None or name in getattr(model, "_replicated_tables", ()) or (world == 1 and not getattr(emb, "arena", False)):
            out[k] = v.detach().clone()
            continue
        if world == 1:
            out[k] = v.detach()[1:].clone()
            continue
        longest = local_row_count(rows, 0, world)
        mine = local_row_count(rows, eng.rank, world)
        pad = v.new_zeros((longest, v.shape[1]))
        lo = 1 if getattr(emb, "arena", False) else 0      # (shard_step.shard_model_step_: the local table is an arena with a leading dummy row)
        pad[:mine] = v.detach()[lo:lo + mine]
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group if group is not None else eng.group)
        out[k] = unshard_tables([p[:local_row_count(rows, r, world)] for r, p in enumerate(parts)])
    return out


def load_full_state_dict_(model, full: Dict[str, torch.Tensor], strict: bool = True):
    """Scatter-on-load: a full (reference-shaped) state_dict into a model converted by shard_model_ -- every rank keeps rows rank::world of
    each sharded table (a rank that owns no row of a tiny table keeps its one-row zero placeholder)."""
    eng = getattr(model, "_shard_engine", None)
    world, rank = (eng.world, eng.rank) if eng is not None else (1, 0)
    local = {}
    for k, v in full.items():
        name = k[len("embedding_tables."):-len(".weight")] if k.startswith("embedding_tables.") and k.endswith(".weight") else None
        emb = model.embedding_tables[name] if name is not None and name in model.embedding_tables else None
        rows = getattr(emb, "global_rows", None)
        if rows is None or name in getattr(model, "_replicated_tables", ()) or (world == 1 and not getattr(emb, "arena", False)):
            local[k] = v
            continue
        if v.shape[0] != rows:
            raise ValueError(f"load_full_state_dict_: {k} has {v.shape[0]} rows, the sharded model was built for {rows}")
        sh = shard_table(v, rank, world)
        if getattr(emb, "arena", False):
            sh = torch.cat([v.new_zeros((1, v.shape[1])), sh])
            if rank == 0 and sh.shape[0] > 1:
                sh[1].zero_()
            local[k] = sh
            continue
        local[k] = sh if sh.shape[0] else v.new_zeros((1, v.shape[1]))
    return model.load_state_dict(local, strict=strict)


# --------------------------------------------------------------------------------- bench path (N > 1)
class ShardedBenchPath:
    """bench.py's N>1 workload: the same synthetic configuration as the single-GPU path, tables
    row-sharded over `world` ranks, B impressions per rank (weak scaling)."""

    def __init__(self, wl: str, device, seed: int, rank: int, world: int, batch: int, mode: str = "row", n_pool: int = 8,
                 replicate_below_bytes: int = 256 << 20, host_staged: bool = False, engine: Optional[str] = None):
        """mode "row": every table row-sharded (the north-star layout).  mode "auto": planner -- tables
        of at most `replicate_below_bytes` are held in full on every rank (no exchange for them), larger
        ones are row-sharded.
        engine (default: NRX_SHARD_ENGINE, else "auto"): "feat" = shard_step.PreparedShardedStep (per-feature routing, the owner side is the
        single-GPU engine: bound forward AND bound row-sparse backward); "legacy" = PreparedShardedForward + the autograd training step;
        "auto" = "feat" whenever a table is row-sharded."""
        import os
        import bench
        feats, self.desc = bench.workload_spec(wl)
        self.rank, self.world, self.batch = rank, world, batch
        self.eng = RowShardedEmbedding(rank, world, overflow_policy="defer", host_staged=host_staged)
        gen = torch.Generator(device=device).manual_seed(seed)
        self.tables: Dict[str, torch.Tensor] = {}
        self.arenas: Dict[str, torch.Tensor] = {}      # row-sharded tables with their leading dummy row (shard_step.make_arena's layout)
        self.feats: List[ShardedFeature] = []
        self.fm = wl == "c2"
        self.n_replicated = self.n_sharded = 0
        rep_of: Dict[str, bool] = {}
        for f in sorted(feats, key=lambda f: f["name"]):
            tname = f.get("share", f["name"])
            if tname not in self.tables:
                rep = mode == "auto" and f["rows"] * f["dim"] * 4 <= replicate_below_bytes
                rep_of[tname] = rep
                nrows = f["rows"] if rep else local_row_count(f["rows"], rank, world)
                if rep:      # identical replica on every rank
                    t = torch.empty((nrows, f["dim"]), dtype=torch.float32, device=device)
                    t.normal_(generator=torch.Generator(device=device).manual_seed(seed - rank + len(self.tables)))
                else:        # the shard is rows 1.. of an arena whose row 0 is the dummy row the bound step's owner ids name with 0
                    arena = torch.empty((nrows + 1, f["dim"]), dtype=torch.float32, device=device)
                    arena[0].zero_()
                    t = arena[1:]
                    t.normal_(generator=gen)
                    self.arenas[tname] = arena
                if rank == 0 or rep:
                    t[0].zero_()
                self.tables[tname] = t
                self.n_replicated += rep
                self.n_sharded += not rep
            kind = NRX_BAG_MASKED_MEAN if f["bag"] else NRX_SPARSE
            self.feats.append(ShardedFeature(f["name"], kind, tname, f["dim"], f["bag"], False, self.fm, rep_of[tname]))
        self.rows = {f["name"]: f["rows"] for f in feats}
        self.pool = []
        for _ in range(n_pool):
            ins, ws = [], []
            for f in self.feats:
                shape = (batch, f.bag_len) if f.bag_len else (batch,)
                ins.append(torch.randint(1, self.rows[f.name], shape, device=device, generator=gen))
                ws.append(torch.ones(shape, dtype=torch.float32, device=device) if f.bag_len else None)
            self.pool.append((ins, ws))
        fdicts = [dict(dim=f.dim, bag=f.bag_len) for f in self.feats]
        self.bytes_per_impr = bench.algorithmic_bytes_per_impression(fdicts, self.fm, 0)
        self.bwd_bytes_per_impr = bench.backward_bytes_per_impression(fdicts, self.fm)
        self.desc += (f" -- {self.n_sharded} tables row-sharded over {world} GPUs, {self.n_replicated} replicated "
                      f"(mode={mode})")

        width = sum(f.dim for f in self.feats)
        out = torch.empty((batch, width), dtype=torch.float32, device=device)       # recycled (see bench.py)
        fmb = torch.empty((batch,), dtype=torch.float32, device=device) if self.fm else None
        engine = engine or os.environ.get("NRX_SHARD_ENGINE", "auto")
        if engine == "auto":
            engine = "feat" if self.n_sharded > 0 else "legacy"
        self.engine = engine
        if engine == "feat":
            from .shard_step import PreparedShardedStep
            tabs = {n: self.arenas.get(n, t) for n, t in self.tables.items()}
            # (the first two calls are bound in training form -- they also leave the FM field sums the backward folds in; train_setup binds their backward)
            # (binary_masks: the synthetic masks are all ones, as DataReader's are 0/1 -- NRX_SHARD_BINARY_MASKS=0 takes the general expansion)
            self.calls = [PreparedShardedStep(self.eng, self.feats, ins, ws, tabs, out=out, fm=fmb, train=(k < 2),
                                              binary_masks=os.environ.get("NRX_SHARD_BINARY_MASKS", "1") != "0")
                          for k, (ins, ws) in enumerate(self.pool)]
            return
        overlap = os.environ.get("NRX_SHARD_NO_OVERLAP") is None            # measurement knob: time the serial form
        self.calls = [PreparedShardedForward(self.eng, self.feats, ins, ws, self.tables, out=out, fm=fmb, overlap_local=overlap)
                      for ins, ws in self.pool]

    @torch.no_grad()
    def step(self, i: int):
        return self.calls[i % len(self.calls)].run()

    def overflowed(self) -> bool:
        return any(c.overflowed() for c in self.calls)

    def train_setup(self):
        """The TRAINING step of the sharded engine (what a sharded model's backward runs; bench.py's sharded fwd_bwd leg): the autograd form of
        the exchange -- forward as `step`, then the gradient all-to-all back to the owners and the owner-side scatter into the local shards'
        dense gradients (nrx_scatter_add_inbox; pooled bags: nrx_pool_inbox_bwd).  Returns False when the shards' dense gradients (one
        [local rows, dim] tensor per table, zero-filled every step -- the reference's nn.Embedding(sparse=False) semantics) do not fit."""
        dev = next(iter(self.tables.values())).device
        width = sum(f.dim for f in self.feats)
        if self.engine == "feat":
            # the BOUND step: shard_step.PreparedShardedStep.backward -- slot scatter of the upstream rows (nrx_embed_bwd_scatter), gradient all-to-all,
            # the owner-side planned reduction into row-sparse (keys, values): no dense shard gradient exists, so every workload fits
            gen = torch.Generator(device=dev).manual_seed(11)
            self._g_out = torch.randn((self.batch, width), device=dev, generator=gen)
            self._g_fm = torch.randn((self.batch,), device=dev, generator=gen) if self.fm else None
            for c in self.calls[:2]:
                c.bind_backward(self._g_out, self._g_fm)
            return True
        shard_bytes = sum(t.numel() * 4 for t in self.tables.values())
        if shard_bytes > (24 << 30):
            return False
        self._train_tables = {n: t.detach().requires_grad_(True) for n, t in self.tables.items()}
        width = sum(f.dim for f in self.feats)
        gen = torch.Generator(device=next(iter(self.tables.values())).device).manual_seed(11)
        self._g_out = torch.randn((self.batch, width), device=next(iter(self.tables.values())).device, generator=gen)
        self._g_fm = torch.randn((self.batch,), device=self._g_out.device, generator=gen) if self.fm else None
        return True

    def train_step(self, i: int):
        if self.engine == "feat":
            c = self.calls[i % 2]
            c.run()
            return c.backward()
        ins, ws = self.pool[i % len(self.pool)]
        out, _, fm = self.eng.forward(self.feats, ins, ws, self._train_tables)
        loss_like = [out]
        grads = [self._g_out]
        if fm is not None:
            loss_like.append(fm)
            grads.append(self._g_fm)
        torch.autograd.backward(loss_like, grads)
        for t in self._train_tables.values():
            t.grad = None

    def a2a_probe(self, steps: int):
        """Times the RETURN all-to-all(s) of one bound forward alone (rows / partial sums coming back from the owners):
        {"bytes": bytes received per rank per step, "ms": mean per step}, or None when nothing is row-sharded."""
        groups = self.calls[0].groups
        if not groups or self.world == 1:
            return None
        pairs = [(g["ret"], g["partial"] if g.get("pooled") else g["rows_out"]) for g in groups if not g.get("placed")]     # (both engines name them so)
        what = "the return all-to-all(s) of the row-sharded exchange alone (rows / partial sums coming back from the owners)"
        if not pairs:
            # one-sided placement: the rows do not come back through a collective at all (the owners write them into the requesters' buffers);
            # what crosses in collectives is the owner ids and the sample positions
            pairs = [(g["inbox"], g["send_ids"]) for g in groups if g.get("placed") and "send_ids" in g]
            pairs += [(g["inbox_pos"], g["send_pos"]) for g in groups if g.get("placed") and "send_pos" in g]
            what = ("one-sided placement: no row all-to-all exists (the owners write the rows into the requesters' mapped buffers); timed here: the id + "
                    "position all-to-alls of the exchange")
            if not pairs:
                return None
        nbytes = sum(r.numel() * r.element_size() for r, _ in pairs)
        for _ in range(3):
            for r, src in pairs:
                self.eng._a2a(r.view(-1), src.view(-1))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            for r, src in pairs:
                self.eng._a2a(r.view(-1), src.view(-1))
        b.record()
        torch.cuda.synchronize()
        return {"bytes": nbytes, "ms": a.elapsed_time(b) / steps, "what": what}
