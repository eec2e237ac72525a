"""The bound training step of the row-sharded engine (round 6): forward exchange + gradient exchange + OWNER-SIDE ROW-SPARSE REDUCTION, every
buffer allocated once, nothing read back to the host, no dense shard gradient, no float atomic.

New in this build: the reference is single-device (every trainer is `devices=1`, src/model/sort/deep/train.py:38-44); what this layer replaces
on each shard is autograd of `BaseModel.get_embeddings_from_batch` (src/model/BaseModel/base_model.py:262-308) followed by the table part of
`configure_optimizers` (src/model/sort/deep/model.py:54-65).

Idea: make the owner's side of the exchange a plain single-GPU batch, so that it IS the single-GPU engine.  `nrx_route_feat` gives every
(source, owner, feature) triple its own `capf` slots; after one equal-split all-to-all the owner holds, per feature, ONE array of
`world * capf` OWNER IDS (0 = nothing, v = local row v - 1) -- a batch of B' = world * capf pseudo-samples over shard tables that carry a
leading dummy row (an "arena": row 0 reads zeros and never trains, exactly the padding row of a single-GPU table).  Then

  forward   requester: route (one launch) -> all-to-all ids -> OWNER: the fused forward (`nrx_embed_fwd`, the same launch as the direct path)
            writes the [B', n * D] concat of its rows -> all-to-all rows -> requester: the fused final launch un-permutes by `slot[]` into the
            [B, sum D] concat (+ FM epilogue)
  backward  requester: `nrx_embed_bwd_scatter` (the placement pass with dest = slot[]: every lookup's upstream row, FM term folded in, goes
            to its place in the [B', n * D] gradient send buffer -- a permutation) -> all-to-all -> OWNER: `PreparedSparseBackward` on the
            pseudo-batch: the planners (one-kernel LDS-bitmap or sorted), the placement pass, the sorted walk, the work lists -> (keys, values,
            counts) of the rows this rank owns, bit-reproducible -> `FusedSparseAdam` on the arenas.

With world == 1 the all-to-alls vanish (the receive buffers ARE the send buffers) and the owner-side plan is the direct path's plan: keys and
values equal the unsharded row-sparse gradient bit for bit (keys shifted by the dummy row).  With world > 1 an owner reduces a row's lookups
in (feature, source rank, sample) order -- the order of the unsharded reduction over the rank-major concatenation of the batches.

Scope: exchange groups of single-valued features (the C2 / C3 / C5 shapes) and planner-replicated / dense features next to them; row-sharded
bag features keep `sharding.RowShardedEmbedding`'s pooled channel."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, ops
from ._lib import NRX_DENSE, NRX_ERR_UNSUPPORTED, NRX_SPARSE, NrxFmGrad
from .sharding import RowShardedEmbedding, ShardedFeature, local_row_count


# --------------------------------------------------------------------------------- arenas
def make_arena(rows: int, dim: int, rank: int, world: int, device, full: Optional[torch.Tensor] = None,
               generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """This rank's shard of a [rows, dim] table WITH the leading dummy row: [1 + local_row_count, dim]; arena[1 + k] = global row
    k * world + rank.  `full` (optional): the unsharded table to take the rows from (scatter-on-load); else N(0, 1) rows (nn.Embedding's init,
    base_model.py:164).  arena[0] is zero and stays zero; on rank 0, arena[1] is the global padding row: zero, never looked up (owner id 0
    stands for it), never trained."""
    n = local_row_count(rows, rank, world)
    a = torch.empty((n + 1, dim), dtype=torch.float32, device=device)
    if full is not None:
        a[1:].copy_(full[rank::world])
    else:
        a[1:].normal_(generator=generator)
    a[0].zero_()
    if rank == 0 and n > 0:
        a[1].zero_()
    return a


def arena_shard(arena: torch.Tensor) -> torch.Tensor:
    """The legacy view of an arena: [local rows, dim] = global rows rank::world (what shard_table / the checkpoints use)."""
    return arena[1:]


class PreparedShardedStep:
    """A bound, re-launchable sharded forward + backward over the SAME id tensors (refill them in place): see the module docstring.
    `arenas` maps table name -> arena tensor (make_arena).  Replicated tables (ShardedFeature.replicated) are plain full tables."""

    def __init__(self, eng: RowShardedEmbedding, feats: Sequence[ShardedFeature], inputs, weights, arenas: Dict[str, torch.Tensor],
                 out_ld: Optional[int] = None, out: Optional[torch.Tensor] = None, fm: Optional[torch.Tensor] = None, train: bool = True,
                 slack: Optional[float] = None):
        self.lib = _lib.load()
        self.eng = eng
        self.feats = list(feats)
        W = eng.world
        self.keep = [inputs, weights, arenas]
        groups, pooled = eng.plan_groups(feats)
        if pooled or any(f.kind not in (NRX_SPARSE, NRX_DENSE) and not f.replicated for f in feats):
            raise NotImplementedError("PreparedShardedStep: row-sharded bag features are not bound here (RowShardedEmbedding pools them at the owner)")
        self.groups: List[dict] = []
        rets = []
        slot_of: Dict[int, torch.Tensor] = {}
        slack = eng.slack if slack is None else slack
        for idxs in groups:
            dev = inputs[idxs[0]].device
            ids = [inputs[i] for i in idxs]
            dt = ids[0].dtype
            if dt not in (torch.int32, torch.int64) or any(x.dtype != dt or not x.is_contiguous() or x.dim() != 1 for x in ids):
                raise TypeError("PreparedShardedStep: the ids of one exchange group must be contiguous 1-D tensors of one dtype (int32 or int64)")
            n, B, D = len(ids), ids[0].numel(), feats[idxs[0]].dim
            if any(x.numel() != B for x in ids):
                raise ValueError("PreparedShardedStep: the features of one exchange group must share the batch size")
            capf = B if W == 1 else (int(B / W * (1.0 + slack)) + 64 + 63) // 64 * 64
            Bp = W * capf
            table_names: List[str] = []
            for i in idxs:
                if feats[i].table not in table_names:
                    table_names.append(feats[i].table)
            tabs = [arenas[t] for t in table_names]
            state_bytes = self.lib.nrx_route_feat_state_bytes(n, B, W)
            if state_bytes < 0:
                raise ValueError("PreparedShardedStep: group outside nrx_route_feat's limits")
            g = dict(n=n, B=B, D=D, capf=capf, Bp=Bp, dev=dev, idxs=list(idxs), ids=ids, bits=ids[0].element_size() * 8, tables=tabs,
                     table_names=table_names, ptrs=(C.c_void_p * n)(*[x.data_ptr() for x in ids]),
                     send_ids=torch.zeros((W, n, capf), dtype=torch.int32, device=dev),
                     slot=torch.empty((n, B), dtype=torch.int32, device=dev),
                     counts=torch.zeros((W, n), dtype=torch.int64, device=dev), overflow=torch.zeros(1, dtype=torch.int64, device=dev),
                     state=torch.zeros(state_bytes, dtype=torch.uint8, device=dev))
            if W == 1:        # a one-rank group exchanges with itself: the "received" buffers ARE the sent ones, [1][n][capf] is already [n][1 * capf]
                g["inbox"] = g["send_ids"]
                g["oid"] = g["send_ids"].view(n, Bp)
            else:
                g["inbox"] = torch.zeros((W, n, capf), dtype=torch.int32, device=dev)
                g["oid"] = torch.zeros((n, Bp), dtype=torch.int32, device=dev)
            # the owner's side: a plain batch of Bp pseudo-samples, n single-valued features, concat [Bp, n * D]
            oslots = [ops.Slot(feats[i].name, NRX_SPARSE, table_names.index(feats[i].table), D, 0, k * D) for k, i in enumerate(idxs)]
            g["owner_fwd"] = ops.PreparedEmbed(ops.EmbedPlan(oslots, out_width=n * D), tabs, [g["oid"][k] for k in range(n)], [None] * n)
            g["rows_out"] = g["owner_fwd"].out                                                   # [Bp, n * D]
            g["ret"] = g["rows_out"] if W == 1 else torch.empty_like(g["rows_out"])
            for k, i in enumerate(idxs):
                slot_of[i] = g["slot"][k]
            rets.append(g["ret"].view(-1, D))
            self.groups.append(g)
        # the requester's final launch: routed features read their returned rows by slot, replicated / dense features their own inputs
        plan = eng._final_plan(feats, groups, ())
        rets += [arenas[t] for t in eng.replicated_tables(feats)]
        final_inputs = [inputs[i] if (f.kind == NRX_DENSE or f.replicated) else slot_of[i] for i, f in enumerate(feats)]
        B0 = inputs[0].shape[0]
        dev0 = inputs[0].device
        self.fm_sums = None
        if train and plan.use_fm:
            self.fm_sums = torch.empty((B0, max(f.dim for f in feats if f.fm)), dtype=torch.float32, device=dev0)
        self.plan = plan
        self.final = ops.PreparedEmbed(plan, rets, final_inputs, list(weights), out_ld=out_ld, out=out, fm=fm, fm_sums=self.fm_sums)
        self.bwd = None

    # ------------------------------------------------------------------ forward
    def run(self):
        eng, lib = self.eng, self.lib
        W = eng.world
        for g in self.groups:
            stream = torch.cuda.current_stream(g["dev"]).cuda_stream
            rc = lib.nrx_route_feat(g["ptrs"], g["n"], g["B"], g["bits"], W, g["capf"], g["send_ids"].data_ptr(), None, g["slot"].data_ptr(),
                                    g["counts"].data_ptr(), g["overflow"].data_ptr(), g["state"].data_ptr(), stream)
            if rc:
                ops.check(rc, "nrx_route_feat")
            if W > 1:
                eng._a2a(g["inbox"].view(-1), g["send_ids"].view(-1))
                rc = lib.nrx_inbox_transpose(g["inbox"].data_ptr(), g["oid"].data_ptr(), None, None, W, g["n"], g["capf"], stream)
                if rc:
                    ops.check(rc, "nrx_inbox_transpose")
            g["owner_fwd"].run()
            if W > 1:
                eng._a2a(g["ret"].view(-1), g["rows_out"].view(-1))
        return self.final.run()

    def overflowed(self) -> bool:
        """True if some (owner, feature) block exceeded its capacity since the last call (one host read per group)."""
        bad = False
        for g in self.groups:
            bad |= int(g["overflow"].item()) > g["capf"]
            g["overflow"].zero_()
        return bad

    # ------------------------------------------------------------------ backward
    def bind_backward(self, g_out: Optional[torch.Tensor], g_fm: Optional[torch.Tensor] = None):
        """Bind the upstream gradient buffers (read in place on every backward()): g_out [B, out_ld] of the concat, g_fm [B] of the FM logit."""
        lib, eng, plan = self.lib, self.eng, self.plan
        W = eng.world
        fwd = self.final
        if not fwd.single:
            raise ValueError("PreparedShardedStep: the backward covers plans of <= 64 features")
        self.g_out = None if g_out is None else ops._f32c(g_out, "g_out")
        self.fmg = None
        if g_fm is not None:
            if self.fm_sums is None:
                raise ValueError("an FM gradient needs train=True on an FM plan")
            self.g_fm = ops._f32c(g_fm, "g_fm")
            self.fmg = NrxFmGrad(self.g_fm.data_ptr(), self.fm_sums.data_ptr(), self.fm_sums.shape[1], fwd.out.data_ptr(), fwd.ld)
        n_final_tables = len(fwd.tables)
        self.bwd = []
        for gi, g in enumerate(self.groups):
            n, D, Bp = g["n"], g["D"], g["Bp"]
            sub = ops.EmbedPlan([plan.slots[i] for i in g["idxs"]], out_width=plan.out_width, wide_width=plan.wide_width)
            slots_in = [g["slot"][k] for k in range(n)]
            g_send = torch.zeros((Bp, n * D), dtype=torch.float32, device=g["dev"])
            # descriptors of the pack launch: the group's features as the final launch sees them (columns, FM flags); table = the send buffer
            arr = ops._fill_features(sub, 0, n, [None] * n_final_tables, slots_in, [None] * n, table_ptrs=[g_send.data_ptr()] * n_final_tables,
                                     fm=self.fmg is not None)
            for k in range(n):
                arr[k].rows = Bp * n
            g_recv = g_send if W == 1 else torch.empty_like(g_send)
            owner_bwd = ops.PreparedSparseBackward(g["owner_fwd"], g_recv)
            self.bwd.append(dict(arr=arr, g_send=g_send, g_recv=g_recv, owner=owner_bwd, scatter_ok=True))
        return self

    def backward(self):
        """Enqueue the gradient exchange and the owner-side reduction.  Returns one entry per exchange group in ops.SparseGradSink's format --
        dict(tables (the arenas, index = table id in the keys), dim, uniq [cap] int64 keys table << 40 | arena row, values [cap, dim], counts,
        cap) -- valid until the next backward(); feed them to optim.FusedSparseAdam through `sink_entries`."""
        lib, eng = self.lib, self.eng
        W = eng.world
        fwd = self.final
        out = []
        for g, b in zip(self.groups, self.bwd):
            stream = torch.cuda.current_stream(g["dev"]).cuda_stream
            rc = NRX_ERR_UNSUPPORTED
            if b["scatter_ok"]:
                rc = lib.nrx_embed_bwd_scatter(b["arr"], g["n"], g["B"], g["D"], ops._ptr(self.g_out), fwd.ld, None, 0, self.fmg,
                                               g["slot"].data_ptr(), b["g_send"].data_ptr(), stream)
                if rc == NRX_ERR_UNSUPPORTED:
                    b["scatter_ok"] = False
                elif rc:
                    ops.check(rc, "nrx_embed_bwd_scatter")
            if rc == NRX_ERR_UNSUPPORTED:
                # outside the placement pass's shapes (odd dims, unaligned columns): the general kernel adds every lookup's row into the zero-filled
                # send buffer -- every slot is written by one lookup, so the float atomics have nothing to reorder
                b["g_send"].zero_()
                ops.check(lib.nrx_embed_bwd(b["arr"], g["n"], g["B"], ops._ptr(self.g_out), fwd.ld, None, 0, self.fmg, stream), "nrx_embed_bwd")
            if W > 1:
                eng._a2a(b["g_recv"].view(-1), b["g_send"].view(-1))
            for og in b["owner"].run():
                out.append(dict(tables=g["tables"], dim=og["dim"], uniq=og["uniq"], values=og["values"], counts=og["counts"], cap=og["cap"],
                                table_ids=list(range(len(g["tables"])))))
        return out

    def sink_entries(self, sink: "ops.SparseGradSink"):
        """backward() into a SparseGradSink (what optim.FusedSparseAdam drains)."""
        sink.pending.extend(self.backward())
        return sink
