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

Bag features (C4's history) keep sharding.py's pooled channel for the forward -- the owner pools the rows it holds, ONE partial row per (sample,
owner) comes back -- and get a bound backward here: the requester sends every owner the samples' upstream rows, `nrx_pool_inbox_expand` turns the
owner's inbox entries into pseudo-lookups (owner id, w * upstream row) of ONE single-valued feature over the arena, and the same planned
reduction takes over (row-sparse, deterministic; nrx_pool_inbox_bwd's float atomics into a dense shard gradient are gone).  A table fed by a
pooled group AND a single-valued group (DSSM's news table) leaves two (keys, values) lists, which FusedSparseAdam merges into one update per row."""
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
                 slack: Optional[float] = None, one_sided: Optional[bool] = None, binary_masks: bool = False, check_index: bool = False):
        """one_sided (default: NRX_SHARD_ONE_SIDED = 1 | 0, else on for plans without an FM epilogue): ONE-SIDED PLACEMENT of the forward -- the
        sample positions travel with the owner ids and the owner's gather (nrx_gather_place_feat) writes every row straight into its place in the
        requester's concat, which every rank maps once (hipIpc through torch's CUDA-IPC sharing; over xGMI a peer mapping): no row buffer, no
        row all-to-all, no final un-permuting launch for those features; a small collective behind the launches is the completion fence.
        The backward is the same either way (its slot map comes from the same routing launch).
        binary_masks: the caller's promise that every bag's non-zero weights are equal (DataReader's 0/1 masks,
        src/dataset/DataReader/data_reader.py:96-109; always true for mean pooling): the pooled channel's backward then never expands the owner's
        entries into rows -- the requester sends the sample gradients pre-multiplied by the sample's weight, the owner's plan lists those rows
        directly (nrx_sparse_plan_ex with NRX_PLAN_PAYLOAD) and the walk reads a [world * n * B, dim] block that stays in the L2 (the single-GPU bag backward's
        form).  False (default): the general form (nrx_pool_inbox_expand: any weights)."""
        import os
        self.lib = _lib.load()
        self.eng = eng
        self.feats = list(feats)
        self.binary_masks = bool(binary_masks)
        # check_index: the owners' launches record lookups that cannot be rows of their shard (an id outside its table: the reference's nn.Embedding
        # raises IndexError on the CPU, src/model/BaseModel/base_model.py:271) in a device status word; check() reads it, agrees over the ranks and raises
        self.status = torch.zeros(4, dtype=torch.int32, device=inputs[0].device) if check_index else None
        W = eng.world
        self.keep = [inputs, weights, arenas]
        groups, pooled = eng.plan_groups(feats)
        for gi, idxs in enumerate(groups):
            if gi not in pooled and any(feats[i].kind != NRX_SPARSE for i in idxs):
                raise NotImplementedError("PreparedShardedStep: row-sharded bag features travel through the pooled channel (pool_bags=True, no wide routing)")
        self.groups: List[dict] = []
        rets = []
        slot_of: Dict[int, torch.Tensor] = {}
        final_weights = list(weights)
        slack = eng.slack if slack is None else slack
        plan = eng._final_plan(feats, groups, pooled)
        B0 = inputs[0].shape[0]
        dev0 = inputs[0].device
        ld0 = int(out_ld) if out_ld else plan.out_width
        if one_sided is None:
            env = os.environ.get("NRX_SHARD_ONE_SIDED")
            one_sided = (env == "1") if env in ("0", "1") else True
        placed_groups = set()
        # an FM epilogue spans a sample's fields: with one-sided placement nobody holds them in one launch, so the logit (and, in training form, the
        # field sums) come from a pass over the FINISHED concat (nrx_fm_fwd_train) -- when every feature is an FM field of one width
        fm_ok = plan.use_fm and all(f.fm and f.kind == NRX_SPARSE and not f.replicated for f in feats) and len({f.dim for f in feats}) == 1 \
            and len(feats) <= ops.NRX_MAX_FEATURES
        if one_sided and (not plan.use_fm or fm_ok) and (ld0 & 3) == 0 and plan.wide_width == 0:
            for gi, idxs in enumerate(groups):
                D = feats[idxs[0]].dim
                if gi not in pooled and D in (16, 32, 64, 128, 256) and all(plan.slots[i].out_col % 4 == 0 for i in idxs):
                    placed_groups.add(gi)
            if plan.use_fm and len(placed_groups) != len(groups):
                placed_groups = set()
        self.peers = None
        if placed_groups:
            if out is None:
                out = torch.empty((B0, ld0), dtype=torch.float32, device=dev0)
            from .sharding import PeerMappingError, PreparedShardedForward
            try:
                self.peers = PreparedShardedForward._map_peer_buffers(eng, out)
                self._peer_ptrs = (C.c_void_p * W)(*[t.data_ptr() for t in self.peers])
                self._fence = (torch.zeros(W, dtype=torch.int32, device=dev0), torch.zeros(W, dtype=torch.int32, device=dev0))
            except PeerMappingError as e:          # (raised on every rank together: all take the buffered form)
                import warnings
                warnings.warn(f"PreparedShardedStep: one-sided placement is not available here, taking the all-to-all form of the forward ({e})")
                self.peers, placed_groups = None, set()
        for gi, idxs in enumerate(groups):
            dev = inputs[idxs[0]].device
            if gi in pooled:
                # bag features: the pooled channel of sharding.py for the forward (nrx_route_bags -> owner-side partial pooling -> one partial row per
                # (sample, owner) comes back); its backward is bound below (bind_backward): expansion into pseudo-lookups + the planned reduction
                from .sharding import PreparedShardedForward
                legacy = {t: arena_shard(a) for t, a in arenas.items()}
                g = PreparedShardedForward._bind_pooled(self, eng, feats, idxs, inputs, weights, legacy, C)
                tnames = sorted({feats[i].table for i in idxs})
                if len(tnames) != 1:
                    raise NotImplementedError("PreparedShardedStep: the bag features of one pooled exchange group must share ONE table")
                g.update(pooled=True, placed=False, idxs=list(idxs), tables=[arenas[tnames[0]]], table_names=tnames)
                if self.binary_masks and train:
                    g["inv"] = [torch.zeros(g["B"], dtype=torch.float32, device=dev) for _ in idxs]
                for k, i in enumerate(idxs):
                    slot_of[i] = eng._pooled_ids(g["B"], g["n"], k, dev)
                    final_weights[i] = None
                rets.append(g["ret"].view(-1, g["D"]))
                self.groups.append(g)
                continue
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
            g = dict(pooled=False, n=n, B=B, D=D, capf=capf, Bp=Bp, dev=dev, idxs=list(idxs), ids=ids, bits=ids[0].element_size() * 8, tables=tabs,
                     table_names=table_names, ptrs=(C.c_void_p * n)(*[x.data_ptr() for x in ids]),
                     send_ids=torch.zeros((W, n, capf), dtype=torch.int32, device=dev),
                     slot=torch.empty((n, B), dtype=torch.int32, device=dev),
                     counts=torch.zeros((W, n), dtype=torch.int64, device=dev), overflow=torch.zeros(1, dtype=torch.int64, device=dev),
                     state=torch.zeros(state_bytes, dtype=torch.uint8, device=dev))
            placed = gi in placed_groups
            g["placed"] = placed
            if placed:
                g["send_pos"] = torch.full((W, n, capf), -1, dtype=torch.int32, device=dev)
            if W == 1:        # a one-rank group exchanges with itself: the "received" buffers ARE the sent ones, [1][n][capf] is already [n][1 * capf]
                g["inbox"] = g["send_ids"]
                g["oid"] = g["send_ids"].view(n, Bp)
                if placed:
                    g["inbox_pos"] = g["send_pos"]
                    g["opos"] = g["send_pos"].view(n, Bp)
            else:
                g["inbox"] = torch.zeros((W, n, capf), dtype=torch.int32, device=dev)
                g["oid"] = torch.zeros((n, Bp), dtype=torch.int32, device=dev)
                if placed:
                    g["inbox_pos"] = torch.full((W, n, capf), -1, dtype=torch.int32, device=dev)
                    g["opos"] = torch.full((n, Bp), -1, dtype=torch.int32, device=dev)
            # the owner's side: a plain batch of Bp pseudo-samples, n single-valued features, concat [Bp, n * D]
            oslots = [ops.Slot(feats[i].name, NRX_SPARSE, table_names.index(feats[i].table), D, 0, k * D) for k, i in enumerate(idxs)]
            g["owner_fwd"] = ops.PreparedEmbed(ops.EmbedPlan(oslots, out_width=n * D), tabs, [g["oid"][k] for k in range(n)], [None] * n,
                                               need_out=not placed,      # (placed: never run -- the descriptor of the owner's pseudo-batch for the backward)
                                               check_index=check_index and not placed)
            for k, i in enumerate(idxs):
                slot_of[i] = g["slot"][k]
            if placed:
                ft = [table_names.index(feats[i].table) for i in idxs]
                g["tp"] = (C.c_void_p * n)(*[tabs[t].data_ptr() for t in ft])
                g["tr"] = (C.c_int64 * n)(*[tabs[t].shape[0] for t in ft])
                g["cols"] = (C.c_int32 * n)(*[plan.slots[i].out_col for i in idxs])
                rets.append(torch.empty((1, D), dtype=torch.float32, device=dev))      # (the final plan's table list keeps a never-read placeholder)
            else:
                g["rows_out"] = g["owner_fwd"].out                                               # [Bp, n * D]
                g["ret"] = g["rows_out"] if W == 1 else torch.empty_like(g["rows_out"])
                rets.append(g["ret"].view(-1, D))
            self.groups.append(g)
        # the requester's final launch: routed features read their returned rows by slot, replicated / dense features their own inputs; features
        # the owners placed are not its business
        rets += [arenas[t] for t in eng.replicated_tables(feats)]
        final_inputs = [inputs[i] if (f.kind == NRX_DENSE or f.replicated) else slot_of[i] for i, f in enumerate(feats)]
        self.fm_sums = None
        if train and plan.use_fm:
            self.fm_sums = torch.empty((B0, max(f.dim for f in feats if f.fm)), dtype=torch.float32, device=dev0)
        self.plan = plan
        placed_feats = {i for gi in placed_groups for i in groups[gi]}
        rest = [i for i in range(len(feats)) if i not in placed_feats]
        self.single = len(feats) <= ops.NRX_MAX_FEATURES
        self.ld = ld0
        self.fm_pass = None
        if not placed_feats:
            self.final = ops.PreparedEmbed(plan, rets, final_inputs, final_weights, out_ld=out_ld, out=out, fm=fm, fm_sums=self.fm_sums)
            self.out = self.final.out
        else:
            self.out = out
            self.final = None
            if plan.use_fm:
                self.fm_out = fm if fm is not None else torch.empty((B0,), dtype=torch.float32, device=dev0)
                self.fm_pass = (len(feats), feats[0].dim)
            if rest:
                sp = ops.EmbedPlan([plan.slots[i] for i in rest], out_width=plan.out_width, wide_width=0)
                self.final = ops.PreparedEmbed(sp, rets, [final_inputs[i] for i in rest], [final_weights[i] for i in rest], out_ld=ld0, out=out)
        self.bwd = None
        # where the owner-side plan (it depends on the owner ids only) is enqueued: "inline" = in backward(), behind the gradient exchange;
        # "backward" = on the planning side stream at the start of backward(), next to the pack launch and the gradient all-to-all;
        # "forward" = on the side stream right behind the id exchange of run() (what ops.PLAN_AHEAD does for the direct training step)
        self.plan_mode = os.environ.get("NRX_SHARD_PLAN", "inline")

    # ------------------------------------------------------------------ forward
    def _side_streams(self):
        """Exchange groups are independent of each other until the final launch (forward) / the returned lists (backward): with more than one --
        the DSSM tower: a single-valued group next to the pooled history bag -- the smaller groups' chains of short launches run on a side
        stream NEXT TO the largest group's (NRX_SHARD_OVERLAP=0: one after the other).  Returns (side stream or None, index of the group that
        stays on the current stream).  Forked and joined with wait_stream on both sides: capturable."""
        if getattr(self, "_overlap", None) is None:
            import os
            how = os.environ.get("NRX_SHARD_OVERLAP", "1")          # 1 (default) | fwd | bwd | 0
            # Launches whose blocks WAIT FOR EACH OTHER must not meet a second launch of that kind on the other stream unless both are sure to be
            # resident together (two half-started chains can hold each other's slots):
            #  * the one-kernel planner (plan_lds_kernel: one 158 KB-LDS block per compute unit, every block resident by construction) may be taken
            #    by single-valued groups only -- a pooled group's plan is always the sorted one -- so at most ONE single-valued group may exist;
            #  * the routing launches (chains over tiles): the forward forks only when the side groups' routing launches are small next to the
            #    device (<= one block per compute unit: they cannot fill an XCD, so the long chain's lowest waiting tile always finds its slot).
            #  * WORLD 1 ONLY.  The fork's gain (C4 404 -> 355 us) is a world-1 measurement -- at world > 1 every group's chain is cut by collectives
            #    anyway -- and nothing here can measure or fully validate two streams of collectives and one-sided writes on several GPUs
            #    (tests/stress_shard_step_multirank.py: world 2 green; at world 3, rank processes sharing one GPU, long sequences showed
            #    history-dependent failures that single towers do not reproduce: DESIGN.md section 9).  So the groups run one after the other there
            #    (NRX_SHARD_OVERLAP_WORLD=1 lifts the restriction for whoever takes this up).
            single = sum(1 for g in self.groups if not g["pooled"])
            world_ok = self.eng.world == 1 or os.environ.get("NRX_SHARD_OVERLAP_WORLD") == "1"
            self._overlap = how if (len(self.groups) > 1 and how != "0" and single <= 1 and world_ok) else ""
            self._side = torch.cuda.Stream(device=self.groups[0]["dev"]) if self._overlap else None
            size = [(self.eng.world * g["cap"] if g["pooled"] else g["n"] * g["B"]) for g in self.groups]
            self._main_group = max(range(len(size)), key=lambda i: size[i]) if size else 0
            tiles = [(sum(-(-g["B"] // max(1, 4096 // L)) for L in g["lens"]) if g["pooled"] else g["n"] * -(-g["B"] // 4096)) for g in self.groups]
            self._fwd_fork_ok = sum(t for i, t in enumerate(tiles) if i != self._main_group) <= 256
        return self._side, self._main_group

    def _each_group(self, fn, what="fwd"):
        """fn(index) for every exchange group: the largest on the current stream, the others on the side stream (fork before, join after)."""
        side, main = self._side_streams()
        # (measured, C4 at world 1: both directions forked 404 -> 355 us per training step; the forward alone gains nothing from its fork -- its short
        # group runs next to the ROUTING launch of the long one, a chain of waiting blocks, and slows it by what it saves -- so a forward-only
        # step does not fork)
        if side is None or self._overlap not in ("1", what) or (what == "fwd" and (not self._fwd_fork_ok or (self._overlap == "1" and self.bwd is None))):
            for gi in range(len(self.groups)):
                fn(gi)
            return
        cur = torch.cuda.current_stream(side.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for gi in range(len(self.groups)):
                if gi != main:
                    fn(gi)
        fn(main)
        cur.wait_stream(side)

    def run(self):
        eng, lib = self.eng, self.lib
        W = eng.world

        def one(gi):
            g = self.groups[gi]
            stream = torch.cuda.current_stream(g["dev"]).cuda_stream
            if g["pooled"]:
                from .sharding import PreparedShardedForward
                PreparedShardedForward._run_pooled(self, g, stream)
                return
            placed = g["placed"]
            rc = lib.nrx_route_feat(g["ptrs"], g["n"], g["B"], g["bits"], W, g["capf"], g["send_ids"].data_ptr(),
                                    g["send_pos"].data_ptr() if placed else None, g["slot"].data_ptr(),
                                    g["counts"].data_ptr(), g["overflow"].data_ptr(), g["state"].data_ptr(), stream)
            if rc:
                ops.check(rc, "nrx_route_feat")
            if W > 1:
                eng._a2a(g["inbox"].view(-1), g["send_ids"].view(-1))
                if placed:
                    eng._a2a(g["inbox_pos"].view(-1), g["send_pos"].view(-1))
                rc = lib.nrx_inbox_transpose(g["inbox"].data_ptr(), g["oid"].data_ptr(), g["inbox_pos"].data_ptr() if placed else None,
                                             g["opos"].data_ptr() if placed else None, W, g["n"], g["capf"], stream)
                if rc:
                    ops.check(rc, "nrx_inbox_transpose")
            if self.bwd is not None and self.plan_mode == "forward" and not g["pooled"] and not self.bwd[self.groups.index(g)]["direct"]:
                self.bwd[self.groups.index(g)]["owner"].plan_ahead()
            if placed:
                rc = lib.nrx_gather_place_feat(g["tp"], g["tr"], g["cols"], g["n"], W, g["capf"], g["oid"].data_ptr(), g["opos"].data_ptr(), g["D"],
                                               self._peer_ptrs, self.ld, self.out.shape[0], ops._ptr(self.status), stream)
                if rc:
                    ops.check(rc, "nrx_gather_place_feat")
                return
            g["owner_fwd"].run()
            if W > 1:
                eng._a2a(g["ret"].view(-1), g["rows_out"].view(-1))

        self._each_group(one)
        res = self.final.run() if self.final is not None else (self.out, None, None)
        if self.peers is not None and W > 1:
            # completion fence: a collective enqueued behind every rank's placing launch -- when it has completed here, every owner's rows are in this
            # rank's buffer (and the next step's first all-to-all keeps the owners from overwriting it too early)
            eng._a2a(self._fence[0], self._fence[1])
        if self.fm_pass is not None:
            nf, D = self.fm_pass
            stream = torch.cuda.current_stream(self.out.device).cuda_stream
            if self.fm_sums is not None:
                rc = lib.nrx_fm_fwd_train(self.out.data_ptr(), self.ld, nf, D, self.out.shape[0], self.fm_out.data_ptr(), self.fm_sums.data_ptr(),
                                          self.fm_sums.shape[1], stream)
            else:
                rc = lib.nrx_fm_fwd(self.out.data_ptr(), self.ld, nf, D, self.out.shape[0], self.fm_out.data_ptr(), stream)
            if rc:
                ops.check(rc, "nrx_fm_fwd")
            return (self.out, res[1], self.fm_out)
        return (self.out, res[1], res[2])

    def check(self):
        """Raise IndexError if a lookup of a run since the last check() named a row outside its table on ANY rank (every rank raises together: one
        small all-reduce + one host read), RuntimeError if a block overflowed its capacity (lookups were dropped: redo with a larger slack).
        Call it off the hot path -- shard_model_step_ does every 64th step."""
        bad = torch.zeros(2, dtype=torch.int64, device=self.groups[0]["dev"] if self.groups else "cpu")
        if self.status is not None:
            bad[0] = self.status[0].to(torch.int64)
        for g in self.groups:
            if not g["pooled"] and not g["placed"] and g["owner_fwd"].status is not None:
                bad[0] += g["owner_fwd"].status[0].to(torch.int64)
            over = (g["overflow"][0] > (g["cap"] if g["pooled"] else g["capf"])).to(torch.int64)
            bad[1] = torch.maximum(bad[1], over)
        self.eng._all_reduce_max(bad)
        n_bad, over = bad.tolist()
        if self.status is not None:
            self.status.zero_()
        for g in self.groups:
            g["overflow"].zero_()
            if not g["pooled"] and not g["placed"] and g["owner_fwd"].status is not None:
                g["owner_fwd"].status.zero_()
        if n_bad:
            raise IndexError("index out of range in self: a routed lookup named a row outside its table on some rank")
        if over:
            raise RuntimeError("PreparedShardedStep: an exchange block overflowed its capacity on some rank (lookups were dropped): ids too skewed for "
                               "this slack -- rebuild the step with a larger `slack`")

    def overflowed(self) -> bool:
        """True if some (owner, feature) block exceeded its capacity since the last call (one host read per group)."""
        bad = False
        for g in self.groups:
            bad |= int(g["overflow"].item()) > (g["cap"] if g["pooled"] else g["capf"])
            g["overflow"].zero_()
        return bad

    # ------------------------------------------------------------------ backward
    def bind_backward(self, g_out: Optional[torch.Tensor], g_fm: Optional[torch.Tensor] = None, direct_grad: Optional[bool] = None):
        """Bind the upstream gradient buffers (read in place on every backward()): g_out [B, out_ld] of the concat, g_fm [B] of the FM logit.
        direct_grad (default: NRX_SHARD_DIRECT_GRAD = 1 | 0, else on where it applies -- exchange groups of 16 / 32 / 64-wide single-valued
        features with aligned columns): THE REQUESTER'S PACK IS THE OWNER'S PLACEMENT PASS.  The owner plans first and its plan's dest[] comes
        back to the requesters (an int32 all-to-all in the ids' layout); a gradient row whose table row is looked up ONCE in the whole exchange
        is written by the requester straight into the owner's values[u] (the owners' gradient arenas are mapped once: hipIpc / peer mappings;
        nrx_embed_bwd_scatter_multi), the others into its block of the owner's receive buffer, and the owner reduces only the listed rows
        (nrx_embed_bwd_walk).  No gradient all-to-all, no owner-side placement pass: at world 1 the backward is the direct path's (plan, one
        placement pass, walk).  Same (keys, values) bit for bit as the buffered form."""
        import os
        lib, eng, plan = self.lib, self.eng, self.plan
        W = eng.world
        if direct_grad is None:
            direct_grad = getattr(self, "_direct_grad_default", None)
        if direct_grad is None:
            env = os.environ.get("NRX_SHARD_DIRECT_GRAD")
            direct_grad = env != "0"
        if not self.single:
            raise ValueError("PreparedShardedStep: the backward covers plans of <= 64 features")
        self.g_out = None if g_out is None else ops._f32c(g_out, "g_out")
        self.fmg = None
        if g_fm is not None:
            if self.fm_sums is None:
                raise ValueError("an FM gradient needs train=True on an FM plan")
            self.g_fm = ops._f32c(g_fm, "g_fm")
            self.fmg = NrxFmGrad(self.g_fm.data_ptr(), self.fm_sums.data_ptr(), self.fm_sums.shape[1], self.out.data_ptr(), self.ld)
        n_final_tables = len(self.groups) + len(eng.replicated_tables(self.feats))
        self.bwd = []
        for gi, g in enumerate(self.groups):
            if g["pooled"]:
                if self.g_out is None:
                    raise ValueError("PreparedShardedStep: a pooled bag group needs the gradient of the concat")
                n, D, B, cap = g["n"], g["D"], g["B"], g["cap"]
                arena = g["tables"][0]
                g_send = torch.empty((W, n * B, D), dtype=torch.float32, device=g["dev"])
                g_recv = g_send if W == 1 else torch.empty_like(g_send)
                oid = torch.zeros(W * cap, dtype=torch.int32, device=g["dev"])
                # the owner's pseudo-batch: ONE single-valued feature of W * cap pseudo-lookups over the arena (never run forward: a descriptor)
                pfwd = ops.PreparedEmbed(ops.EmbedPlan([ops.Slot(g["table_names"][0], NRX_SPARSE, 0, D, 0, 0)], out_width=D), [arena], [oid], [None],
                                         need_out=False)
                b = dict(pooled=True, direct=False, g_send=g_send, g_recv=g_recv, oid=oid, cols=[plan.slots[i].out_col for i in g["idxs"]], binary=self.binary_masks)
                if self.binary_masks:
                    # the upstream rows ARE the received block [W * n * B, D] (pre-scaled by the requester); order[] is rewritten to name its rows
                    b["rows"] = None
                    b["payload"] = torch.zeros(W * cap, dtype=torch.int32, device=g["dev"])
                    b["owner"] = ops.PreparedSparseBackward(pfwd, g_recv.view(-1, D), place_feats=0, payload=b["payload"])
                else:
                    b["rows"] = torch.zeros((W * cap, D), dtype=torch.float32, device=g["dev"])
                    b["owner"] = ops.PreparedSparseBackward(pfwd, b["rows"])
                b["owner"].groups[0]["arr"][0].flags |= _lib.NRX_FEAT_MANY_PER_ROW      # (~L lookups per row: the bag threshold of the work lists)
                self.bwd.append(b)
                continue
            n, D, Bp = g["n"], g["D"], g["Bp"]
            sub = ops.EmbedPlan([plan.slots[i] for i in g["idxs"]], out_width=plan.out_width, wide_width=plan.wide_width)
            slots_in = [g["slot"][k] for k in range(n)]
            # descriptors of the pack launch: the group's features as the final launch sees them (columns, FM flags); table = the send buffer (set below)
            arr = ops._fill_features(sub, 0, n, [None] * n_final_tables, slots_in, [None] * n, table_ptrs=[0] * n_final_tables,
                                     fm=self.fmg is not None)
            for k in range(n):
                arr[k].rows = Bp * n
            cap_v = n * Bp                                   # the owner's lookups = the worst-case number of unique rows
            shift = max(1, (cap_v + Bp * n - 1).bit_length())
            direct = bool(direct_grad) and D in (16, 32, 64) and (self.ld & 3) == 0 and all(plan.slots[i].out_col % 4 == 0 for i in g["idxs"]) \
                and self.g_out is not None and shift <= 30 and W <= (1 << (31 - shift)) and g["owner_fwd"].single
            if direct:
                # ONE arena per group on every rank: the owner's values rows first, its receive buffer [source][k][f] behind them
                arena = torch.zeros((cap_v + Bp * n, D), dtype=torch.float32, device=g["dev"])
                values, g_recv = arena[:cap_v], arena[cap_v:].view(Bp, n * D)
                owner_bwd = ops.PreparedSparseBackward(g["owner_fwd"], g_recv, values=values)
                og = owner_bwd.groups[0]
                if og["pmask"] is None:
                    direct = False
            if direct:
                from .sharding import PeerMappingError, PreparedShardedForward
                try:
                    peers = PreparedShardedForward._map_peer_buffers(eng, arena)
                except PeerMappingError as e:      # (raised on every rank together: all take the buffered form)
                    import warnings
                    warnings.warn(f"PreparedShardedStep: the gradient arenas cannot be mapped here, taking the all-to-all form of the backward ({e})")
                    direct = False
            if direct:
                b = dict(pooled=False, direct=True, arr=arr, owner=owner_bwd, arena=arena, peers=peers, shift=shift, cap_v=cap_v,
                         bases=(C.c_void_p * W)(*[t.data_ptr() for t in peers]), dest2=torch.empty(n * g["B"], dtype=torch.int32, device=g["dev"]),
                         fence=(torch.zeros(W, dtype=torch.int32, device=g["dev"]), torch.zeros(W, dtype=torch.int32, device=g["dev"])))
                if W > 1:       # the plan's dest[] back to the requesters: [f][s][k] -> [s][f][k] -> all-to-all -> [o][f][k]
                    b["dest_t"] = torch.empty((W, n, g["capf"]), dtype=torch.int32, device=g["dev"])
                    b["dest_req"] = torch.empty((W, n, g["capf"]), dtype=torch.int32, device=g["dev"])
                self.bwd.append(b)
                continue
            g_send = torch.zeros((Bp, n * D), dtype=torch.float32, device=g["dev"])
            for k in range(n):
                arr[k].table = g_send.data_ptr()
            g_recv = g_send if W == 1 else torch.empty_like(g_send)
            owner_bwd = ops.PreparedSparseBackward(g["owner_fwd"], g_recv)
            self.bwd.append(dict(pooled=False, direct=False, arr=arr, g_send=g_send, g_recv=g_recv, owner=owner_bwd, scatter_ok=True))
        return self

    def backward(self):
        """Enqueue the gradient exchange and the owner-side reduction.  Returns one entry per exchange group in ops.SparseGradSink's format --
        dict(tables (the arenas, index = table id in the keys), dim, uniq [cap] int64 keys table << 40 | arena row, values [cap, dim], counts,
        cap) -- valid until the next backward(); feed them to optim.FusedSparseAdam through `sink_entries`."""
        lib, eng = self.lib, self.eng
        W = eng.world
        if self.plan_mode == "backward":
            for b in self.bwd:
                if not b["pooled"] and not b["direct"]:
                    b["owner"].plan_ahead()
        outs = [[] for _ in self.groups]

        def one(gi):
            g, b, out = self.groups[gi], self.bwd[gi], outs[gi]
            stream = torch.cuda.current_stream(g["dev"]).cuda_stream
            if b["pooled"]:
                # d partial[o][tag] = the sample's upstream row for EVERY owner o (the requester adds the world partials): the same [n * B, D] block
                # goes to every owner; the owner turns its inbox entries into pseudo-lookups with upstream rows w * g and reduces them like any batch
                n, D, B = g["n"], g["D"], g["B"]
                for k, col in enumerate(b["cols"]):
                    scale = None
                    if b["binary"]:       # every live entry of a sample carries the same normalised weight (0 for an empty bag): the forward left it
                        scale = g["inv"][k] if g.get("inv") is not None else g["wn"][k].amax(dim=1).contiguous()
                    ops.check(lib.nrx_bag_upstream_rows(self.g_out.data_ptr(), self.ld, col, D, B, None if scale is None else scale.data_ptr(), W,
                                                        n * B * D, b["g_send"].data_ptr() + 4 * k * B * D, stream), "nrx_bag_upstream_rows")
                if W > 1:
                    eng._a2a(b["g_recv"].view(-1), b["g_send"].view(-1))
                runs = g.get("runs_state") is not None      # (the tags stayed at the source: the per-entry words come from the run bounds)
                if b["binary"] and runs:
                    rc = lib.nrx_pool_inbox_runs_words(g["tables"][0].shape[0] - 1, n, B, W, g["cap"], g["recv2d"].data_ptr(), g["inbox"].data_ptr(),
                                                       g["run"].data_ptr(), int(eng.rank == 0), None, b["oid"].data_ptr(), b["payload"].data_ptr(), stream)
                elif b["binary"]:
                    rc = lib.nrx_pool_inbox_owner_ids(g["tables"][0].shape[0] - 1, n, B, W, g["cap"], g["recv2d"].data_ptr(), g["inbox"].data_ptr(),
                                                      g["inbox_tag"].data_ptr(), int(eng.rank == 0), b["oid"].data_ptr(), b["payload"].data_ptr(), stream)
                else:
                    if runs:
                        ops.check(lib.nrx_pool_inbox_runs_words(g["tables"][0].shape[0] - 1, n, B, W, g["cap"], g["recv2d"].data_ptr(), g["inbox"].data_ptr(),
                                                                g["run"].data_ptr(), 0, g["inbox_tag"].data_ptr(), None, None, stream), "nrx_pool_inbox_runs_words")
                    rc = lib.nrx_pool_inbox_expand(g["tables"][0].shape[0] - 1, n, B, W, g["cap"], g["recv2d"].data_ptr(), g["inbox"].data_ptr(),
                                                   g["inbox_tag"].data_ptr(), g["inbox_w"].data_ptr(), D, b["g_recv"].data_ptr(), int(eng.rank == 0),
                                                   b["oid"].data_ptr(), b["rows"].data_ptr(), stream)
                if rc:
                    ops.check(rc, "nrx_pool_inbox_expand")
                for og in b["owner"].run():
                    out.append(dict(tables=g["tables"], dim=og["dim"], uniq=og["uniq"], values=og["values"], counts=og["counts"], cap=og["cap"],
                                    table_ids=[0]))
                return
            if b["direct"]:
                owner = b["owner"]
                owner.plan_only()
                dest_req = owner.groups[0]["dest"]
                if W > 1:
                    # (nrx_inbox_transpose with the roles of its two outer dimensions swapped: [f][s][k] -> [s][f][k])
                    ops.check(lib.nrx_inbox_transpose(dest_req.data_ptr(), b["dest_t"].data_ptr(), None, None, g["n"], W, g["capf"], stream), "nrx_inbox_transpose")
                    eng._a2a(b["dest_req"].view(-1), b["dest_t"].view(-1))
                    dest_req = b["dest_req"]
                ops.check(lib.nrx_shard_dest_combine(g["slot"].data_ptr(), dest_req.data_ptr(), g["n"], g["B"], g["capf"], b["cap_v"], eng.rank, W,
                                                     b["shift"], b["dest2"].data_ptr(), stream), "nrx_shard_dest_combine")
                if W == 1:      # one arena: the single-destination form (whole 128-byte lines per instruction for 64-byte rows), dest2 = the arena row
                    ops.check(lib.nrx_embed_bwd_scatter(b["arr"], g["n"], g["B"], g["D"], ops._ptr(self.g_out), self.ld, None, 0, self.fmg,
                                                        b["dest2"].data_ptr(), b["arena"].data_ptr(), stream), "nrx_embed_bwd_scatter")
                else:
                    ops.check(lib.nrx_embed_bwd_scatter_multi(b["arr"], g["n"], g["B"], g["D"], ops._ptr(self.g_out), self.ld, self.fmg, b["dest2"].data_ptr(),
                                                              b["bases"], W, b["shift"], stream), "nrx_embed_bwd_scatter_multi")
                if W > 1:       # completion fence: when this collective has completed here, every requester's rows are in this rank's arena
                    eng._a2a(b["fence"][0], b["fence"][1])
                for og in owner.run_walk():
                    out.append(dict(tables=g["tables"], dim=og["dim"], uniq=og["uniq"], values=og["values"], counts=og["counts"], cap=og["cap"],
                                    table_ids=list(range(len(g["tables"])))))
                return
            rc = NRX_ERR_UNSUPPORTED
            if b["scatter_ok"]:
                rc = lib.nrx_embed_bwd_scatter(b["arr"], g["n"], g["B"], g["D"], ops._ptr(self.g_out), self.ld, None, 0, self.fmg,
                                               g["slot"].data_ptr(), b["g_send"].data_ptr(), stream)
                if rc == NRX_ERR_UNSUPPORTED:
                    b["scatter_ok"] = False
                elif rc:
                    ops.check(rc, "nrx_embed_bwd_scatter")
            if rc == NRX_ERR_UNSUPPORTED:
                # outside the placement pass's shapes (odd dims, unaligned columns): the general kernel adds every lookup's row into the zero-filled
                # send buffer -- every slot is written by one lookup, so the float atomics have nothing to reorder
                b["g_send"].zero_()
                ops.check(lib.nrx_embed_bwd(b["arr"], g["n"], g["B"], ops._ptr(self.g_out), self.ld, None, 0, self.fmg, stream), "nrx_embed_bwd")
            if W > 1:
                eng._a2a(b["g_recv"].view(-1), b["g_send"].view(-1))
            for og in b["owner"].run():
                out.append(dict(tables=g["tables"], dim=og["dim"], uniq=og["uniq"], values=og["values"], counts=og["counts"], cap=og["cap"],
                                table_ids=list(range(len(g["tables"])))))

        self._each_group(one, "bwd")
        return [e for o in outs for e in o]

    def sink_entries(self, sink: "ops.SparseGradSink"):
        """backward() into a SparseGradSink (what optim.FusedSparseAdam drains)."""
        sink.pending.extend(self.backward())
        return sink


# --------------------------------------------------------------------------------- model integration
class _ShardedStepFn(torch.autograd.Function):
    """The bound step inside autograd: forward = step.run(); backward = the gradient exchange + the owner-side reduction, whose (keys, values)
    go to the model's SparseGradSink (what optim.FusedSparseAdam drains) -- the tables never see a dense .grad."""

    @staticmethod
    def forward(ctx, step, sink, scale, want_out, want_fm, anchor):
        out, _, fm = step.run()
        ctx.step, ctx.sink, ctx.scale = step, sink, scale
        ctx.set_materialize_grads(False)
        return (out if want_out else None), (fm if want_fm else None)

    @staticmethod
    def backward(ctx, g_out, g_fm):
        step = ctx.step
        if g_out is None and g_fm is None:
            return None, None, None, None, None, None
        if ctx.scale != 1.0:           # every rank's loss is a mean over ITS batch: the tables see the gradient of the global-batch mean
            g_out = None if g_out is None else g_out * ctx.scale
            g_fm = None if g_fm is None else g_fm * ctx.scale
        step.set_upstream(g_out, g_fm)
        ctx.sink.pending.extend(step.backward())
        return None, None, None, None, None, None


def _set_upstream(self, g_out, g_fm):
    """(Re)bind the upstream gradients of the next backward(): the first call allocates the exchange and plan buffers (bind_backward), later
    calls only swap the pointers the pack launch reads."""
    g_out = None if g_out is None else ops._f32c(g_out, "g_out")
    g_fm = None if g_fm is None else ops._f32c(g_fm, "g_fm")
    if self.bwd is None:
        self.bind_backward(g_out, g_fm)
        self._had = (g_out is not None, g_fm is not None)
        return
    if (g_out is not None, g_fm is not None) != self._had:
        raise RuntimeError("PreparedShardedStep: the set of upstream gradients changed between steps (bind a step per loss form)")
    self.g_out = g_out
    if g_fm is not None:
        self.g_fm = g_fm
        self.fmg = NrxFmGrad(g_fm.data_ptr(), self.fm_sums.data_ptr(), self.fm_sums.shape[1], self.out.data_ptr(), self.ld)


PreparedShardedStep.set_upstream = _set_upstream


def shard_model_step_(model, rank: int, world: int, group=None, host_staged: bool = False, slack: float = 0.05, binary_masks: bool = True,
                      one_sided: Optional[bool] = None, direct_grad: Optional[bool] = None, grad_average: bool = True):
    """Convert a BaseModel in place to row-sharded tables TRAINED BY THE BOUND STEP (the counterpart of sharding.shard_model_, whose backward
    forms dense shard gradients): every `embedding_tables[name].weight` becomes this rank's ARENA ([1 + local rows, D]: row 0 the dummy row,
    rows 1.. = global rows rank::world), `_embed` runs a PreparedShardedStep bound per (feature set, batch size) -- the batch's ids are copied into
    the step's own buffers -- and the backward leaves row-sparse (keys, values) in `model._sparse_sink`; `configure_optimizers()` then builds
    SparseDenseAdam over the arenas (FusedSparseAdam on the looked-up rows) + AdamW for the dense parameters, which the caller all-reduces
    (sharding.allreduce_dense_grads(sharding.data_parallel_params(model), world)).  state_dict keys are unchanged; the table values are the
    arenas (full_state_dict / load_full_state_dict_ convert to and from the reference's full tables).
    binary_masks: the array features' masks are 0/1 (DataReader's, src/dataset/DataReader/data_reader.py:96-109).
    grad_average: scale the table gradients by 1 / world (every rank's loss is a mean over its own batch)."""
    import torch.nn as nn
    from .sharding import RowShardedEmbedding, ShardedFeature
    eng = RowShardedEmbedding(rank, world, group, None, slack=slack, host_staged=host_staged, overflow_policy="defer")
    model.sparse_grad = "fused"
    if getattr(model, "_sparse_sink", None) is None:
        model._sparse_sink = ops.SparseGradSink()
    model._replicated_tables = ()
    for name, emb in list(model.embedding_tables.items()):
        w = emb.weight.data
        arena = make_arena(emb.num_embeddings, w.shape[1], rank, world, w.device, full=w)
        new = nn.Embedding(arena.shape[0], arena.shape[1])
        new.weight = nn.Parameter(arena, requires_grad=False)      # (updated in place by FusedSparseAdam from the sink, never through .grad)
        new.global_rows = emb.num_embeddings
        new.arena = True
        model.embedding_tables[name] = new
    model._shard_engine = eng
    model._shard_steps = {}
    scale = 1.0 / world if (grad_average and world > 1) else 1.0

    def _embed_step(batch, feature_names, fm=False, wide_names=(), out_ld=None, need_out=True):
        plan, table_names, dims, present = model._plan(batch, feature_names, fm, wide_names)
        if not present:
            return None, None, None, [], []
        if wide_names:
            raise NotImplementedError("shard_model_step_: Wide&Deep column routing is not bound (sharding.shard_model_ serves it)")
        if out_ld is not None and out_ld < 0:
            out_ld = None
        names = [s.name for s in plan.slots]
        masks = [f"{s.name}_mask" if s.kind == ops.NRX_BAG_MASKED_MEAN else None for s in plan.slots]
        B = batch[names[0]].shape[0]
        key = (tuple(names), B, bool(fm), out_ld, tuple(str(batch[n].dtype) for n in names))
        ent = model._shard_steps.get(key)
        if ent is None:
            feats = [ShardedFeature(s.name, s.kind, '' if s.kind == NRX_DENSE else table_names[s.table], s.dim, s.bag_len, False, bool(s.fm_field))
                     for s in plan.slots]
            bufs = [batch[n].detach().clone().contiguous() for n in names]
            wbufs = [None if m is None else batch[m].detach().clone().contiguous() for m in masks]
            arenas = {t: model.embedding_tables[t].weight.data for t in table_names}
            step = PreparedShardedStep(eng, feats, bufs, wbufs, arenas, out_ld=out_ld, train=True, slack=slack, one_sided=one_sided,
                                       binary_masks=binary_masks, check_index=getattr(model, "index_check", "deferred") != "off")
            step._calls = 0
            step._direct_grad_default = direct_grad
            anchor = torch.zeros(1, device=bufs[0].device, requires_grad=True)
            ent = model._shard_steps[key] = (step, bufs, wbufs, anchor)
        step, bufs, wbufs, anchor = ent
        for b_, n in zip(bufs, names):
            b_.copy_(batch[n])
        for w_, m in zip(wbufs, masks):
            if w_ is not None:
                w_.copy_(batch[m])
        step._calls += 1
        if step._calls % 64 == 1 and step._calls > 1:       # deferred: out-of-range ids / dropped lookups of the last 64 steps surface here, on every rank
            step.check()
        if torch.is_grad_enabled():
            out, fmv = _ShardedStepFn.apply(step, model._sparse_sink, scale, need_out, bool(fm), anchor)
        else:
            o, _, f_ = step.run()
            out, fmv = (o if need_out else None), (f_ if fm else None)
        return out, None, fmv, list(dims), list(present)

    model._embed = _embed_step
    return model
