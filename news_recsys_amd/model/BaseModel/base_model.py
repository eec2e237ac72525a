"""BaseModel -- the drop-in boundary of the path.

Mirror of the reference's `src/model/BaseModel/base_model.py` *embedding half* (:35-166, :262-308):
same constructor `(config_path)`, same attributes (`embedding_tables`, `*_feature_names` sets,
`embedding_size`, `embedding_table_size`, `share_emb_table_features`, `array_max_length`,
`item_input_dim`, `user_input_dim`, `train_hparams`, `config`), same methods and return types, same
state_dict keys (`embedding_tables.<emb_name>.weight`, fp32 [rows, dim], row 0 = padding).

What is different underneath: `get_embeddings_from_batch` is ONE fused HIP launch (gather + masked
mean-pool + concat written in place; optional Wide&Deep column routing and FM epilogue) instead of a
Python loop of nn.Embedding / mul / sum / div / torch.cat kernels, and there is no eager fallback:
on a CPU tensor the call raises.

Documented deviations from the reference (SURVEY facts 4, 5 and hard part 5):
  * feature order is `sorted(...)` everywhere, including table construction (the reference iterates
    Python sets there, so its init RNG order depends on PYTHONHASHSEED);
  * `dense_feature_dim` exists (default 1); the reference reads it but never sets it (:94 vs :129);
  * the third return value of `get_embeddings_from_batch` lists only the features actually present in
    the batch, so it always lines up with `dims` (the reference returns the unfiltered list, :308).
The validation half (base_model.py:320-528) is rebuilt on the device (news_recsys_amd/metrics.py); the remaining
trainer glue (train.log / model_info.log, :215-256) is outside the path.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, List, Optional, Sequence, Set, Tuple

import torch
import torch.nn as nn

from ... import ops
from ..._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_DENSE, NRX_FEAT_BAG_CSR, NRX_SPARSE
from ...config import load_config, to_container
from ...lightning_shim import LightningModule

logger = logging.getLogger("BaseModel")


class BaseModel(LightningModule):
    def __init__(self, config_path: str):
        super().__init__()
        self._load_config(config_path)
        self._validate_config()
        self.item_input_dim = self._calculate_input_dim(self.item_feature_names)
        self.user_input_dim = self._calculate_input_dim(self.user_feature_names)
        logger.info("Input Dimensions - Item: %d, User: %d", self.item_input_dim, self.user_input_dim)
        self.embedding_tables = self._build_embedding_tables()
        if self.user_history_path and os.path.exists(self.user_history_path):
            with open(self.user_history_path, "r") as f:
                self.user_history = json.load(f)
        self._init_metrics_state()
        self.save_hyperparameters(to_container(self.config))
        self.feature_id_mapper = None
        self._plan_cache: Dict[tuple, Tuple[ops.EmbedPlan, List[str], List[int], List[str]]] = {}
        self._embed_cache: Dict[tuple, tuple] = {}
        self._first_table: Dict[tuple, str] = {}
        self._none_lists: Dict[int, list] = {}

    # ------------------------------------------------------------------ config (base_model.py:69-139)
    def _load_config(self, config_path: str) -> None:
        self.config = load_config(config_path)
        paths_cfg = self.config.get("paths", {}) or {}
        self.out_basedir: str = paths_cfg.get("out_basedir", "")
        self.user_history_path: str = paths_cfg.get("user_history_path", "")

        f = self.config.get("features", {}) or {}
        self.sparse_feature_names: Set[str] = set(f.get("sparse_feature_names") or [])
        self.dense_feature_names: Set[str] = set(f.get("dense_feature_names") or [])
        self.array_feature_names: Set[str] = set(f.get("array_feature_names") or [])
        self.item_feature_names: Set[str] = set(f.get("item_feature_names") or [])
        self.user_feature_names: Set[str] = set(f.get("user_feature_names") or [])
        self.array_max_length: Dict[str, int] = dict(f.get("array_max_length") or {})
        self.dense_feature_dim: int = int(f.get("dense_feature_dim", 1))

        e = self.config.get("embeddings", {}) or {}
        self.embedding_size: Dict[str, int] = dict(e.get("embedding_size") or {})
        self.embedding_table_size: Dict[str, int] = dict(e.get("embedding_table_size") or {})
        self.share_emb_table_features: Dict[str, str] = dict(e.get("share_emb_table_features") or {})
        # new optional key (default = reference behaviour: dense weight.grad): deterministic row-sparse
        # table gradients, for tables too large to zero-fill / dense-update every step
        # "fused": the same reduction stays on the device and optim.FusedSparseAdam updates the touched rows in
        # one launch (no COO tensors, no host read per step)
        # "exact": the sink again, drained by optim.ExactDenseAdamW -- the reference's dense AdamW over EVERY row (weight decay, decaying
        # moments: the same numbers as the default mode) streamed once per step, without a dense gradient tensor in between
        sg = e.get("sparse_grad", False)
        self.sparse_grad = str(sg).lower() if str(sg).lower() in ("fused", "exact") else bool(sg)
        self._sparse_sink = None
        # new optional key: how an out-of-range id surfaces on the fused batch path (get_embeddings_from_batch / forward).
        # "deferred" (default): no device synchronisation -- the kernel records the offence in a host-mapped status word
        # and the NEXT call (or ops.flush_index_checks(), called at the end of every validation epoch) raises IndexError;
        # "sync": IndexError in the offending call, as torch does on CPU (one device sync per call); "off": never checked.
        # get_feature_embedding (the per-feature reference API) always raises in the offending call.
        self.index_check = str(e.get("index_check", "deferred")).lower()
        if self.index_check not in ("deferred", "sync", "lazy", "off"):
            raise ValueError("embeddings.index_check must be 'deferred', 'sync', 'lazy' or 'off'")

        self.dataset_cfg = self.config.get("dataset", {}) or {}
        self.train_hparams = self.config.get("train_hparams", {}) or {}

    def _validate_config(self) -> None:
        if not self.out_basedir:
            logger.warning("out_basedir is not set in config.")
        for fname in sorted(self.sparse_feature_names):
            emb_name = self._get_emb_feature_name(fname)
            if emb_name not in self.embedding_size:
                logger.warning("Embedding size for feature '%s' (from '%s') is missing!", emb_name, fname)

    def _get_emb_feature_name(self, feature_name: str) -> str:
        return self.share_emb_table_features.get(feature_name, feature_name)

    def _calculate_input_dim(self, feature_names) -> int:
        total = 0
        for fname in sorted(feature_names):
            if fname in self.dense_feature_names:
                total += self.dense_feature_dim
                continue
            dim = self.embedding_size.get(self._get_emb_feature_name(fname))
            if dim is None:
                logger.warning("Feature '%s' has no embedding size config. Using default 8.", fname)
                dim = 8
            total += dim
        return total

    # ------------------------------------------------------------------ tables (base_model.py:141-166)
    def _build_embedding_tables(self) -> nn.ModuleDict:
        tables = nn.ModuleDict()
        for fname in sorted(self.sparse_feature_names | self.array_feature_names):
            emb_fname = self._get_emb_feature_name(fname)
            if emb_fname in tables:
                continue
            size = self.embedding_table_size.get(emb_fname)
            dim = self.embedding_size.get(emb_fname)
            if size is None or dim is None:
                logger.error("Missing embedding config (size/dim) for feature: %s", emb_fname)
                continue
            # nn.Embedding is kept as the parameter container (state_dict key, N(0,1) init, zero padding
            # row); its forward() is never used -- lookups go through the HIP kernels.
            tables[emb_fname] = nn.Embedding(size, dim, padding_idx=0)
        return tables

    def _init_metrics_state(self) -> None:
        self.best_metrics = {"AUC": 0.0, "LogLoss": float("inf"), "GAUC": 0.0, "HR@10": 0.0, "NDCG@10": 0.0,
                             "MRR@10": 0.0, "Step": -1}
        self.user_scores_dict = {}
        self._val_uid, self._val_score, self._val_label = [], [], []

    # ------------------------------------------------------------------ lookups (base_model.py:262-308)
    def _table_weight(self, feature_name: str) -> torch.Tensor:
        emb_fname = self._get_emb_feature_name(feature_name)
        if emb_fname not in self.embedding_tables:
            raise ValueError(f"Embedding table not found for {feature_name} (mapped to {emb_fname})")
        return self.embedding_tables[emb_fname].weight

    def get_feature_embedding(self, feature_name: str, feature_value: torch.Tensor) -> torch.Tensor:
        """Dense: value.float().unsqueeze(1).  Else a row gather: ids [B] -> [B, D], ids [B, L] -> [B, L, D]."""
        if feature_name in self.dense_feature_names:
            return feature_value.float().unsqueeze(1)
        weight = self._table_weight(feature_name)
        D = weight.shape[1]
        flat = feature_value.reshape(-1)
        plan = ops.EmbedPlan([ops.Slot(feature_name, NRX_SPARSE, 0, D, 0, 0)], out_width=D)
        out = ops.embed_apply(plan, [weight], [flat], [None], index_check="sync")[0]
        return out.view(*feature_value.shape, D)

    def array_feature_pooling(self, embedding: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.bag_pool(embedding, mask)

    def _plan(self, batch: Dict[str, torch.Tensor], feature_names, fm: bool, wide_names: Sequence[str]):
        present = [n for n in sorted(feature_names) if n in batch]
        key_items = []
        for n in present:
            if n in self.dense_feature_names:
                key_items.append((n, NRX_DENSE, 0))
            elif n in self.array_feature_names:
                if batch.get(f"{n}_offsets") is not None:
                    # CSR batch (ColumnarLoader(csr_bags=True)): ids [nnz] + offsets [B + 1] instead of DataReader's padded
                    # ids + mask; same result, the padded length comes from the config (negative L marks the form here)
                    L = self.array_max_length.get(n)
                    if L is None:
                        raise ValueError(f"Max length for array feature '{n}' missing in config.")
                    key_items.append((n, NRX_BAG_MASKED_MEAN, -int(L)))
                    continue
                has_mask = f"{n}_mask" in batch and batch[f"{n}_mask"] is not None
                key_items.append((n, NRX_BAG_MASKED_MEAN if has_mask else NRX_BAG_MEAN, int(batch[n].shape[1])))
            else:
                key_items.append((n, NRX_SPARSE, 0))
        key = (tuple(key_items), bool(fm), tuple(sorted(wide_names)))
        hit = self._plan_cache.get(key)
        if hit is not None:
            return hit
        slots, table_names, dims = [], [], []
        col = wcol = 0
        for n, kind, L in key_items:
            if kind == NRX_DENSE:
                if fm:
                    raise RuntimeError("FM fields must be embedding features of one common dim "
                                       "(the reference's torch.stack fails on a dense field)")
                slots.append(ops.Slot(n, NRX_DENSE, -1, 1, 0, col))
                dims.append(1)
                col += 1
                continue
            tname = self._get_emb_feature_name(n)
            if tname not in self.embedding_tables:
                raise ValueError(f"Embedding table not found for {n} (mapped to {tname})")
            if tname not in table_names:
                table_names.append(tname)
            D = self.embedding_tables[tname].embedding_dim
            wide = n in wide_names
            if wide and kind != NRX_SPARSE:
                raise ValueError(f"wide feature '{n}' must be a single-valued sparse feature")
            slots.append(ops.Slot(n, kind, table_names.index(tname), D, abs(L), col, wide_col=wcol if wide else -1,
                                  fm_field=int(fm), flags=NRX_FEAT_BAG_CSR if L < 0 else 0))
            dims.append(D)
            if wide:
                wcol += 1
                col += D - 1
            else:
                col += D
        if fm and len({s.dim for s in slots}) > 1:
            raise RuntimeError("stack expects each tensor to be equal size: FM fields must share one embedding dim")
        plan = ops.EmbedPlan(slots, out_width=col, wide_width=wcol, use_fm=fm)
        entry = (plan, table_names, dims, present)
        self._plan_cache[key] = entry
        return entry

    def _embed(self, batch: Dict[str, torch.Tensor], feature_names, fm: bool = False, wide_names: Sequence[str] = (),
               out_ld: Optional[int] = None, need_out: bool = True):
        """One fused launch.  Returns (out | None, wide | None, fm | None, dims, names).
        Host cost: the plan, the table list and the input key lists are cached per (feature set, batch keys) -- a call is
        two dict lookups + one list comprehension over the batch before ops.embed_apply (profiles/r02_host_overhead.txt)."""
        ck = (frozenset(feature_names), tuple(batch), bool(fm), tuple(wide_names))
        ent = self._embed_cache.get(ck)
        if ent is not None:
            plan, tables, in_names, mask_names, bag_lens, dims, present = ent
            for n, L in bag_lens:                       # array features: the padded length is part of the plan
                if batch[n].dim() != 2 or batch[n].shape[1] != L:
                    ent = None
                    break
            if ent is not None and tables and tables[0] is not self.embedding_tables[self._first_table[ck]].weight:
                ent = None                              # the tables were replaced (e.g. shard_model_): rebuild
        if ent is None:
            plan, table_names, dims, present = self._plan(batch, feature_names, fm, wide_names)
            if not present:
                return None, None, None, [], []
            tables = [self.embedding_tables[t].weight for t in table_names]
            in_names = [s.name for s in plan.slots]
            mask_names = [(f"{s.name}_offsets" if s.flags & NRX_FEAT_BAG_CSR else f"{s.name}_mask") if s.kind == NRX_BAG_MASKED_MEAN
                          else None for s in plan.slots]
            bag_lens = [(s.name, s.bag_len) for s in plan.slots
                        if s.kind in (NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN) and not s.flags & NRX_FEAT_BAG_CSR]
            any_mask = any(m is not None for m in mask_names)
            ent = (plan, tables, in_names, mask_names if any_mask else None, bag_lens, list(dims), list(present))
            if table_names:
                self._first_table[ck] = table_names[0]
            self._embed_cache[ck] = ent
            mask_names = ent[3]
        inputs = [batch[n] for n in in_names]
        weights = self._no_weights(len(in_names)) if mask_names is None else [None if m is None else batch.get(m) for m in mask_names]
        sg = self.sparse_grad
        if sg in ("fused", "exact"):
            if self._sparse_sink is None:
                self._sparse_sink = ops.SparseGradSink()
            sg = self._sparse_sink if torch.is_grad_enabled() else False
        narrow = False
        if out_ld is not None and out_ld < 0:          # -k: pad the row stride to a multiple of k floats, hand back the [B, width] view
            k = -int(out_ld)
            out_ld = (plan.out_width + k - 1) // k * k
            narrow = out_ld != plan.out_width
            if not narrow:
                out_ld = None
        out, wide, fmv = ops.embed_apply(plan, tables, inputs, weights, out_ld=out_ld, need_out=need_out, sparse_grad=sg,
                                         index_check=self.index_check, narrow=narrow)
        return out, wide, fmv, list(dims), list(present)

    def _no_weights(self, n: int):
        w = self._none_lists.get(n)
        if w is None:
            w = self._none_lists[n] = [None] * n
        return w

    def get_embeddings_from_batch(self, batch: Dict[str, torch.Tensor], feature_names) -> Tuple[torch.Tensor, List[int], List[str]]:
        """(features [B, sum D], dims, names) for `sorted(feature_names)` present in the batch."""
        out, _, _, dims, names = self._embed(batch, feature_names)
        if out is None:
            return torch.tensor([]).to(self.device), [], []
        return out, dims, names

    # ------------------------------------------------------------------ validation (:181-218, :320-528)
    def setup(self, stage: str = None):
        """Log / checkpoint dirs, `val_log.log`, and the warm-user set (base_model.py:181-211).  Tolerant where
        the reference is not: a missing `preprocess/train_user_ids.json` means "no cold users" instead of a crash."""
        logger_dir = getattr(getattr(self, "logger", None), "log_dir", None)
        self.log_dir = logger_dir or (self.out_basedir if self.out_basedir else "./logs")
        self.ckpt_dir = os.path.join(self.log_dir, "ckpts")
        os.makedirs(self.ckpt_dir, exist_ok=True)
        val_log_path = os.path.join(self.log_dir, "val_log.log")
        if not os.path.exists(val_log_path):
            open(val_log_path, "w").close()
        self.user_in_train_path = os.path.join(self.out_basedir, "preprocess", "train_user_ids.json")
        self.user_in_train_set = set()
        if os.path.exists(self.user_in_train_path):
            with open(self.user_in_train_path, "r") as f:
                self.user_in_train_set = set(json.load(f))

    def validation_step(self, batch, batch_idx):
        """Keeps the step's (user, score, label) on the device -- the reference copies them to the host and
        appends Python tuples per sample (:320-330)."""
        scores = self.inference(batch).reshape(-1)
        n = scores.numel()
        self._val_uid.append(batch["user_id"].reshape(-1)[:n])
        self._val_score.append(scores.detach().float())
        self._val_label.append(batch["label"].reshape(-1)[:n].float())     # the reference zips view(-1): first n labels

    def backward(self, loss, *args, **kwargs):
        """Lightning hook (LightningModule.backward; the reference defines none: Lightning's default is ``loss.backward()``).  Every
        reference trainer runs on ONE device (sort/deep/train.py:38-44, ``devices=1``), so the backward's nodes run on the calling thread:
        PyTorch's engine otherwise hands each step to its device thread -- ~140 us of wake-up and GIL hand-over per step on the hosts
        measured (profiles/r03_host_overhead.txt), more than the path's kernels take.  Same graph, same kernels, same results."""
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward(*args, **kwargs)

    def on_train_epoch_end(self):
        """Lightning hook (the reference defines none on BaseModel): an out-of-range id met by the LAST training batches'
        deferred index checks must not stay unreported until some later call -- it raises here (IndexError)."""
        ops.flush_index_checks()

    def on_fit_end(self):
        ops.flush_index_checks()

    def on_validation_epoch_end(self):
        """GAUC / NDCG@10 / HR@10 / MRR@10 / AUC / LogLoss with the warm-cold split, computed on the device
        (news_recsys_amd/metrics.py); prints and appends the reference's text block to val_log.log.  Unlike the
        reference (whose `user_scores_dict` is never cleared and so accumulates across epochs, :179), the
        epoch's samples are dropped afterwards."""
        from ...metrics import format_val_log, ranking_metrics
        if not self._val_score:
            return None
        uid, sc, lb = torch.cat(self._val_uid), torch.cat(self._val_score), torch.cat(self._val_label)
        self._val_uid, self._val_score, self._val_label = [], [], []
        warm = getattr(self, "user_in_train_set", None)
        warm_ids = [int(u) for u in warm if str(u).lstrip("-").isdigit()] if warm else None
        ops.flush_index_checks()           # an out-of-range id of the epoch's deferred checks surfaces here at the latest
        results = ranking_metrics(uid, sc, lb, warm_ids, k=10)
        msg = format_val_log(results, getattr(self, "current_epoch", 0), k=10)
        print(msg)
        if hasattr(self, "log_dir"):
            with open(os.path.join(self.log_dir, "val_log.log"), "a") as f:
                f.write(msg)
        self.last_val_results = results
        return results

    # ------------------------------------------------------------------ abstract interface (:313-318)
    def forward(self, x):
        raise NotImplementedError("Subclasses must implement forward()")

    def inference(self, batch):
        raise NotImplementedError("Subclasses must implement inference()")

    # ------------------------------------------------------------------ checkpoint (:531-536)
    def load_model(self, model_path: str):
        if not os.path.exists(model_path):
            raise FileNotFoundError(f"Model checkpoint not found: {model_path}")
        state = torch.load(model_path, map_location=self.device)
        if isinstance(state, dict) and "state_dict" in state:
            state = state["state_dict"]
        self.load_state_dict(state, strict=True)
        return self

    # ------------------------------------------------------------------ shared training glue
    def bceLoss(self, preds, labels):
        return torch.nn.functional.binary_cross_entropy(preds.view(-1), labels.view(-1), reduction="mean")

    def _ranking_training_step(self, batch):
        """training_step of every sort model (e.g. sort/deep/model.py:45-52)."""
        scores = self.forward(batch)
        labels = batch["label"][:, 0]
        loss = self.bceLoss(scores, labels)
        try:
            from sklearn.metrics import roc_auc_score
            train_auc = roc_auc_score(labels.detach().cpu().numpy(), scores.detach().cpu().numpy().reshape(-1))
            self.log("train_auc", train_auc, prog_bar=True, on_step=False, on_epoch=True)
        except ValueError:          # a batch with a single class has no AUC
            pass
        self.log("train_loss", loss, prog_bar=True, on_epoch=True, on_step=False)
        return loss

    def _ranking_optimizers(self):
        """configure_optimizers of every sort model (e.g. sort/deep/model.py:54-65)."""
        from ..model_utils.lr_schedule import CosinDecayLR
        hp = self.train_hparams
        if self.sparse_grad:
            # AdamW cannot take sparse grads: tables go to SparseAdam (lazy moments, no weight decay --
            # a documented deviation from the reference's dense AdamW), everything else stays AdamW
            from ..model_utils.optim import SparseDenseAdam
            table_params = [e.weight for e in self.embedding_tables.values()]
            ids = {id(p) for p in table_params}
            sink = None
            if self.sparse_grad in ("fused", "exact"):
                if self._sparse_sink is None:
                    self._sparse_sink = ops.SparseGradSink()
                sink = self._sparse_sink
            optimizer = SparseDenseAdam(table_params, [p for p in self.parameters() if id(p) not in ids], lr=hp.lr, fused_sink=sink,
                                        exact=self.sparse_grad == "exact")
        else:
            from ..model_utils.optim import dense_adamw
            optimizer = dense_adamw(self.parameters(), lr=hp.lr, betas=(0.9, 0.999))     # torch.optim.AdamW; its one-pass kernel on the GPU
        sched = CosinDecayLR(optimizer, lrs=[hp.lr, hp.min_lr], milestones=list(hp.lr_milestones))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}
