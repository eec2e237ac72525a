"""DSSM two-tower recall.  Reference: src/model/recall/DSSM/model.py (towers :26-44, forward :51-73,
losses :75-110, get_user_embedding / get_item_embedding :148-180).  The reference file is stale
(MovieLens-era imports, calls a method BaseModel does not define -- SURVEY fact 3); its arithmetic is
the spec.  The tower inputs -- per-feature lookup, masked mean-pool of array features, concat -- are one
fused HIP launch per tower; towers / normalize / losses are plain torch.

Deviations, all documented in SURVEY: features are concatenated in SORTED order (the reference iterates
a Python set: order depends on PYTHONHASHSEED); `forward(x, perms=None)` accepts explicit negative-
sampling permutations so results are reproducible (the reference draws torch.randperm, :63).

Recall evaluation (`on_train_epoch_end` :230-254, `hit_rate` :182-228): the reference builds a faiss
IndexFlatIP on the host and searches one user at a time (it raises unless batch_size == 1), over-fetching
k + len(history) and filtering in Python.  Here the item matrix stays in HBM and a whole validation batch
is one `nrx_topk_ip` call with the history passed as per-query exclusion lists; any batch size works.
`user_history` is what BaseModel loads from `paths.user_history_path` (base_model.py:55-58; JSON, so user keys are
strings and a user's entry is a dict keyed by item id, model.py:206) -- or any {user id -> iterable of item ids}
assigned by hand.  The reference translates batch ids to the history's id space through `emb_idx_2_val_dict`
(model.py:205,215), which its tree never defines; here it is an optional attribute
{feature name: {str(embedding index): true id}}: when set, ids are mapped through it exactly as the reference
does, otherwise the history is taken to be in the id space of the batch columns `user_id_feature` /
`item_id_feature` (hparams; default 'user_id' / 'movie_id' as in the reference).  Keys may be ints or strings."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ...BaseModel.base_model import BaseModel


def _tower(in_dim):
    return nn.Sequential(nn.Linear(in_dim, 128), nn.LeakyReLU(0.2), nn.Linear(128, 128), nn.LeakyReLU(0.2),
                         nn.Linear(128, 64), nn.LeakyReLU(0.2), nn.Linear(64, 16))


class DSSM(BaseModel):
    def __init__(self, config_path, dataloaders={}, hparams={}):
        super().__init__(config_path)
        self.save_hyperparameters(hparams)
        self.hparams_ = dict(hparams)
        self.user_fc = _tower(self.user_input_dim)
        self.item_fc = _tower(self.item_input_dim)
        self.movies_dataloader = dataloaders.get("movies_dataloader", None)
        self.val_dataloader_ = dataloaders.get("val_dataloader", None)
        self.user_id_feature = self.hparams_.get("user_id_feature", "user_id")
        self.item_id_feature = self.hparams_.get("item_id_feature", "movie_id")
        if not hasattr(self, "user_history"):      # BaseModel.__init__ has loaded paths.user_history_path if configured
            self.user_history = {}
        self.emb_idx_2_val_dict = None        # optional {feature: {str(emb idx): true id}} (reference model.py:205,215)
        self.all_item_embeddings = None       # [N, 16] fp32, device resident (reference: numpy + faiss index)
        self.idx_item_emb_dic = {}            # index position -> item id value (model.py:233,247)
        self.last_hit_rate = None

    def get_user_embedding(self, batch):
        out, _, _ = self.get_embeddings_from_batch(batch, self.user_feature_names)
        return out

    def get_item_embedding(self, batch):
        out, _, _ = self.get_embeddings_from_batch(batch, self.item_feature_names)
        return out

    # the stale reference spells it `get_features_embedding` (model.py:151)
    def get_features_embedding(self, feature_name, feature_value):
        return self.get_feature_embedding(feature_name, feature_value)

    def forward(self, x, perms=None):
        user_emb = self.user_fc(self.get_user_embedding(x))
        item_emb = self.item_fc(self.get_item_embedding(x))
        B = item_emb.size(0)
        n_neg = int(self.hparams_.get("negative_sample_rate", 1))
        neg = []
        for i in range(n_neg):
            idx = perms[i].to(item_emb.device) if perms is not None else torch.randperm(B, device=item_emb.device)
            neg.append(item_emb[idx])
        neg_item_emb = torch.stack(neg, dim=1)
        return (F.normalize(user_emb, p=2, dim=1), F.normalize(item_emb, p=2, dim=1),
                F.normalize(neg_item_emb, p=2, dim=-1))

    def triplet_loss(self, user_emb, pos_item_emb, neg_item_emb, margin=1.0, mask=None):
        """As written in the reference (model.py:75-90), including its [B] vs [B,1] broadcast: the hinge is
        taken over a [B, B] matrix (entry [i, j] = margin - pos_j + neg_i) and averaged over all of it."""
        n_neg = neg_item_emb.size(1)
        pos_scores = torch.sum(user_emb * pos_item_emb, dim=1) * n_neg
        neg_scores = torch.bmm(user_emb.unsqueeze(1), neg_item_emb.permute(0, 2, 1)).squeeze(1)
        neg_scores = torch.sum(neg_scores, dim=1).unsqueeze(1)
        losses = F.relu(margin - pos_scores + neg_scores)
        if mask is not None:
            losses = losses * mask
        return losses.mean()

    def infoNCE_loss(self, user_emb, pos_item_emb, neg_item_emb, temperature=0.1, mask=None):
        """model.py:92-110."""
        pos = torch.sum(user_emb * pos_item_emb, dim=1) / temperature
        neg = torch.bmm(user_emb.unsqueeze(1), neg_item_emb.permute(0, 2, 1)).squeeze(1) / temperature
        logits = torch.cat([pos.unsqueeze(1), neg], dim=1)
        labels = torch.zeros(user_emb.size(0), dtype=torch.long, device=user_emb.device)
        losses = F.cross_entropy(logits, labels, reduction="none")
        if mask is not None:
            losses = losses * mask
        return losses.mean()

    def training_step(self, batch, batch_idx):
        user_emb, item_emb, neg_item_emb = self.forward(batch)
        loss = self.infoNCE_loss(user_emb, item_emb, neg_item_emb, mask=batch["label"][:, 1])
        self.log("train_loss", loss)
        return loss

    def validation_step(self, batch, batch_idx):
        pass

    def test_step(self, batch, batch_idx):
        pass

    def configure_optimizers(self):
        from ...model_utils.lr_schedule import CosinDecayLR
        hp = self.hparams_
        from ...model_utils.optim import dense_adamw
        optimizer = dense_adamw(self.parameters(), lr=hp["lr"], betas=(0.9, 0.999))          # torch.optim.AdamW; its one-pass kernel on the GPU
        sched = CosinDecayLR(optimizer, lrs=[hp["lr"], hp["min_lr"]], milestones=list(hp["lr_milestones"]))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}

    @torch.no_grad()
    def inference(self, batch):
        return (F.normalize(self.user_fc(self.get_user_embedding(batch)), p=2, dim=1),
                F.normalize(self.item_fc(self.get_item_embedding(batch)), p=2, dim=1))

    # ---- recall evaluation -------------------------------------------------------------------------
    def _to_device(self, batch):
        return {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}

    @torch.no_grad()
    def build_item_index(self, movies_dataloader=None):
        """model.py:232-252: run every item through the item tower, L2-normalise, keep the matrix (on the
        device) and the position -> item id map."""
        loader = movies_dataloader if movies_dataloader is not None else self.movies_dataloader
        embs, ids = [], []
        for batch in loader:
            batch = self._to_device(batch)
            embs.append(F.normalize(self.item_fc(self.get_item_embedding(batch)), p=2, dim=1))
            ids.append(batch[self.item_id_feature].reshape(-1).to(torch.int64))
        self.all_item_embeddings = torch.cat(embs, dim=0).contiguous()
        self.item_index_ids = torch.cat(ids, dim=0)
        self.idx_item_emb_dic = dict(enumerate(self.item_index_ids.tolist()))
        self._item_pos = {v: i for i, v in self.idx_item_emb_dic.items()}
        self._item_pos_true = None
        return self.all_item_embeddings

    @staticmethod
    def _lookup(d, key, default=None):
        """dict lookup tolerant of JSON's string keys: tries key, str(key), int(key)."""
        if key in d:
            return d[key]
        sk = str(key)
        if sk in d:
            return d[sk]
        try:
            ik = int(key)
        except (TypeError, ValueError):
            return default
        return d.get(ik, default)

    def _history_positions(self, uid):
        """Index positions (rows of all_item_embeddings) of the items user `uid` has interacted with
        (reference model.py:205-206,213-217: those items are removed from the ranking before the top k is taken)."""
        m = self.emb_idx_2_val_dict
        if m is not None:
            uid = self._lookup(m.get(self.user_id_feature, {}), uid, uid)
        hist = self._lookup(self.user_history, uid, ())
        if not hist:
            return []
        if m is not None:          # history holds TRUE item ids: compare after mapping every indexed item id
            true_pos = getattr(self, "_item_pos_true", None)
            if true_pos is None:
                im = m.get(self.item_id_feature, {})
                true_pos = self._item_pos_true = {}
                for item_id, pos in self._item_pos.items():
                    true_pos.setdefault(str(self._lookup(im, item_id, item_id)), []).append(pos)
            return sorted(p for h in hist for p in true_pos.get(str(h), ()))
        out = []
        for h in hist:             # a dict (the reference's JSON) iterates its keys
            p = self._lookup(self._item_pos, h)
            if p is not None:
                out.append(p)
        return out

    @torch.no_grad()
    def hit_rate(self, k=10, val_dataloader=None):
        """model.py:182-228 for any batch size: a hit = the batch's target item id is among the k best
        items by inner product once the user's history is removed."""
        from ...model_utils.TopKSearcher import exclusion_csr
        if self.all_item_embeddings is None:
            raise ValueError("item index not built: call build_item_index() / on_train_epoch_end() first")
        loader = val_dataloader if val_dataloader is not None else self.val_dataloader_
        hits = torch.zeros((), dtype=torch.int64, device=self.all_item_embeddings.device)
        all_nums = 0
        for batch in loader:
            batch = self._to_device(batch)
            user_emb = F.normalize(self.user_fc(self.get_user_embedding(batch)), p=2, dim=1)
            uids = batch[self.user_id_feature].reshape(-1).tolist()
            lists = [self._history_positions(u) for u in uids]
            excl = exclusion_csr(lists, user_emb.device) if any(lists) else None
            idx, _ = ops.topk_ip(self.all_item_embeddings, user_emb.contiguous(), k, exclude=excl)
            found = self.item_index_ids[idx.clamp_min(0)].masked_fill(idx < 0, -1)
            target = batch[self.item_id_feature].reshape(-1, 1).to(torch.int64)
            hits += (found == target).any(dim=1).sum()
            all_nums += len(uids)
        hit_rate = (int(hits.item()) / all_nums) if all_nums > 0 else 0
        self.last_hit_rate = hit_rate
        self.log(f"Hit_Rate_{k}", hit_rate)
        print(f"Hit Rate@{k}: {hit_rate}")
        return hit_rate

    @torch.no_grad()
    def on_train_epoch_end(self):
        super().on_train_epoch_end()          # deferred index checks of the epoch's last batches
        if self.movies_dataloader is None or self.val_dataloader_ is None:
            return
        self.build_item_index()
        self.hit_rate(k=10)
