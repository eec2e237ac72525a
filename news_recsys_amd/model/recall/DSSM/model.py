"""DSSM two-tower recall.  Reference: src/model/recall/DSSM/model.py (towers :26-44, forward :51-73,
losses :75-110, get_user_embedding / get_item_embedding :148-180).  The reference file is stale
(MovieLens-era imports, calls a method BaseModel does not define -- SURVEY fact 3); its arithmetic is
the spec.  The tower inputs -- per-feature lookup, masked mean-pool of array features, concat -- are one
fused HIP launch per tower; towers / normalize / losses are plain torch.

Deviations, all documented in SURVEY: features are concatenated in SORTED order (the reference iterates
a Python set: order depends on PYTHONHASHSEED); `forward(x, perms=None)` accepts explicit negative-
sampling permutations so results are reproducible (the reference draws torch.randperm, :63)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

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
        optimizer = torch.optim.AdamW(self.parameters(), lr=hp["lr"], betas=(0.9, 0.999))
        sched = CosinDecayLR(optimizer, lrs=[hp["lr"], hp["min_lr"]], milestones=list(hp["lr_milestones"]))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}

    @torch.no_grad()
    def inference(self, batch):
        return (F.normalize(self.user_fc(self.get_user_embedding(batch)), p=2, dim=1),
                F.normalize(self.item_fc(self.get_item_embedding(batch)), p=2, dim=1))
