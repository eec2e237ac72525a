"""Logistic regression = dim-1 embeddings summed.  Reference: src/model/sort/lr/model.py:13-31
(note: returns shape [B], not [B,1]).  With dim-1 tables the row sum IS the first-order term of the
FM epilogue, so the whole model is one fused launch that writes 4 bytes per impression."""
import torch

from ...BaseModel.base_model import BaseModel


class LR(BaseModel):
    def __init__(self, config_path):
        super().__init__(config_path)
        self.score_fc = torch.sum

    def get_inp_embedding(self, batch):
        features, _, _ = self.get_embeddings_from_batch(batch, self.user_feature_names | self.item_feature_names)
        return features

    def forward(self, x):
        names = self.user_feature_names | self.item_feature_names
        plan, _, dims, _ = self._plan(x, names, False, ())
        if dims and all(d == 1 for d in dims) and all(s.kind == 0 for s in plan.slots):
            _, _, logit, _, _ = self._embed(x, names, fm=True, need_out=torch.is_grad_enabled())
            return torch.sigmoid(logit)
        return torch.sigmoid(self.score_fc(self.get_inp_embedding(x), dim=1))

    def training_step(self, batch, batch_idx):
        return self._ranking_training_step(batch)

    def configure_optimizers(self):
        return self._ranking_optimizers()

    @torch.no_grad()
    def inference(self, batch):
        return self.forward(batch)
