"""FM ranker.  Column 0 of every field's embedding is its first-order weight, columns 1.. its factor
vector; second order by the sum-square identity.  Reference: src/model/sort/fm/model.py
(FMModel :12-26, FM.get_inp_embedding :48-59).

`forward(batch)` runs ONE fused launch: the FM sums are formed while the looked-up rows are still in
registers, so neither w [B,F] nor v [B,F,D-1] is materialised (and in inference not even the concat).
`FMModel.forward(w, v)` keeps the reference signature for callers that do have w / v tensors."""
import torch

from .... import ops
from ...BaseModel.base_model import BaseModel


class FMModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.bias = torch.nn.Parameter(torch.zeros(1))

    def forward(self, w, v):
        """w [B,F], v [B,F,K] -> sigmoid(bias + sum_f w + 0.5*sum_k[(sum_f v)^2 - sum_f v^2]) [B,1]."""
        B, Fn, K = v.shape
        feat = torch.cat([w.unsqueeze(2), v], dim=2).reshape(B, Fn * (K + 1))
        return self.from_logit(ops.fm_interaction(feat, Fn, K + 1))

    def from_logit(self, fm_logit):
        return ops.fm_head(fm_logit, self.bias)          # sigmoid(bias + logit) [B, 1]: one launch (and one in the backward)


class FM(BaseModel):
    def __init__(self, config_path):
        super().__init__(config_path)
        self.score_fc = FMModel()

    def _names(self):
        return self.user_feature_names | self.item_feature_names

    def get_inp_embedding(self, batch):
        """(w [B,F], v [B,F,D-1]) like the reference -- materialising API, for compatibility/debugging."""
        features, dims, _ = self.get_embeddings_from_batch(batch, self._names())
        if len(set(dims)) != 1:
            raise RuntimeError("stack expects each tensor to be equal size: FM fields must share one embedding dim")
        f3 = features.view(features.shape[0], len(dims), dims[0])
        return f3[:, :, 0], f3[:, :, 1:]

    def forward(self, x):
        _, _, fm, _, _ = self._embed(x, self._names(), fm=True, need_out=torch.is_grad_enabled())
        return self.score_fc.from_logit(fm)

    def training_step(self, batch, batch_idx):
        return self._ranking_training_step(batch)

    def configure_optimizers(self):
        return self._ranking_optimizers()

    @torch.no_grad()
    def inference(self, batch):
        return self.forward(batch)
