"""Wide & Deep ranker.  For features listed in `wide_and_deep_cfg.wide_feature_names` column 0 of the
embedding is the wide (linear) term and columns 1.. go to the deep MLP; other features go to the MLP
whole.  Reference: src/model/sort/widedeep/model.py (WideDeepModel :14-27, WideDeep :29-69).
The column routing happens inside the fused gather launch (no slice / cat passes)."""
import torch

from ...BaseModel.base_model import BaseModel
from ...model_utils.utils import MLP


class WideDeepModel(torch.nn.Module):
    def __init__(self, input_dim, hidden_dims=(32, 32, 1)):
        super().__init__()
        self.wide_network = torch.sum
        self.deep_network = MLP(dims=[input_dim] + list(hidden_dims))
        self.bias = torch.nn.Parameter(torch.zeros(1))

    def forward(self, wide_x, deep_x):
        wide_out = self.wide_network(wide_x, dim=1, keepdim=True) + self.bias
        return torch.sigmoid(wide_out + self.deep_network(deep_x))


class WideDeep(BaseModel):
    def __init__(self, config_path):
        super().__init__(config_path)
        self.wide_feature_names = set(self.config.wide_and_deep_cfg.wide_feature_names)
        self.score_fc = WideDeepModel(input_dim=self.user_input_dim + self.item_input_dim - len(self.wide_feature_names),
                                      hidden_dims=[128, 128, 128, 64, 1])

    def get_inp_embedding(self, batch):
        # deep_x is the [B, deep_width] view of a buffer whose row stride is padded to a multiple of 32 floats (one 128-byte line): every
        # wide feature takes one column out of the deep row, so with the natural stride (C5: 1270 floats) neither the rows nor the features'
        # 128-byte pieces start on a line and every piece is written as two partial lines -- the launch is bound by its L2 request count
        # (C5: 154.5 us at stride 1270, 143.3 us at 1280 with aligned 16-byte chunks, 133.5 us for the plain concat; profiles/r04_wide_split.txt).
        # The MLP's first Linear reads the strided view as it is (lda = the padded stride).
        deep_x, wide_x, _, _, _ = self._embed(batch, self.user_feature_names | self.item_feature_names,
                                              wide_names=tuple(self.wide_feature_names), out_ld=-32)
        return wide_x, deep_x

    def forward(self, x):
        wide_x, deep_x = self.get_inp_embedding(x)
        return self.score_fc(wide_x, deep_x)

    def training_step(self, batch, batch_idx):
        return self._ranking_training_step(batch)

    def configure_optimizers(self):
        return self._ranking_optimizers()

    @torch.no_grad()
    def inference(self, batch):
        return self.forward(batch)
