"""Deep ranker: fused embedding concat -> MLP -> sigmoid.
Reference: src/model/sort/deep/model.py (DeepModel :12-21, Deep :24-70)."""
import torch

from ...BaseModel.base_model import BaseModel
from ...model_utils.utils import MLP


class DeepModel(torch.nn.Module):
    def __init__(self, input_dim, hidden_dims=(32, 32, 1)):
        super().__init__()
        self.network = MLP(dims=[input_dim] + list(hidden_dims))

    def forward(self, x):
        return torch.sigmoid(self.network(x))


class Deep(BaseModel):
    def __init__(self, config_path):
        super().__init__(config_path)
        self.score_fc = DeepModel(input_dim=self.user_input_dim + self.item_input_dim, hidden_dims=[128, 128, 128, 64, 1])

    def get_inp_embedding(self, batch):
        features, _, _ = self.get_embeddings_from_batch(batch, self.user_feature_names | self.item_feature_names)
        return features

    def forward(self, x):
        return self.score_fc(self.get_inp_embedding(x))

    def training_step(self, batch, batch_idx):
        return self._ranking_training_step(batch)

    def configure_optimizers(self):
        return self._ranking_optimizers()

    @torch.no_grad()
    def inference(self, batch):
        return self.score_fc(self.get_inp_embedding(batch))
