"""DeepFM = FM part + Deep MLP sharing one set of embedding tables.

The reference ships NO DeepFM model -- only a documented config block
(documents/config_file_introduction.md:153-176: `deepfm_cfg.fm_feature_names`, `fm_dim`).  It is
composed here from the reference's own parts the way WideDeep composes wide + deep
(sort/widedeep/model.py:24-27):  sigmoid(fm_logit(FM fields) + bias + MLP(concat of all features)),
with the FM part exactly FMModel's pre-sigmoid output (sort/fm/model.py:18-26: column 0 of a field =
first-order weight, columns 1.. = factors).  PARITY: the two parts are pinned by goldens, the
composition is "parity unpinned" (nothing in the reference to compare with).

When the FM fields are all features (default) the whole embedding side is ONE launch producing both
the [B, sum D] concat for the MLP and the FM logit."""
import torch

from ...BaseModel.base_model import BaseModel
from ...model_utils.utils import MLP


class DeepFMModel(torch.nn.Module):
    def __init__(self, input_dim, hidden_dims=(32, 32, 1)):
        super().__init__()
        self.deep_network = MLP(dims=[input_dim] + list(hidden_dims))
        self.bias = torch.nn.Parameter(torch.zeros(1))

    def forward(self, fm_logit, deep_x):
        return torch.sigmoid(fm_logit.unsqueeze(1) + self.bias + self.deep_network(deep_x))


class DeepFM(BaseModel):
    def __init__(self, config_path):
        super().__init__(config_path)
        cfg = self.config.get("deepfm_cfg", {}) or {}
        all_names = self.user_feature_names | self.item_feature_names
        self.fm_feature_names = set(cfg.get("fm_feature_names") or all_names)
        unknown = self.fm_feature_names - all_names
        if unknown:
            raise ValueError(f"deepfm_cfg.fm_feature_names not among the model's features: {sorted(unknown)}")
        fm_dims = {self.embedding_size.get(self._get_emb_feature_name(n)) for n in self.fm_feature_names}
        if len(fm_dims) != 1:
            raise ValueError("DeepFM: FM fields must share one embedding dim (col 0 = weight, cols 1.. = factors)")
        fm_dim = cfg.get("fm_dim")
        if fm_dim is not None and int(fm_dim) != fm_dims.pop() - 1:
            raise ValueError("deepfm_cfg.fm_dim must equal embedding dim - 1 (column 0 is the first-order weight)")
        self.score_fc = DeepFMModel(input_dim=self.user_input_dim + self.item_input_dim, hidden_dims=[128, 128, 128, 64, 1])

    def forward(self, x):
        all_names = self.user_feature_names | self.item_feature_names
        if self.fm_feature_names == all_names:
            deep_x, _, fm, _, _ = self._embed(x, all_names, fm=True)
        else:
            deep_x, _, _, _, _ = self._embed(x, all_names)
            _, _, fm, _, _ = self._embed(x, self.fm_feature_names, fm=True, need_out=torch.is_grad_enabled())
        return self.score_fc(fm, deep_x)

    def training_step(self, batch, batch_idx):
        return self._ranking_training_step(batch)

    def configure_optimizers(self):
        return self._ranking_optimizers()

    @torch.no_grad()
    def inference(self, batch):
        return self.forward(batch)
