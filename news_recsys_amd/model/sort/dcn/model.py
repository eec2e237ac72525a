"""DCN ranker: sigmoid(MLP(cat[x, cross(x)])).  Reference: src/model/sort/dcn/model.py
(DCNModel :15-29, DCN :31-76).  The reference hard-codes 3 v1 cross layers (:36); optional config
keys `dcn_cfg.cross_num_layers`, `dcn_cfg.version` (1 = reference behaviour, 2 = DCNv2Net, which
the reference defines but never instantiates) and, for version 2, `dcn_cfg.math` (fp32 | bf16x3) are accepted, defaulting to the
reference.

forward(batch) writes the concat straight into the left half of a [B, 2D] buffer and the cross
output into its right half, so the reference's torch.cat([x, cross]) costs no extra pass."""
import torch

from ...BaseModel.base_model import BaseModel
from ...model_utils.utils import MLP
from .dcn_arch import DCNNet, DCNv2Net, DCNLayer, DCNv2Layer  # noqa: F401  (re-exported like the reference)


class DCNModel(torch.nn.Module):
    def __init__(self, input_dim, cross_num_layers=3, deep_hidden_dims=(32, 32, 1), version=1, math="fp32"):
        super().__init__()
        self.input_dim = input_dim
        self.version = version
        if version == 1:
            self.cross_net = DCNNet(input_dim=input_dim, num_layers=cross_num_layers)
        else:
            self.cross_net = DCNv2Net(input_dim=input_dim, num_layers=cross_num_layers, math=math)
        self.score_fc = MLP(dims=[input_dim * 2] + list(deep_hidden_dims))

    def forward(self, x):
        cross_f = self.cross_net(x)
        return torch.sigmoid(self.score_fc(torch.cat([x, cross_f], dim=1)))

    def forward_buf_(self, buf):
        """buf [B, 2D], left half = x.  v1 only."""
        return torch.sigmoid(self.score_fc(self.cross_net.forward_cat_(buf)))


class DCN(BaseModel):
    def __init__(self, config_path):
        super().__init__(config_path)
        cfg = self.config.get("dcn_cfg", {}) or {}
        # "auto" (default): one fused launch (gather -> cat[x, cross(x)]) whenever the grouped kernel applies
        # (ops.fused_cross_is_fast: uniform 32/64-wide rows, 2..8 plain features); true: whenever the library accepts
        # the feature mix; false: always gather + cross as two launches
        fz = cfg.get("fuse_gather_cross", "auto")
        self.fuse_gather_cross = "auto" if str(fz).lower() == "auto" else bool(fz)
        self.score_fc = DCNModel(input_dim=self.user_input_dim + self.item_input_dim,
                                 cross_num_layers=int(cfg.get("cross_num_layers", 3)),
                                 deep_hidden_dims=[128, 128, 128, 64, 1], version=int(cfg.get("version", 1)),
                                 math=str(cfg.get("math", "fp32")))      # v2 only: fp32 (exact fma chain) | bf16x3 (split-bf16 MFMA)

    def get_inp_embedding(self, batch):
        features, _, _ = self.get_embeddings_from_batch(batch, self.user_feature_names | self.item_feature_names)
        return features

    def forward(self, x):
        m = self.score_fc
        names = self.user_feature_names | self.item_feature_names
        if m.version == 1 and len(m.cross_net.cross_net) > 0:
            if self.fuse_gather_cross and getattr(self, "_shard_engine", None) is None:
                # gather -> cat[x, cross(x)] in ONE launch (training too: backward = cross backward on the saved buffer,
                # then the embedding backward).  C3 shape: 47.9 us vs 60.0 us for the two launches below.
                from .... import ops
                plan, table_names, _, present = self._plan(x, names, False, ())
                if present and (self.fuse_gather_cross is True or ops.fused_cross_is_fast(plan)):
                    try:
                        w, b = m.cross_net.stacked()
                        sg = self.sparse_grad
                        if sg == "fused":
                            if self._sparse_sink is None:
                                self._sparse_sink = ops.SparseGradSink()
                            sg = self._sparse_sink if torch.is_grad_enabled() else False
                        buf = ops.embed_dcn_v1(plan, [self.embedding_tables[t].weight for t in table_names],
                                               [x[s.name] for s in plan.slots], w, b, sparse_grad=sg)
                        return torch.sigmoid(m.score_fc(buf))
                    except ops.FusedUnsupported:
                        pass
            buf, _, _, _, _ = self._embed(x, names, out_ld=2 * m.input_dim)
            return m.forward_buf_(buf)
        return m(self.get_inp_embedding(x))

    def training_step(self, batch, batch_idx):
        return self._ranking_training_step(batch)

    def configure_optimizers(self):
        return self._ranking_optimizers()

    @torch.no_grad()
    def inference(self, batch):
        return self.forward(batch)
