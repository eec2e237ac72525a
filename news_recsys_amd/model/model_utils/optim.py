"""Optimizer for the row-sparse gradient mode (`embeddings.sparse_grad: true`).

The reference trains everything with one dense `AdamW(self.parameters())` (sort/deep/model.py:55): on a
100M-row table that is a full-table read-modify-write of weights and both moments every step.  With
row-sparse table grads the tables are updated by `torch.optim.SparseAdam` (moments touched only for the
looked-up rows; no weight decay) and the dense parameters keep AdamW.  This wrapper presents both as
one `Optimizer` so `configure_optimizers()` keeps its reference shape (one optimizer + one scheduler)."""
import torch


class SparseDenseAdam(torch.optim.Optimizer):
    def __init__(self, sparse_params, dense_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        sparse_params, dense_params = list(sparse_params), list(dense_params)
        groups = [{"params": sparse_params, "sparse": True}]
        if dense_params:
            groups.append({"params": dense_params, "sparse": False})
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._sparse = torch.optim.SparseAdam(sparse_params, lr=lr, betas=betas, eps=eps)
        self._dense = torch.optim.AdamW(dense_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) if dense_params else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for g in self.param_groups:           # a scheduler edits self.param_groups: forward the lr
            for inner in ((self._sparse,) if g["sparse"] else ((self._dense,) if self._dense else ())):
                for ig in inner.param_groups:
                    ig["lr"] = g["lr"]
        self._sparse.step()
        if self._dense is not None:
            self._dense.step()
        return loss

    def zero_grad(self, set_to_none: bool = True):
        self._sparse.zero_grad(set_to_none)
        if self._dense is not None:
            self._dense.zero_grad(set_to_none)
