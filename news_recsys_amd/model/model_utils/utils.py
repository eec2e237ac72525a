"""MLP head (reference: src/model/model_utils/utils.py:6-17).  The dense head is plain GEMM work
and stays nn.Linear on rocBLAS/hipBLASLt -- it is adjacent to, not part of, the HIP hot path.
Module nesting (`.network` = nn.Sequential of Linear/ReLU) keeps the reference's state_dict keys."""
import torch.nn as nn


class MLP(nn.Module):
    def __init__(self, dims=(16, 32, 32, 1)):
        super().__init__()
        dims = list(dims)
        layers = []
        last = len(dims) - 2
        for i, (d_in, d_out) in enumerate(zip(dims[:-1], dims[1:])):
            layers.append(nn.Linear(d_in, d_out))
            if i < last:
                layers.append(nn.ReLU())
        self.network = nn.Sequential(*layers)

    def forward(self, x):
        return self.network(x)
