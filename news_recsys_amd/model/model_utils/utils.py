"""MLP head (reference: src/model/model_utils/utils.py:6-17).  The dense head is plain GEMM work
and stays nn.Linear on rocBLAS/hipBLASLt -- it is adjacent to, not part of, the HIP hot path.
Module nesting (`.network` = nn.Sequential of Linear/ReLU) keeps the reference's state_dict keys.

Opt-in (`NRX_MLP_WGRAD=1`, or `utils.MLP_WGRAD = True` before the model is built): the layers' WEIGHT gradient runs on
nrx_linear_wgrad -- at batch 65 536 that GEMM contracts over the batch and is the largest single item of a full training step on
the vendor library (tools/bench_full_step_c2.py).  Forward, input gradient, parameters and state_dict keys are unchanged."""
import os

import torch.nn as nn

MLP_WGRAD = os.environ.get("NRX_MLP_WGRAD", "0") == "1"


class _Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose backward computes g_W with nrx_linear_wgrad on fp32 CUDA inputs."""

    def forward(self, x):
        if x.is_cuda and x.dtype == self.weight.dtype and (x.requires_grad or self.weight.requires_grad):
            from ... import ops
            return ops.linear(x, self.weight, self.bias)
        return super().forward(x)


class MLP(nn.Module):
    def __init__(self, dims=(16, 32, 32, 1)):
        super().__init__()
        dims = list(dims)
        layers = []
        last = len(dims) - 2
        linear = _Linear if MLP_WGRAD else nn.Linear
        for i, (d_in, d_out) in enumerate(zip(dims[:-1], dims[1:])):
            layers.append(linear(d_in, d_out))
            if i < last:
                layers.append(nn.ReLU())
        self.network = nn.Sequential(*layers)

    def forward(self, x):
        return self.network(x)
