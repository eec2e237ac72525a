"""Cross networks.  Reference: src/model/sort/dcn/dcn_arch.py (DCNLayer :5-30, DCNv2Layer :33-50,
DCNNet :53-70, DCNv2Net :73-91).  Parameter shapes / module nesting follow the reference so its
checkpoints load unchanged; the arithmetic runs in the HIP kernels:

  v1  x_{l+1} = (x0 x_l^T) w + b + x_l  is evaluated as  x0 * (x_l . w) + b + x_l  (no [B,D,D]
      temporary), all layers in one launch with the row in registers;
  v2  x_{l+1} = relu(x0 * (W x_l + b) + x_l) on the fp32 matrix cores, epilogue fused."""
import torch
import torch.nn as nn

from .... import ops


class DCNLayer(nn.Module):
    def __init__(self, dim=32):
        super().__init__()
        self.w = nn.Parameter(torch.empty(dim, 1))
        self.b = nn.Parameter(torch.zeros(dim, 1))
        nn.init.xavier_uniform_(self.w)

    def forward(self, x_l, x_0):
        """Per-layer API of the reference (dcn_arch.py:14-30): x_0 * (x_l . w) + b + x_l for ANY layer -- x_l need not
        be x_0 (nrx_dcn_v1_fwd / _bwd take the layer-0 input separately).  DCNNet.forward still runs the whole stack
        in one launch; chaining layers through this method is the same arithmetic, one launch per layer."""
        return ops.dcn_v1(x_l, self.w[:, 0].unsqueeze(0), self.b[:, 0].unsqueeze(0), x0=x_0)


class DCNv2Layer(nn.Module):
    def __init__(self, dim=32):
        super().__init__()
        self.linear = nn.Linear(dim, dim, bias=True)

    def forward(self, x_l, x_0):
        """x_0 * (W x_l + b) + x_l on the matrix cores (dcn_arch.py:39-50); no ReLU (DCNv2Net adds it)."""
        return ops.dcn_v2_layer(x_0, x_l, self.linear.weight, self.linear.bias, relu=False)


class DCNNet(nn.Module):
    def __init__(self, input_dim, num_layers=3):
        super().__init__()
        self.cross_net = nn.ModuleList([DCNLayer(input_dim) for _ in range(num_layers)])

    def stacked(self):
        """(w, b) as [n_layers, dim] tensors for the fused launches -- each ONE autograd node over the layers' own [dim, 1] parameters
        (ops.pack_rows: no per-layer select + torch.stack nodes, per-layer gradients are views of the kernels' [n_layers, dim] outputs)."""
        if len(self.cross_net) == 0:
            return None, None
        return ops.pack_rows([l.w for l in self.cross_net]), ops.pack_rows([l.b for l in self.cross_net])

    def forward(self, x):
        if len(self.cross_net) == 0:
            return x
        # the layers' own [dim, 1] parameters go in as they are (packed inside the op, outside autograd): no torch.stack nodes
        return ops.dcn_v1_layers(x, [l.w for l in self.cross_net], [l.b for l in self.cross_net])

    def forward_cat_(self, buf):
        """buf [B, 2D] with x in the left half: fills the right half with cross(x) in place."""
        w, b = self.stacked()
        return ops.dcn_v1_cat_(buf, w, b)


class DCNv2Net(nn.Module):
    def __init__(self, input_dim, num_layers=3, math="fp32"):
        super().__init__()
        layers = []
        for _ in range(num_layers):
            layers.append(DCNv2Layer(input_dim))
            layers.append(nn.ReLU())
        self.cross_net = nn.ModuleList(layers)
        self.math = math            # "fp32" (default, exact fp32 fma chain) | "bf16x3" (split-bf16 matrix math, dcn_cfg.math)

    def forward(self, x):
        lins = [l.linear for l in self.cross_net if isinstance(l, DCNv2Layer)]
        if not lins:
            return x
        # the layers' own parameter tensors go to the kernels as they are: no torch.stack (two kernels + autograd nodes per step)
        return ops.dcn_v2(x, [l.weight for l in lins], [l.bias for l in lins], relu=True, math=self.math)
