"""HIP-graph capture of a whole step (inference or training) of a model built on the fused path.

Small batches -- the reference's YAMLs train with a few hundred to a few thousand samples per step -- are
launch-bound: a Deep training step is ~40 kernel launches plus Python glue (1.6 ms per step at B = 1024 on an
MI355X host) for well under 0.4 ms of GPU work.  The fused HIP launches only enqueue kernels on torch's current
stream and allocate through torch's caching allocator, so the complete step -- fused gather forward, MLP,
loss, backward including the HIP scatter, optimizer -- can be captured once in a `torch.cuda.CUDAGraph` (a
hipGraph) and replayed: 367 us per step at B = 1024, same losses, parameters equal to eager within float-atomic
reordering (tools/bench_train_step.py, tests/test_graph_capture_gpu.py).

Constraints while capturing (checked / arranged by `GraphedStep`): the out-of-range-id check must not read
back (`ops.set_index_check("off")` for the duration of the capture and of every replay's semantics: ids are
not validated inside a replay), dense table gradients only (`embeddings.sparse_grad` needs a host read per
step), optimizers constructed with `capturable=True`, and batches of ONE fixed shape (inputs are copied into
static tensors before each replay)."""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import ops


class GraphedStep:
    """`step_fn(batch) -> tensor` (e.g. zero_grad / forward / loss / backward / optimizer.step, returning the
    loss) captured once; `__call__(batch)` copies the batch into the static inputs, replays the graph and
    returns the (static) result tensor -- valid until the next call."""

    def __init__(self, step_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], example_batch: Dict[str, torch.Tensor],
                 warmup: int = 3):
        self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        prev = ops._INDEX_CHECK
        ops.flush_index_checks()
        ops.set_index_check("off")
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # allocator warm-up and lazy initialisation happen outside the graph
                for _ in range(max(1, warmup)):
                    step_fn(self._static)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._result = step_fn(self._static)
        finally:
            ops.set_index_check(prev)

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        for k, dst in self._static.items():
            if torch.is_tensor(dst):
                src = batch[k]
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError(f"GraphedStep: '{k}' is {tuple(src.shape)} {src.dtype}, captured {tuple(dst.shape)} {dst.dtype}")
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self._result
