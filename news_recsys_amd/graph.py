"""HIP-graph capture of a whole step (inference or training) of a model built on the fused path.

Small batches -- the reference's YAMLs train with a few hundred to a few thousand samples per step -- are
launch-bound: a Deep training step is ~40 kernel launches plus Python glue (1.6 ms per step at B = 1024 on an
MI355X host) for well under 0.4 ms of GPU work.  The fused HIP launches only enqueue kernels on torch's current
stream and allocate through torch's caching allocator, so the complete step -- fused gather forward, MLP,
loss, backward including the HIP scatter, optimizer -- can be captured once in a `torch.cuda.CUDAGraph` (a
hipGraph) and replayed: 367 us per step at B = 1024, same losses, parameters equal to eager within float-atomic
reordering (tools/bench_train_step.py, tests/test_graph_capture_gpu.py).

Constraints while capturing (checked / arranged by `GraphedStep`): optimizers constructed with `capturable=True`, batches of ONE fixed
shape (inputs are copied into static tensors before each replay), and `embeddings.sparse_grad: true` (COO gradients) is out: it reads the
unique-row count back every step (`fused` and the default dense-gradient mode do not).

The reference's error contract survives the capture (round 4): torch on CPU raises IndexError for an out-of-range id
(src/model/BaseModel/base_model.py:271), and so does a replay -- one call late.  The step is captured in the DEFERRED check mode: every
gather kernel records an offending id in a host-mapped status word (no read-back, nothing for the graph to replay but the kernel itself); the
next `__call__`, `check()` or `ops.flush_index_checks()` reads that word on the host and raises IndexError naming the feature.  With
`deterministic=True` the dense table gradients come from a deterministic reduction at ANY batch size -- the one-launch small kernel
(nrx_embed_bwd_small) where every table of the launch takes <= 4096 lookups, else the planned sorted reduction (planned inline, count read on the
device): replays are bit-reproducible run to run, where the float-atomic scatter is not.  The same switch puts the weight gradients this package
computes over the batch (DCN-v1 / DCN-v2 cross layers, `ops.linear`) in their ordered mode (per-slice / per-block partial sums added in a fixed
order by a second launch, `ops.WGRAD_ORDERED`); a capture that would still need float atomics somewhere is refused."""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import ops


class GraphedStep:
    """`step_fn(batch) -> tensor` (e.g. zero_grad / forward / loss / backward / optimizer.step, returning the
    loss) captured once; `__call__(batch)` copies the batch into the static inputs, replays the graph and
    returns the (static) result tensor -- valid until the next call."""

    def __init__(self, step_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], example_batch: Dict[str, torch.Tensor],
                 warmup: int = 3, deterministic: bool = False):
        self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        prev, prev_sorted, prev_wgrad = ops._INDEX_CHECK, ops.DENSE_BWD_SORTED, ops.WGRAD_ORDERED
        ops.flush_index_checks()
        ops.set_index_check("deferred")            # the check stays ON inside the graph: a status word the kernels write, read on the host later
        if deterministic:
            ops.DENSE_BWD_SORTED = "det"           # the choice is baked into the captured launches: the one-launch deterministic kernel
                                                   # where it applies (tables fed by <= 4096 lookups each), else the planned reduction
            ops.WGRAD_ORDERED = True               # ... and the weight gradients of DCN-v2 layers / ops.linear add their batch slices in a fixed order
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # allocator warm-up and lazy initialisation happen outside the graph
                for _ in range(max(1, warmup)):
                    step_fn(self._static)
            torch.cuda.current_stream().wait_stream(side)
            ops.flush_index_checks()               # (the example batch must be clean)
            del ops._host_status_names[:]
            self.graph = torch.cuda.CUDAGraph()
            atomic_before = ops.dense_bwd_paths["atomic"]
            # captured on the stream the warm-up steps ran on: whatever those steps set up per stream (the one-kernel planner's control block,
            # ops._lds_state) is what the captured launches use
            with torch.cuda.graph(self.graph, stream=side):
                self._result = step_fn(self._static)
            self._names = list(ops._host_status_names)      # the feature-name lists of the captured gather launches
            if deterministic and ops.dense_bwd_paths["atomic"] != atomic_before:
                raise RuntimeError("GraphedStep(deterministic=True): a backward launch of the captured step took float atomics (an embedding launch "
                                   "with more than 64 tables or a feature on a routed-row buffer, or a DCN-v1 stack beyond the ordered mode): its replays "
                                   "would not be bit-reproducible")
        finally:
            ops.set_index_check(prev)
            ops.DENSE_BWD_SORTED = prev_sorted
            ops.WGRAD_ORDERED = prev_wgrad

    def check(self) -> None:
        """Raise IndexError if a replay so far met an out-of-range id (synchronises the device)."""
        ops._remember_status_names(self._names)
        ops.flush_index_checks()

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        dsts, srcs = [], []
        for k, dst in self._static.items():
            if torch.is_tensor(dst):
                src = batch[k]
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError(f"GraphedStep: '{k}' is {tuple(src.shape)} {src.dtype}, captured {tuple(dst.shape)} {dst.dtype}")
                if src.device != dst.device:
                    dst.copy_(src, non_blocking=True)          # host batches: one H2D copy each
                else:
                    dsts.append(dst)
                    srcs.append(src)
        if dsts:
            torch._foreach_copy_(dsts, srcs)                   # device batches: ONE multi-tensor launch per dtype, not one copy per feature
                                                               # (26 copy_ calls were 135 us of host time per step at the C2 feature set)
        if ops.deferred_index_error_pending():     # recorded by an earlier replay (or launch): raise now, like the reference would have then
            self.check()
        self.graph.replay()
        ops._remember_status_names(self._names)
        return self._result
