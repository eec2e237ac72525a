"""Exact inner-product top-k retrieval on the GPU -- the reference's faiss wrapper, same surface.
Reference: src/model/model_utils/TopKSearcher.py (`__init__` :8-17, `update_embedding` :19-48,
`search` :50-84).  The item matrix stays resident in HBM; `search` is one HIP launch pair
(`nrx_topk_ip`, include/nrx_embed.h) instead of a host round trip through faiss.

Differences, all additive: `use_gpu` is accepted and ignored (there is no CPU path here);
`search_tensor` takes / returns device tensors and per-query exclusion lists (DSSM.hit_rate's history
filter) so a whole validation set is one call; ties resolve toward the lower item index (faiss leaves
the order of equal scores unspecified)."""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ... import ops


def exclusion_csr(lists: Sequence[Sequence[int]], device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-query python lists of item positions -> (offsets [Q+1], sorted positions) int64 on `device`."""
    offs = [0]
    flat: List[int] = []
    for l in lists:
        flat.extend(sorted(set(int(x) for x in l)))
        offs.append(len(flat))
    return (torch.tensor(offs, dtype=torch.int64, device=device),
            torch.tensor(flat, dtype=torch.int64, device=device))


class TopKSearcher:
    def __init__(self, k: int, use_gpu: bool = True):
        self.k = k
        self.index: Optional[torch.Tensor] = None      # [N, d] fp32 on the GPU (faiss: IndexFlatIP)
        self.use_gpu = use_gpu
        self.dimension: Optional[int] = None

    def update_embedding(self, emb_layer, normalize: bool = False, verbose: bool = True):
        """TopKSearcher.py:19-48.  `emb_layer`: nn.Embedding (as in the reference) or a [N, d] tensor."""
        w = emb_layer.weight if isinstance(emb_layer, nn.Embedding) else emb_layer
        w = w.detach().to(torch.float32)
        if not w.is_cuda:
            w = w.cuda()
        if normalize:
            w = _normalize_l2(w)
        self.index = w.contiguous()
        self.dimension = int(w.shape[1])
        if verbose:
            print(f"[TopKSearcher] Index updated. Size: {w.shape[0]}, Dim: {self.dimension}")

    def search_tensor(self, queries: torch.Tensor, k: Optional[int] = None, normalize: bool = False, exclude=None):
        if self.index is None:
            raise ValueError("Index not initialized. Please call update_embedding first.")
        q = queries.detach().to(device=self.index.device, dtype=torch.float32)
        if normalize:
            q = _normalize_l2(q)
        return ops.topk_ip(self.index, q, self.k if k is None else k, exclude=exclude)

    def search(self, query_embeddings: List[torch.Tensor], normalize: bool = False) -> Tuple[List[List[int]], List[List[float]]]:
        """TopKSearcher.py:50-84: list of [d] tensors in, python lists (indices, scores) out."""
        if self.index is None:
            raise ValueError("Index not initialized. Please call update_embedding first.")
        if len(query_embeddings) == 0:
            return [], []
        idx, score = self.search_tensor(torch.stack(query_embeddings), normalize=normalize)
        return idx.tolist(), score.tolist()


def _normalize_l2(x: torch.Tensor) -> torch.Tensor:
    """faiss.normalize_L2: x / ||x||_2 per row, rows of zero norm left untouched."""
    n = x.norm(dim=1, keepdim=True)
    return torch.where(n > 0, x / n.clamp_min(1e-30), x)
