"""On-device validation metrics.

The reference accumulates `(score, label)` tuples per user in a Python dict every validation step
(base_model.py:320-330: a device->host copy and a per-sample Python loop per batch) and computes
per-user sklearn AUC / NDCG@10 / HR@10 / MRR@10, a warm/cold split by `train_user_ids.json`, and global
AUC / LogLoss in Python at epoch end (:333-492), then prints / appends a fixed-format block to
`val_log.log` (:494-528, parsed by scripts/log_analysis.py:16-23).

Here the epoch's samples stay on the GPU: two stable sorts order them by (user, score desc, arrival),
ONE kernel (`nrx_user_rank_metrics`, one thread per user segment) yields the per-user metrics in fp64,
and the global AUC (ties at 1/2, like sklearn) / LogLoss are a handful of fp64 tensor reductions.
`ranking_metrics` returns the same nested dict as the reference's `results`; `format_val_log` renders
the same text."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from . import _lib
from .ops import _dev, _stream_ptr


def _auc_ties(scores: torch.Tensor, labels: torch.Tensor) -> float:
    """Global AUC with ties counted 1/2 (what sklearn.roc_auc_score returns); 0.0 when one class only."""
    y = (labels == 1).to(torch.float64)
    P = y.sum()
    N = y.numel() - P
    if y.numel() == 0 or P.item() == 0 or N.item() == 0:
        return 0.0
    s, order = torch.sort(scores, descending=True, stable=True)
    y = y[order]
    _, counts = torch.unique_consecutive(s, return_counts=True)
    ends = torch.cumsum(counts, 0)
    cy = torch.cumsum(y, 0)
    pos_upto = cy[ends - 1]
    p = torch.diff(pos_upto, prepend=pos_upto.new_zeros(1))
    q = counts.to(torch.float64) - p
    neg_above = torch.cumsum(q, 0) - q
    num = (p * (N - neg_above - 0.5 * q)).sum()
    return float((num / (P * N)).item())


def _auc_logloss(scores: torch.Tensor, labels: torch.Tensor):
    if scores.numel() == 0:
        return 0.0, 0.0
    auc = _auc_ties(scores, labels)
    # float32 on purpose: the reference clips float32 scores with np.clip(p, 1e-15, 1 - 1e-15)
    # (base_model.py:445-449), whose upper bound rounds to 1.0 -- a score of exactly 1.0 yields inf / nan there.
    p = scores.float().clamp(1e-15, 1 - 1e-15)
    y = labels.float()
    ll = float((-(y * torch.log(p) + (1 - y) * torch.log(1 - p)).mean()).item())
    return auc, ll


def _nanmean(x: torch.Tensor) -> float:
    ok = ~torch.isnan(x)
    return float(x[ok].mean().item()) if bool(ok.any()) else 0.0


def ranking_metrics(user_ids: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor,
                    warm_users: Optional[Iterable[int]] = None, k: int = 10) -> Dict[str, Dict[str, float]]:
    """Metrics of one validation epoch from flat device tensors (in arrival order)."""
    lib = _lib.load()
    _dev(scores, "scores")
    uid = user_ids.reshape(-1).long()
    sc = scores.reshape(-1).float().contiguous()
    lb = labels.reshape(-1).float().contiguous()
    dev = sc.device
    # (user, score desc, arrival order): stable sort by score, then stable sort by user
    o1 = torch.sort(sc, descending=True, stable=True).indices
    o2 = torch.sort(uid[o1], stable=True).indices
    order = o1[o2]
    s_uid, s_sc, s_lb = uid[order], sc[order].contiguous(), lb[order].contiguous()
    users, counts = torch.unique_consecutive(s_uid, return_counts=True)
    nu = users.numel()
    seg = torch.zeros(nu + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=seg[1:])
    outs = [torch.empty(nu, dtype=torch.float64, device=dev) for _ in range(4)]
    _lib.check(lib.nrx_user_rank_metrics(s_sc.data_ptr(), s_lb.data_ptr(), seg.data_ptr(), nu, k,
                                         *[o.data_ptr() for o in outs], _stream_ptr(sc)), "nrx_user_rank_metrics")
    auc, ndcg, hr, mrr = outs
    warm_list = list(warm_users) if warm_users is not None else []
    if warm_list:          # an empty / missing train-user set means nobody is cold (base_model.py:355-359)
        cold_u = ~torch.isin(users, torch.tensor(warm_list, dtype=torch.int64, device=dev))
    else:
        cold_u = torch.zeros(nu, dtype=torch.bool, device=dev)
    cold_s = torch.repeat_interleave(cold_u, counts)

    def block(umask, smask, with_count):
        a, ll = _auc_logloss(s_sc[smask], s_lb[smask]) if smask is not None else _auc_logloss(s_sc, s_lb)
        sel = (lambda t: t[umask]) if umask is not None else (lambda t: t)
        n_sel = int(umask.sum().item()) if umask is not None else nu
        mean = (lambda t: float(sel(t).mean().item()) if n_sel else 0.0)
        d = {"AUC": a, "LogLoss": ll, "GAUC": _nanmean(sel(auc)) if n_sel else 0.0, f"NDCG@{k}": mean(ndcg),
             f"HR@{k}": mean(hr), f"MRR@{k}": mean(mrr)}
        if with_count:
            d["User_Count"] = n_sel
        return d

    return {"Overall": block(None, None, False), "Warm_Start": block(~cold_u, ~cold_s, True),
            "Cold_Start": block(cold_u, cold_s, True)}


def format_val_log(results: Dict[str, Dict[str, float]], epoch: int, k: int = 10) -> str:
    """The text block the reference prints and appends to val_log.log (base_model.py:494-519)."""
    def sec(r):
        return (f"  AUC:      {r['AUC']:.4f}\n  LogLoss:  {r['LogLoss']:.4f}\n  GAUC:     {r['GAUC']:.4f}\n"
                f"  NDCG@{k}:  {r[f'NDCG@{k}']:.4f}\n  HR@{k}:    {r[f'HR@{k}']:.4f}\n  MRR@{k}:   {r[f'MRR@{k}']:.4f}\n")
    return (f"\n{'=' * 20} Epoch {epoch} Validation Results {'=' * 20}\n"
            f"Overall:\n{sec(results['Overall'])}"
            f"Warm Start Users ({results['Warm_Start']['User_Count']}):\n{sec(results['Warm_Start'])}"
            f"Cold Start Users ({results['Cold_Start']['User_Count']}):\n{sec(results['Cold_Start'])}"
            f"{'=' * 60}\n")
