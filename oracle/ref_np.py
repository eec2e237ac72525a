"""CPU ORACLE (numpy) -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.

A plain-numpy restatement of the reference's embedding / pooling / feature-interaction hot
path (ZhangHaoyang493/News_Recsys, paths below relative to /root/reference).  It exists so
that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
HIP path against the reference's arithmetic on machines where the reference itself is absent.
Nothing under news_recsys_amd/ may import this module.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/*.npz, which were produced by running the reference's own Python on
CPU (tests/golden/gen_golden.py).  The one exception is the DeepFM composition, for which the
reference has no implementation (only a documented config block,
documents/config_file_introduction.md:153-176): its parts (FM, Deep) are pinned, the
composition is "parity unpinned".

Integer / copy work (gather, concat, column split, routing) is bit-exact by construction;
floating reductions use float32 like the reference, in a possibly different order -- the
tolerances are stated in the tests.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------
# a1/a2  tables + row gather                      src/model/BaseModel/base_model.py:141-166, 262-271
# ----------------------------------------------------------------------------------------
def emb_table_name(feature: str, share: Dict[str, str]) -> str:
    """base_model.py:119-122 `_get_emb_feature_name`."""
    return share.get(feature, feature)


def gather_rows(table: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """nn.Embedding.__call__(idx.long()) (base_model.py:271): verbatim fp32 row copy.
    Out-of-range ids raise IndexError like torch on CPU."""
    idx = np.asarray(idx).astype(np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= table.shape[0]):
        raise IndexError("index out of range in self")
    return table[idx]


def dense_feature(value: np.ndarray) -> np.ndarray:
    """base_model.py:264-265: value.float().unsqueeze(1)."""
    return np.asarray(value).astype(F32)[:, None]


# ----------------------------------------------------------------------------------------
# a3  array_feature_pooling                        base_model.py:273-282
# ----------------------------------------------------------------------------------------
def array_pool(emb: np.ndarray, mask: Optional[np.ndarray]) -> np.ndarray:
    """mask None -> mean over L (padding included); else sum(emb*mask)/(sum(mask)+1e-8)."""
    emb = emb.astype(F32)
    if mask is None:
        return emb.mean(axis=1, dtype=F32)
    m = mask.astype(F32)[:, :, None]
    s = (emb * m).sum(axis=1, dtype=F32)
    den = m.sum(axis=1, dtype=F32) + F32(1e-8)
    return (s / den).astype(F32)


def array_pool_bwd(emb: np.ndarray, mask: Optional[np.ndarray], gout: np.ndarray) -> np.ndarray:
    """d out / d emb (autograd of base_model.py:273-282)."""
    B, L, D = emb.shape
    if mask is None:
        return np.broadcast_to((gout / F32(L))[:, None, :], (B, L, D)).astype(F32)
    m = mask.astype(F32)
    den = m.sum(axis=1, dtype=F32) + F32(1e-8)
    return ((gout / den[:, None])[:, None, :] * m[:, :, None]).astype(F32)


# ----------------------------------------------------------------------------------------
# a4  get_embeddings_from_batch                    base_model.py:284-308
# ----------------------------------------------------------------------------------------
class FeatureSpace:
    """The slice of the reference's config that the path reads (base_model.py:69-106)."""

    def __init__(self, sparse: Sequence[str], dense: Sequence[str], array: Sequence[str],
                 share: Optional[Dict[str, str]] = None):
        self.sparse = set(sparse)
        self.dense = set(dense)
        self.array = set(array)
        self.share = dict(share or {})

    @classmethod
    def from_yaml_dict(cls, cfg: dict) -> "FeatureSpace":
        f = cfg.get("features", {})
        e = cfg.get("embeddings", {})
        return cls(f.get("sparse_feature_names") or [], f.get("dense_feature_names") or [],
                   f.get("array_feature_names") or [], e.get("share_emb_table_features") or {})


def embed_concat(space: FeatureSpace, tables: Dict[str, np.ndarray], batch: Dict[str, np.ndarray],
                 feature_names) -> Tuple[np.ndarray, List[int], List[str]]:
    """base_model.py:284-308.  Returns (features[B, sum D], dims, names).

    The reference returns the UNFILTERED sorted name list even when a feature is missing from
    the batch (so `names` can be longer than `dims`); this restatement returns exactly that, and
    `names_used` = the filtered list is available as the 4th element of embed_concat_ex."""
    feats, dims, names, _ = embed_concat_ex(space, tables, batch, feature_names)
    return feats, dims, names


def embed_concat_ex(space, tables, batch, feature_names):
    sorted_features = sorted(list(feature_names))
    parts, dims, used = [], [], []
    for fname in sorted_features:
        if fname not in batch:
            continue
        val = batch[fname]
        if fname in space.dense:
            emb = dense_feature(val)
        else:
            tname = emb_table_name(fname, space.share)
            if tname not in tables:
                raise ValueError(f"Embedding table not found for {fname} (mapped to {tname})")
            emb = gather_rows(tables[tname], val)
        if fname in space.array:
            emb = array_pool(emb, batch.get(fname + "_mask"))
        parts.append(emb.astype(F32))
        dims.append(int(emb.shape[1]))
        used.append(fname)
    if not parts:
        return np.zeros((0,), F32), [], [], []
    return np.concatenate(parts, axis=1), dims, sorted_features, used


def embedding_grad_dense(idx: np.ndarray, upstream: np.ndarray, rows: int) -> np.ndarray:
    """Dense weight.grad of nn.Embedding(padding_idx=0) (base_model.py:164; a11):
    scatter-add of the upstream rows, row 0 forced to zero.  idx [...], upstream [..., D]."""
    D = upstream.shape[-1]
    g = np.zeros((rows, D), np.float64)
    np.add.at(g, np.asarray(idx).reshape(-1).astype(np.int64), upstream.reshape(-1, D).astype(np.float64))
    g[0] = 0.0
    return g.astype(F32)


# ----------------------------------------------------------------------------------------
# a5  FM                                            src/model/sort/fm/model.py:18-26, 48-59
# ----------------------------------------------------------------------------------------
def fm_split(features: np.ndarray, dims: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """fm/model.py:48-59: w = col 0 of every field, v = cols 1.. stacked [B, F, D-1]."""
    w, v, s = [], [], 0
    for d in dims:
        w.append(features[:, s:s + 1])
        v.append(features[:, s + 1:s + d])
        s += d
    if len({x.shape[1] for x in v}) != 1:
        raise RuntimeError("stack expects each tensor to be equal size")  # torch.stack would raise
    return np.concatenate(w, axis=1), np.stack(v, axis=1)


def fm_logit(w: np.ndarray, v: np.ndarray, bias) -> np.ndarray:
    """Pre-sigmoid FM output: bias + sum_f w + 0.5 * sum_k[(sum_f v)^2 - sum_f v^2]  -> [B,1]."""
    first = w.astype(F32).sum(axis=1, keepdims=True, dtype=F32)
    sv = v.astype(F32).sum(axis=1, dtype=F32)
    sq = (v.astype(F32) ** 2).sum(axis=1, dtype=F32)
    second = F32(0.5) * (sv * sv - sq).sum(axis=1, keepdims=True, dtype=F32)
    return (np.asarray(bias, F32).reshape(1, 1) + first + second).astype(F32)


def sigmoid(x: np.ndarray) -> np.ndarray:
    x = x.astype(F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def fm_forward(w, v, bias) -> np.ndarray:
    """fm/model.py:18-26 FMModel.forward."""
    return sigmoid(fm_logit(w, v, bias))


def fm_logit_bwd(w, v, glogit):
    """Gradients of fm_logit w.r.t. (w, v, bias) given d/dlogit [B,1]."""
    gw = np.broadcast_to(glogit, w.shape).astype(F32)
    sv = v.astype(F32).sum(axis=1, keepdims=True, dtype=F32)
    gv = (glogit[:, :, None] * (sv - v)).astype(F32)
    return gw, gv, glogit.sum(dtype=F32).reshape(1)


# ----------------------------------------------------------------------------------------
# a6  DCN v1 cross                                  src/model/sort/dcn/dcn_arch.py:5-30, 53-70
# ----------------------------------------------------------------------------------------
def dcn_v1_reference_form(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Literal form of DCNLayer.forward (dcn_arch.py:22-28): (x0 x_l^T) w + b + x_l, with the
    [B,D,D] outer product materialised.  w, b: [n_layers, D].  Small inputs only."""
    x0 = x.astype(F32)
    xl = x0
    for l in range(w.shape[0]):
        outer = x0[:, :, None] * xl[:, None, :]                 # B x D x D
        cross = outer @ w[l].astype(F32)[:, None]               # B x D x 1
        xl = (cross[:, :, 0] + b[l].astype(F32)[None, :] + xl).astype(F32)
    return xl


def dcn_v1(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Algebraically identical O(B*D) form: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l."""
    x0 = x.astype(F32)
    xl = x0
    for l in range(w.shape[0]):
        s = (xl * w[l].astype(F32)[None, :]).sum(axis=1, keepdims=True, dtype=F32)
        xl = (x0 * s + b[l].astype(F32)[None, :] + xl).astype(F32)
    return xl


def dcn_v1_bwd(x, w, b, gout):
    """Backward of dcn_v1: returns (gx, gw[n,D], gb[n,D])."""
    n = w.shape[0]
    x0 = x.astype(np.float64)
    xs, ss = [x0], []
    for l in range(n):
        s = (xs[-1] * w[l][None, :]).sum(axis=1, keepdims=True)
        ss.append(s)
        xs.append(x0 * s + b[l][None, :] + xs[-1])
    g = gout.astype(np.float64)
    gx0 = np.zeros_like(x0)
    gw = np.zeros((n, x.shape[1]))
    gb = np.zeros((n, x.shape[1]))
    for l in reversed(range(n)):
        gb[l] = g.sum(axis=0)
        gs = (g * x0).sum(axis=1, keepdims=True)
        gx0 += g * ss[l]
        gw[l] = (gs * xs[l]).sum(axis=0)
        g = g + gs * w[l][None, :]
    gx = g + gx0
    return gx.astype(F32), gw.astype(F32), gb.astype(F32)


# ----------------------------------------------------------------------------------------
# a7  DCN v2 cross                                  dcn_arch.py:33-50, 73-91
# ----------------------------------------------------------------------------------------
def dcn_v2(x: np.ndarray, W: np.ndarray, b: np.ndarray) -> np.ndarray:
    """DCNv2Net.forward: x <- relu(x0 * (x W_l^T + b_l) + x) per layer (ReLU after EVERY layer,
    dcn_arch.py:78-81).  W: [n, D, D] (nn.Linear weight, out x in), b: [n, D]."""
    x0 = x.astype(F32)
    xl = x0
    for l in range(W.shape[0]):
        lin = (xl @ W[l].astype(F32).T + b[l].astype(F32)[None, :]).astype(F32)
        xl = np.maximum(x0 * lin + xl, F32(0)).astype(F32)
    return xl


def dcn_v2_bwd(x, W, b, gout):
    n = W.shape[0]
    x0 = x.astype(np.float64)
    xs, lins, pre = [x0], [], []
    for l in range(n):
        lin = xs[-1] @ W[l].astype(np.float64).T + b[l][None, :]
        p = x0 * lin + xs[-1]
        lins.append(lin)
        pre.append(p)
        xs.append(np.maximum(p, 0.0))
    g = gout.astype(np.float64)
    gx0 = np.zeros_like(x0)
    gW = np.zeros(W.shape)
    gb = np.zeros(b.shape)
    for l in reversed(range(n)):
        g = g * (pre[l] > 0)
        glin = g * x0
        gx0 += g * lins[l]
        gW[l] = glin.T @ xs[l]
        gb[l] = glin.sum(axis=0)
        g = g + glin @ W[l].astype(np.float64)
    return (g + gx0).astype(F32), gW.astype(F32), gb.astype(F32)


# ----------------------------------------------------------------------------------------
# a8  Wide & Deep split                             src/model/sort/widedeep/model.py:24-27, 53-69
# ----------------------------------------------------------------------------------------
def wide_split(features: np.ndarray, dims: Sequence[int], names: Sequence[str], wide_names) -> Tuple[np.ndarray, np.ndarray]:
    wide, deep, s = [], [], 0
    for d, n in zip(dims, names):
        if n in wide_names:
            wide.append(features[:, s:s + 1])
            deep.append(features[:, s + 1:s + d])
        else:
            deep.append(features[:, s:s + d])
        s += d
    return np.concatenate(wide, axis=1), np.concatenate(deep, axis=1)


# ----------------------------------------------------------------------------------------
# a9  MLP heads / Deep / LR                         model_utils/utils.py:6-17, deep/model.py:12-21, lr/model.py:24-27
# ----------------------------------------------------------------------------------------
def mlp(x: np.ndarray, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray]) -> np.ndarray:
    """Linear+ReLU stack, no activation after the last layer (utils.py:9-14)."""
    h = x.astype(F32)
    for i, (W, bb) in enumerate(zip(weights, biases)):
        h = (h @ W.astype(F32).T + bb.astype(F32)[None, :]).astype(F32)
        if i < len(weights) - 1:
            h = np.maximum(h, F32(0))
    return h


def mlp_params(params: Dict[str, np.ndarray], prefix: str):
    """Collect `<prefix>.<2i>.weight/bias` of an nn.Sequential(Linear, ReLU, ...)."""
    ws, bs, i = [], [], 0
    while f"{prefix}.{i}.weight" in params:
        ws.append(params[f"{prefix}.{i}.weight"])
        bs.append(params[f"{prefix}.{i}.bias"])
        i += 2
    return ws, bs


def deep_forward(features, params) -> np.ndarray:
    """deep/model.py:20-21: sigmoid(MLP(x))."""
    return sigmoid(mlp(features, *mlp_params(params, "score_fc.network.network")))


def lr_forward(features) -> np.ndarray:
    """lr/model.py:24-27: sigmoid(sum over columns) -> shape [B] (not [B,1])."""
    return sigmoid(features.astype(F32).sum(axis=1, dtype=F32))


def widedeep_forward(wide_x, deep_x, params) -> np.ndarray:
    """widedeep/model.py:24-27."""
    wide_out = wide_x.astype(F32).sum(axis=1, keepdims=True, dtype=F32) + params["score_fc.bias"].astype(F32)
    deep_out = mlp(deep_x, *mlp_params(params, "score_fc.deep_network.network"))
    return sigmoid(wide_out + deep_out)


def dcn_model_forward(x, params, n_layers) -> Tuple[np.ndarray, np.ndarray]:
    """dcn/model.py:25-29: sigmoid(MLP(cat[x, cross(x)]))."""
    w = np.stack([params[f"score_fc.cross_net.cross_net.{l}.w"][:, 0] for l in range(n_layers)])
    b = np.stack([params[f"score_fc.cross_net.cross_net.{l}.b"][:, 0] for l in range(n_layers)])
    cross = dcn_v1(x, w, b)
    out = sigmoid(mlp(np.concatenate([x, cross], axis=1), *mlp_params(params, "score_fc.score_fc.network")))
    return out, cross


def deepfm_forward(features, dims, fm_bias, mlp_w, mlp_b) -> np.ndarray:
    """DeepFM = FM part (fm/model.py:18-26, pre-sigmoid) + Deep MLP (deep/model.py:12-21, pre-sigmoid)
    composed the way WideDeep composes wide+deep (widedeep/model.py:24-27).  PARITY UNPINNED:
    the reference has no DeepFM model."""
    w, v = fm_split(features, dims)
    return sigmoid(fm_logit(w, v, fm_bias) + mlp(features, mlp_w, mlp_b))


def bce_loss(pred: np.ndarray, label: np.ndarray) -> np.ndarray:
    """F.binary_cross_entropy(reduction='mean') (deep/model.py:32-33); log clamped at -100 like torch."""
    p = pred.reshape(-1).astype(np.float64)
    y = label.reshape(-1).astype(np.float64)
    lp = np.maximum(np.log(p), -100.0)
    l1p = np.maximum(np.log1p(-p), -100.0)
    return F32(-(y * lp + (1 - y) * l1p).mean())


# ----------------------------------------------------------------------------------------
# a10  DSSM                                         src/model/recall/DSSM/model.py:26-110, 148-180
# ----------------------------------------------------------------------------------------
def leaky_relu(x, slope=0.2):
    return np.where(x > 0, x, x * F32(slope)).astype(F32)


def dssm_tower(x, params, prefix) -> np.ndarray:
    """Linear-LReLU(.2) x3 + Linear (model.py:26-44)."""
    h = x.astype(F32)
    for i in (0, 2, 4, 6):
        h = (h @ params[f"{prefix}.{i}.weight"].T + params[f"{prefix}.{i}.bias"][None, :]).astype(F32)
        if i < 6:
            h = leaky_relu(h)
    return h


def l2_normalize(x, axis=-1, eps=1e-12):
    n = np.sqrt((x.astype(F32) ** 2).sum(axis=axis, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(eps))).astype(F32)


def dssm_tower_input(space, tables, batch, names) -> np.ndarray:
    """get_user_embedding / get_item_embedding (model.py:148-180) in SORTED feature order
    (the reference iterates a set).  Array features WITHOUT a mask are left
    un-pooled by the reference and would break torch.cat; only the masked form is covered."""
    parts = []
    for fname in sorted(names):
        if fname in space.dense:
            parts.append(dense_feature(batch[fname]))
            continue
        emb = gather_rows(tables[emb_table_name(fname, space.share)], batch[fname])
        if fname in space.array:
            m = batch[fname + "_mask"].astype(F32)
            emb = (emb * m[:, :, None]).sum(axis=1, dtype=F32) / (m.sum(axis=1, keepdims=True, dtype=F32) + F32(1e-8))
        parts.append(emb.astype(F32))
    return np.concatenate(parts, axis=1)


def dssm_negatives(raw_item_emb, perms) -> np.ndarray:
    """model.py:59-71 with explicit permutations instead of torch.randperm."""
    return l2_normalize(np.stack([raw_item_emb[p] for p in perms], axis=1), axis=-1)


def _log_softmax0(logits):
    m = logits.max(axis=1, keepdims=True)
    z = logits - m
    return z[:, 0] - np.log(np.exp(z).sum(axis=1))


def infonce_loss(u, pos, neg, temperature=0.1, mask=None):
    """model.py:92-110."""
    ps = (u * pos).sum(axis=1) / temperature
    ns = np.einsum("bd,bnd->bn", u, neg) / temperature
    losses = -_log_softmax0(np.concatenate([ps[:, None], ns], axis=1).astype(np.float64))
    if mask is not None:
        losses = losses * mask
    return F32(losses.mean())


def triplet_loss(u, pos, neg, margin=1.0, mask=None):
    """model.py:75-90.  NOTE the reference's shapes: pos_scores is [B] while neg_scores is
    unsqueezed to [B,1] (model.py:85), so `margin - pos + neg` BROADCASTS to a [B,B] matrix
    (entry [i,j] = margin - pos_j + neg_i) and the mask [B] multiplies along columns.  The mean is
    over all B*B entries.  Restated as written (pinned by tests/golden/model_dssm.npz)."""
    nn_ = neg.shape[1]
    ps = (u * pos).sum(axis=1) * nn_                            # [B]
    ns = np.einsum("bd,bnd->bn", u, neg).sum(axis=1)[:, None]   # [B,1]
    losses = np.maximum(margin - ps[None, :] + ns, 0.0)         # [B,B]
    if mask is not None:
        losses = losses * np.asarray(mask)[None, :]
    return F32(losses.mean())


# ----------------------------------------------------------------------------------------
# LR schedule                                        model_utils/lr_schedule.py:6-28
# ----------------------------------------------------------------------------------------
def cosine_decay_lr(step: int, lrs: Sequence[float], milestones: Sequence[int]) -> float:
    if step < milestones[0]:
        return lrs[0]
    if step >= milestones[-1]:
        return lrs[-1]
    progress = (step - milestones[0]) / max(1, milestones[1] - milestones[0])
    return lrs[1] + (lrs[0] - lrs[1]) * 0.5 * (1.0 + math.cos(math.pi * progress))


# ----------------------------------------------------------------------------------------
# integer utilities of the row-sharded path (new in the build) -- exact definitions
# ----------------------------------------------------------------------------------------
def owner_of(ids: np.ndarray, world: int) -> Tuple[np.ndarray, np.ndarray]:
    """Round-robin row sharding: row r lives on rank r % world at local row r // world."""
    ids = np.asarray(ids, np.int64)
    return ids % world, ids // world


def bucketize_by_owner(ids: np.ndarray, world: int):
    """Stable bucketing of a flat id list by owner rank.
    Returns (counts[world], perm) where perm lists source positions grouped by owner, in
    ascending source order inside each bucket; send buffer = (ids // world)[perm]."""
    owner, _ = owner_of(ids, world)
    perm = np.argsort(owner, kind="stable").astype(np.int64)
    counts = np.bincount(owner, minlength=world).astype(np.int64)
    return counts, perm


def csr_from_mask(mask: np.ndarray):
    """Padded [B, L] 0/1 mask -> CSR offsets[B+1] (valid = mask != 0) and flat positions."""
    valid = np.asarray(mask) != 0
    lens = valid.sum(axis=1).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pos = np.flatnonzero(valid.reshape(-1)).astype(np.int64)
    return offsets, pos


def csr_bag_to_padded(values: np.ndarray, offsets: np.ndarray, bag_len: int):
    """CSR bags -> the padded ids [B, L] + mask [B, L] that DataReader.__getitem__ builds per sample
    (src/dataset/DataReader/data_reader.py:96-109: short bags are 0-padded with mask 0, long ones cut to the first L).
    The definition of what a NRX_FEAT_BAG_CSR feature pools."""
    values, offsets = np.asarray(values), np.asarray(offsets, np.int64)
    B = len(offsets) - 1
    ids = np.zeros((B, bag_len), values.dtype if values.size else np.int64)
    mask = np.zeros((B, bag_len), np.float32)
    for b in range(B):
        n = min(int(offsets[b + 1] - offsets[b]), bag_len)
        ids[b, :n] = values[offsets[b]:offsets[b] + n]
        mask[b, :n] = 1.0
    return ids, mask


def unique_inverse(ids: np.ndarray):
    u, inv = np.unique(np.asarray(ids, np.int64), return_inverse=True)
    return u, inv.astype(np.int64)


def route_ids(id_arrays, world: int, cap: int):
    """Fixed-capacity routing (definition of nrx_route_ids).  id_arrays: list of integer arrays, one per
    feature, flattened feature-major.  On the wire a row is an int32 local row.  Returns (send_rows [world*cap] with unused slots = -1,
    slot [N] (-1 where a block overflowed), counts2d [world, F], max block count)."""
    flat = [np.asarray(a, np.int64).reshape(-1) for a in id_arrays]
    ids = np.concatenate(flat) if flat else np.zeros(0, np.int64)
    fid = np.concatenate([np.full(a.size, f, np.int64) for f, a in enumerate(flat)]) if flat else ids
    I32MAX = (1 << 31) - 1
    bad = (ids < 0) | (ids > I32MAX)                 # cannot be rows of any table: go to rank 0 as -1 / INT32_MAX
    owner = np.where(bad, 0, ids % world)
    F = len(flat)
    counts2d = np.bincount(owner * F + fid, minlength=world * F).reshape(world, F).astype(np.int64)
    send = np.full(world * cap, -1, np.int64)
    slot = np.full(ids.size, -1, np.int64)
    for o in range(world):
        pos = np.flatnonzero(owner == o)                 # ascending source order = stable
        k = np.arange(pos.size)
        ok = k < cap
        slot[pos[ok]] = o * cap + k[ok]
        loc = np.where(ids[pos] < 0, -1, np.where(ids[pos] > I32MAX, I32MAX, ids[pos] // world))
        send[o * cap + k[ok]] = loc[ok]
    return send, slot, counts2d, int(counts2d.sum(axis=1).max()) if world else 0


def route_feat(id_arrays, world: int, capf: int):
    """Per-feature fixed-capacity routing (definition of nrx_route_feat; new in this build -- the reference is single-device,
    src/model/sort/deep/train.py:38-44).  id_arrays: n integer arrays of B ids each.  Every (owner o, feature f) pair owns capf slots:
      send_ids [world, n, capf] int32   OWNER IDS in sample order: 0 = nothing (empty slot / the global padding id 0), v >= 1 = local row v - 1 of
                                        o's shard (id // world + 1); ids < 0 / >= 2^31 - 1 go to rank 0 as -1 / INT32_MAX
      send_pos [world, n, capf] int32   the sample of each sent id (-1 in the tails)
      slot     [n, B] int32             (o * capf + k) * n + f, or -1 where k >= capf
      counts   [world, n] int64         ids of f owned by o
    Returns (send_ids, send_pos, slot, counts, counts.max())."""
    n = len(id_arrays)
    B = int(np.asarray(id_arrays[0]).size) if n else 0
    I32MAX = (1 << 31) - 1
    send_ids = np.zeros((world, n, capf), np.int32)
    send_pos = np.full((world, n, capf), -1, np.int32)       # (-1 in the tails: an empty slot; owner id 0 WITH a position is a padding lookup)
    slot = np.full((n, B), -1, np.int32)
    counts = np.zeros((world, n), np.int64)
    for f, a in enumerate(id_arrays):
        ids = np.asarray(a, np.int64).reshape(-1)
        assert ids.size == B
        bad_lo, bad_hi = ids < 0, ids >= I32MAX
        owner = np.where(bad_lo | bad_hi, 0, ids % world)
        val = np.where(bad_lo, -1, np.where(bad_hi, I32MAX, np.where(ids == 0, 0, ids // world + 1)))
        for o in range(world):
            pos = np.flatnonzero(owner == o)                 # ascending sample order = stable
            counts[o, f] = pos.size
            k = np.arange(pos.size)
            ok = k < capf
            send_ids[o, f, k[ok]] = val[pos[ok]]
            send_pos[o, f, k[ok]] = pos[ok]
            slot[f, pos[ok]] = (o * capf + k[ok]) * n + f
    return send_ids, send_pos, slot, counts, int(counts.max()) if counts.size else 0


def owner_ids_from_inbox(inbox: np.ndarray) -> np.ndarray:
    """[world, n, capf] (what the equal-split all-to-all of route_feat's send_ids leaves on an owner: block s came from rank s) ->
    [n, world * capf]: per feature ONE array of owner ids over world * capf pseudo-samples b' = s * capf + k (definition of nrx_inbox_transpose)."""
    w, n, capf = inbox.shape
    return np.ascontiguousarray(inbox.transpose(1, 0, 2)).reshape(n, w * capf)


def gather_inbox(tables, feat_table, world: int, cap: int, recv2d, inbox_rows, dim: int):
    """Owner side (definition of nrx_gather_inbox): block s of the inbox holds sum_f recv2d[s, f] valid
    local rows, feature-major; slot p reads tables[feat_table[f]][inbox_rows[p]].  Unwritten slots = 0."""
    out = np.zeros((world * cap, dim), F32)
    for s_ in range(world):
        j = 0
        for f, n in enumerate(np.asarray(recv2d).reshape(world, -1)[s_]):
            n = int(n)
            take = max(0, min(n, cap - j))
            if take:
                rows = inbox_rows[s_ * cap + j: s_ * cap + j + take]
                out[s_ * cap + j: s_ * cap + j + take] = gather_rows(tables[feat_table[f]], rows)
            j += n
    return out


def route_ids_dedup(id_arrays, table_of, table_local_rows, world: int, cap: int):
    """Definition of nrx_route_ids_dedup: every distinct (owner, table, local row) is sent once; inside owner o's block
    the unique rows are ordered by (table, row).  Ids that cannot be rows go to rank 0 as `table_local_rows[t]` (one past
    the largest shard).  Returns (send_rows [world*cap] unused = -1, slot [N] (-1 on overflow), counts2d [world, n_tables],
    largest block's unique count)."""
    flat = [np.asarray(a, np.int64).reshape(-1) for a in id_arrays]
    ids = np.concatenate(flat) if flat else np.zeros(0, np.int64)
    tab = np.concatenate([np.full(a.size, table_of[f], np.int64) for f, a in enumerate(flat)]) if flat else ids
    lr = np.asarray(table_local_rows, np.int64)[tab] if ids.size else ids
    I32MAX = (1 << 31) - 1
    bad = (ids < 0) | (ids > I32MAX)
    owner = np.where(bad, 0, ids % world)
    local = np.where(bad, lr, np.minimum(ids // world, lr))
    nt = len(table_local_rows)
    send = np.full(world * cap, -1, np.int64)
    slot = np.full(ids.size, -1, np.int64)
    counts2d = np.zeros((world, nt), np.int64)
    worst = 0
    for o in range(world):
        pos = np.flatnonzero(owner == o)
        keys = tab[pos] * (1 << 32) + local[pos]
        uniq, inv = np.unique(keys, return_inverse=True)
        worst = max(worst, uniq.size)
        ok = inv < cap
        slot[pos[ok]] = o * cap + inv[ok]
        k = np.arange(uniq.size)
        send[o * cap + k[k < cap]] = (uniq & ((1 << 32) - 1))[k < cap]
        counts2d[o] = np.bincount(uniq >> 32, minlength=nt)
    return send, slot, counts2d, worst


def bag_norm_weights(mask, batch: int, bag_len: int, kind: str):
    """Definition of nrx_bag_norm_weights: pooling weights with the normalisation folded in.
    kind 'masked_mean': w / (sum_l w + 1e-8) (array_feature_pooling, base_model.py:278-282); 'mean': 1 / L (:275-276);
    'sum': w (or 1)."""
    if kind == "mean":
        return np.full((batch, bag_len), 1.0, F32) / F32(bag_len)
    m = np.ones((batch, bag_len), F32) if mask is None else np.asarray(mask, F32)
    if kind == "sum":
        return m
    den = m.sum(axis=1, dtype=F32, keepdims=True) + F32(1e-8)
    return (m / den).astype(F32)


def route_bags(id_arrays, weights, world: int, cap: int):
    """Definition of nrx_route_bags (pooled-bag channel): id_arrays[f] is [B, L_f]; lookups with weight != 0 go, in
    source order (feature-major, then sample, then position), to their owner's block as (local row, tag = f*B + sample,
    weight).  Returns (send_rows, send_tag, send_w [world*cap] (unused slots -1 / -1 / 0), counts2d [world, F], max block)."""
    F = len(id_arrays)
    B = id_arrays[0].shape[0] if F else 0
    send = np.full(world * cap, -1, np.int64)
    tag = np.full(world * cap, -1, np.int64)
    sw = np.zeros(world * cap, F32)
    counts2d = np.zeros((world, F), np.int64)
    fill = np.zeros(world, np.int64)
    I32MAX = (1 << 31) - 1
    for f, ids in enumerate(id_arrays):
        ids = np.asarray(ids, np.int64)
        L = ids.shape[1]
        w = np.ones((B, L), F32) if weights[f] is None else np.asarray(weights[f], F32)
        flat, wf = ids.reshape(-1), w.reshape(-1)
        keep = wf != 0
        bad = (flat < 0) | (flat > I32MAX)
        owner = np.where(bad, 0, flat % world)
        loc = np.where(flat < 0, -1, np.where(flat > I32MAX, I32MAX, flat // world))
        smp = np.arange(flat.size) // L
        for o in range(world):
            pos = np.flatnonzero(keep & (owner == o))
            counts2d[o, f] = pos.size
            k = fill[o] + np.arange(pos.size)
            ok = k < cap
            send[o * cap + k[ok]] = loc[pos[ok]]
            tag[o * cap + k[ok]] = f * B + smp[pos[ok]]
            sw[o * cap + k[ok]] = wf[pos[ok]]
            fill[o] += pos.size
    return send, tag, sw, counts2d, int(fill.max()) if world else 0


def route_bags_runs(id_arrays, weights, world: int, cap: int):
    """Definition of nrx_route_bags_runs' send_run: run[o, tag] = (first slot, one past the last slot) of tag's entries inside owner o's block
    under route_bags' order (feature-major, then sample, then position: the entries of a tag are contiguous there), both CLAMPED to cap (a run
    that overflows the block ends at cap; one that lies beyond it is the empty run (cap, cap)); (0, 0) for a tag without an entry."""
    F = len(id_arrays)
    B = id_arrays[0].shape[0] if F else 0
    run = np.zeros((world, F * B, 2), np.int32)
    fill = np.zeros(world, np.int64)
    I32MAX = (1 << 31) - 1
    for f, ids in enumerate(id_arrays):
        ids = np.asarray(ids, np.int64)
        L = ids.shape[1]
        w = np.ones((B, L), F32) if weights[f] is None else np.asarray(weights[f], F32)
        flat, wf = ids.reshape(-1), w.reshape(-1)
        keep = wf != 0
        bad = (flat < 0) | (flat > I32MAX)
        owner = np.where(bad, 0, flat % world)
        smp = np.arange(flat.size) // L
        for o in range(world):
            pos = np.flatnonzero(keep & (owner == o))
            t = f * B + smp[pos]
            if t.size:
                first = np.flatnonzero(np.r_[True, t[1:] != t[:-1]])
                last = np.r_[first[1:], t.size]
                run[o, t[first], 0] = np.minimum(fill[o] + first, cap)
                run[o, t[first], 1] = np.minimum(fill[o] + last, cap)
            fill[o] += pos.size
    return run


def tags_from_runs(run, world: int, cap: int):
    """Definition of nrx_pool_inbox_runs_words' tag_out: block s of `run` ([world, n_tags, 2], as it ARRIVED at the owner) names, per tag, the
    slots of block s that hold its entries; every slot inside a run gets its tag, the others stay -1."""
    run = np.asarray(run).reshape(world, -1, 2)
    tag = np.full(world * cap, -1, np.int64)
    for s_ in range(world):
        for t in np.flatnonzero(run[s_, :, 1] > run[s_, :, 0]):
            tag[s_ * cap + run[s_, t, 0]: s_ * cap + run[s_, t, 1]] = t
    return tag


def pool_inbox(tables, feat_table, batch: int, world: int, cap: int, recv2d, inbox_rows, inbox_tag, inbox_w, dim: int):
    """Definition of nrx_pool_inbox_fwd: partial[s, tag] = sum over block s's valid entries with that tag of
    w * tables[feat_table[tag // batch]][row], accumulated in entry order (fp32, product then add)."""
    nf = len(feat_table)
    out = np.zeros((world, nf * batch, dim), F32)
    r2 = np.asarray(recv2d).reshape(world, -1)
    for s_ in range(world):
        total = min(int(r2[s_].sum()), cap)
        for j in range(total):
            t = int(inbox_tag[s_ * cap + j])
            row = int(inbox_rows[s_ * cap + j])
            tab = tables[feat_table[t // batch]]
            if row < 0 or row >= tab.shape[0]:
                raise IndexError("routed bag lookup out of range")
            out[s_, t] = out[s_, t] + tab[row] * F32(inbox_w[s_ * cap + j])
    return out


# ----------------------------------------------------------------------------------------
# validation metrics                               src/model/BaseModel/base_model.py:320-528
# ----------------------------------------------------------------------------------------
def auc_ties(scores: np.ndarray, labels: np.ndarray) -> float:
    """roc_auc_score for binary labels (what sklearn computes: trapezoid = Mann-Whitney with ties at 0.5)."""
    s = np.asarray(scores, np.float64)
    y = np.asarray(labels, np.float64)
    P, N = float((y == 1).sum()), float((y != 1).sum())
    order = np.argsort(-s, kind="stable")
    s, y = s[order], y[order]
    num, neg_above, i = 0.0, 0.0, 0
    while i < s.size:
        j = i
        while j < s.size and s[j] == s[i]:
            j += 1
        p = float((y[i:j] == 1).sum())
        q = float(j - i) - p
        num += p * (N - neg_above - q + 0.5 * q)
        neg_above += q
        i = j
    return num / (P * N)


def validation_metrics(user_ids, scores, labels, warm_users, k: int = 10):
    """Restatement of BaseModel.on_validation_epoch_end (base_model.py:333-492) over flat arrays in the
    order validation_step appended them (:320-330).  Returns the same nested dict as its `results`."""
    groups = {}
    for u, s, y in zip(user_ids, scores, labels):
        groups.setdefault(int(u), []).append((float(s), float(y)))
    warm = set(int(u) for u in warm_users) if warm_users is not None else None
    lists = {g: {"auc": [], "ndcg": [], "hr": [], "mrr": [], "p": [], "y": []} for g in ("Overall", "Warm_Start", "Cold_Start")}
    for uid, items in groups.items():
        is_cold = bool(warm) and uid not in warm          # :355-359 (an empty / missing set means nobody is cold)
        tgt = lists["Cold_Start" if is_cold else "Warm_Start"]
        preds = [x[0] for x in items]
        labs = [x[1] for x in items]
        for dst in (lists["Overall"], tgt):
            dst["p"].extend(preds)
            dst["y"].extend(labs)
        if len(set(labs)) > 1:
            a = auc_ties(np.array(preds), np.array(labs))
            lists["Overall"]["auc"].append(a)
            tgt["auc"].append(a)
        top = sorted(items, key=lambda x: x[0], reverse=True)[:k]      # stable: ties keep arrival order
        npos = sum(1 for x in items if x[1] == 1)
        if npos == 0:
            vals = (0.0, 0.0, 0.0)
        else:
            hr = 1.0 if any(x[1] == 1 for x in top) else 0.0
            dcg = sum(1.0 / np.log2(r + 1) for r, (_, y) in enumerate(top, start=1) if y == 1)
            idcg = sum(1.0 / np.log2(r + 1) for r in range(1, min(npos, k) + 1))
            mrr = next((1.0 / r for r, (_, y) in enumerate(top, start=1) if y == 1), 0.0)
            vals = (hr, dcg / idcg if idcg > 0 else 0.0, mrr)
        for dst in (lists["Overall"], tgt):
            dst["hr"].append(vals[0])
            dst["ndcg"].append(vals[1])
            dst["mrr"].append(vals[2])

    def auc_logloss(p, y):
        if not p:
            return 0.0, 0.0
        p, y = np.array(p, np.float32), np.array(y, np.float32)
        auc = auc_ties(p, y) if len(set(y.tolist())) > 1 else 0.0
        # As written in the reference (:445-449): the scores are float32, so the clip's upper bound
        # 1 - 1e-15 rounds to exactly 1.0 and a score of 1.0 gives log(0): LogLoss = inf / nan.
        with np.errstate(divide="ignore", invalid="ignore"):
            pc = np.clip(p, 1e-15, 1 - 1e-15)
            ll = -np.mean(y * np.log(pc) + (1 - y) * np.log(1 - pc))
        return auc, float(ll)

    mean = lambda l: float(np.mean(l)) if l else 0.0
    res = {}
    for g, d in lists.items():
        auc, ll = auc_logloss(d["p"], d["y"])
        res[g] = {"AUC": auc, "LogLoss": ll, "GAUC": mean(d["auc"]), f"NDCG@{k}": mean(d["ndcg"]),
                  f"HR@{k}": mean(d["hr"]), f"MRR@{k}": mean(d["mrr"])}
        if g != "Overall":
            res[g]["User_Count"] = len(d["hr"])
    return res


# ---------------------------------------------------------------------------------------------------
# Recall evaluation: exact inner-product top-k + hit rate
# ---------------------------------------------------------------------------------------------------
FLT_MAX = float(np.finfo(np.float32).max)


def topk_ip(items: np.ndarray, queries: np.ndarray, k: int, exclude=None):
    """faiss.IndexFlatIP(d).add(items); .search(queries, k) as TopKSearcher.search wraps it
    (src/model/model_utils/TopKSearcher.py:50-84): exhaustive inner products, the k largest per query in
    decreasing order, (-1, -FLT_MAX) where fewer than k items qualify.  faiss is absent from the
    reference tree and this image (parity with faiss itself is UNPINNED; anchored on the call sites):
    its tie order / summation order are unspecified, this restatement takes ties toward the lower index
    and accumulates in float64 (the C oracle and the HIP kernel use the same fp32 fma chain; they agree with this
    to ~1e-6 and exactly on the order wherever scores are separated by more than that).
    exclude: optional list (per query) of item positions that must not be returned = the reference's
    over-fetch-and-filter of the user's history (recall/DSSM/model.py:209-221)."""
    items = np.asarray(items, np.float32)
    queries = np.asarray(queries, np.float32)
    Q = queries.shape[0]
    idx = np.full((Q, k), -1, np.int64)
    score = np.full((Q, k), -FLT_MAX, np.float32)
    if items.shape[0] == 0:
        return idx, score
    s = queries.astype(np.float64) @ items.astype(np.float64).T
    for q in range(Q):
        row = s[q].copy()
        keep = np.ones(items.shape[0], bool)
        if exclude is not None and len(exclude[q]):
            keep[np.asarray(exclude[q], np.int64)] = False
        cand = np.nonzero(keep)[0]
        order = cand[np.argsort(-row[cand], kind="stable")][:k]
        idx[q, :len(order)] = order
        score[q, :len(order)] = row[order].astype(np.float32)
    return idx, score


def hit_rate_reference_loop(items: np.ndarray, queries: np.ndarray, targets: np.ndarray, histories, k: int) -> float:
    """DSSM.hit_rate as written (recall/DSSM/model.py:182-228), one query at a time: search k + len(history),
    drop the history, keep the first k, count the target.  `histories[q]` = item positions the user already
    interacted with; `targets[q]` = position of the held-out item."""
    hits = 0
    for q in range(queries.shape[0]):
        h = set(int(x) for x in histories[q])
        I, _ = topk_ip(items, queries[q:q + 1], min(k + len(h), items.shape[0]))
        filtered = []
        for it in I[0]:
            if it >= 0 and int(it) not in h:
                filtered.append(int(it))
            if len(filtered) >= k:
                break
        hits += int(int(targets[q]) in filtered)
    return hits / queries.shape[0] if queries.shape[0] else 0


def sparse_plan(id_arrays, table_of, rows, n_tables: int):
    """Definition of nrx_sparse_plan (planning step of the row-sparse backward; groups the lookups that
    autograd of nn.Embedding, base_model.py:262-308, would scatter-add): flat feature-major lookup p of feature
    f gets key (table_of[f], row) with out-of-range ids on row 0; `order` = stable argsort of the keys;
    uniq_keys = distinct (table << 40 | row) ascending; seg_start = first sorted position of each; counts =
    [n_unique, first unique index with table >= t for t in 0..n_tables]."""
    keys = []
    for ids, t, r in zip(id_arrays, table_of, rows):
        ids = np.asarray(ids, np.int64).reshape(-1)
        ids = np.where((ids < 0) | (ids >= r), 0, ids)
        keys.append((np.int64(t) << np.int64(40)) | ids)
    keys = np.concatenate(keys) if keys else np.zeros(0, np.int64)
    order = np.argsort(keys, kind="stable").astype(np.int64)
    sk = keys[order]
    head = np.ones(len(sk), bool)
    head[1:] = sk[1:] != sk[:-1]
    uniq = sk[head]
    seg = np.concatenate([np.nonzero(head)[0], [len(sk)]]).astype(np.int64)
    counts = np.array([len(uniq)] + [int(np.searchsorted(uniq, np.int64(t) << np.int64(40))) for t in range(n_tables + 1)], np.int64)
    return order, uniq, seg, counts


def sparse_plan_place(id_arrays, table_of, rows, n_tables: int, place_feats=None):
    """Definition of nrx_sparse_plan_place: sparse_plan plus the PLACEMENT of the rows that need no reduction.  A unique
    (table, row) looked up exactly once in the launch, not the padding row, by a feature listed in `place_feats` (default:
    all), is `placed`: dest[p] = its unique index for that lookup p; every other lookup has dest -1.  walk = the
    unique indices that are not placed (several lookups, a non-placeable feature's lookup, or row 0), ascending.
    The backward of nn.Embedding (base_model.py:262-308) for a placed row is the lookup's upstream row itself."""
    order, uniq, seg, counts = sparse_plan(id_arrays, table_of, rows, n_tables)
    lens = [int(np.asarray(x).size) for x in id_arrays]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = int(off[-1])
    ok = np.ones(len(lens), bool) if place_feats is None else np.array([f in set(place_feats) for f in range(len(lens))], bool)
    feat_of = np.searchsorted(off, np.arange(n), side="right") - 1 if n else np.zeros(0, np.int64)
    dest = np.full(n, -1, np.int32)
    seg_len = seg[1:] - seg[:-1]
    row = uniq & ((np.int64(1) << np.int64(40)) - 1)
    first = order[seg[:-1]] if n else np.zeros(0, np.int64)
    placed = (seg_len == 1) & (row != 0)
    if n:
        placed &= ok[feat_of[first]]
    dest[first[placed]] = np.nonzero(placed)[0].astype(np.int32)
    walk = np.nonzero(~placed)[0].astype(np.int32)
    return order, uniq, seg, counts, dest, walk


def sparse_plan_pairs(id_arrays, table_of, rows, n_tables: int):
    """Definition of nrx_sparse_plan_lds (the one-kernel planner; same role as sparse_plan_place for the autograd of nn.Embedding,
    base_model.py:262-308): uniq / counts as sparse_plan; per row with unique index u
      looked up once (not the padding row)   -> dest[p] = u for its lookup p
      looked up exactly twice (not padding)  -> dest = -1 for both lookups, one PAIR record (u, p1, p2), p1 < p2
      otherwise (3+ lookups, or the padding row) -> dest = -1; u is on the walk list, its lookups (ascending) are walk_lookups[u].
    Returns (uniq, counts, dest, pairs: int32 [k, 3] ascending by u, walk: int32 ascending, walk_lookups: dict u -> int64 array)."""
    order, uniq, seg, counts = sparse_plan(id_arrays, table_of, rows, n_tables)
    n = len(order)
    dest = np.full(n, -1, np.int32)
    seg_len = seg[1:] - seg[:-1]
    row = uniq & ((np.int64(1) << np.int64(40)) - 1)
    pairs, walk, walk_lookups = [], [], {}
    for u in range(len(uniq)):
        ps = order[seg[u]:seg[u + 1]]
        if row[u] != 0 and seg_len[u] == 1:
            dest[ps[0]] = u
        elif row[u] != 0 and seg_len[u] == 2:
            pairs.append((u, int(ps[0]), int(ps[1])))
        else:
            walk.append(u)
            walk_lookups[u] = np.asarray(ps, np.int64)
    return uniq, counts, dest, np.asarray(pairs, np.int32).reshape(-1, 3), np.asarray(walk, np.int32), walk_lookups
