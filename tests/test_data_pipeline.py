"""Input pipeline against goldens produced by the REFERENCE DataReader + torch
default_collate (tests/golden/gen_golden.py:gen_datareader): the text reader mirror and the columnar
loader must both reproduce the reference's batches exactly (ids, masks, float64 dense values, labels)."""
import os
import shutil

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from news_recsys_amd.dataset.DataReader.columnar import ColumnarDataset, ColumnarLoader, convert_features_txt
from news_recsys_amd.dataset.DataReader.data_reader import DataReader
from news_recsys_amd.dataset.DataReader.pl_dataloader import MINDDataModule
from tests.conftest import CONFIGS, GOLDEN

CFG = os.path.join(CONFIGS, "cf_array_small.yaml")
TXT = os.path.join(GOLDEN, "data", "features_small.txt")


def golden_batches():
    g = dict(np.load(os.path.join(GOLDEN, "datareader.npz")))
    n = int(g.pop("n"))
    nb = 1 + max(int(k.split("/")[0][1:]) for k in g)
    return n, [{k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(f"b{i}/")} for i in range(nb)]


def assert_batch_equal(got, want):
    assert sorted(got) == sorted(want)
    for k, v in want.items():
        t = got[k].cpu().numpy()
        assert t.shape == v.shape, k
        if v.dtype.kind == "f":
            assert t.dtype == v.dtype, k                        # dense float64, masks/labels float32
            assert np.array_equal(t, v), k
        else:
            assert np.array_equal(t.astype(np.int64), v), k      # ids may travel as int32


def test_text_reader_matches_reference():
    n, want = golden_batches()
    ds = DataReader(CFG, TXT)
    assert len(ds) == n == 23
    got = list(DataLoader(ds, batch_size=7, shuffle=False))
    assert len(got) == len(want) == 4
    for gb, wb in zip(got, want):
        assert_batch_equal(gb, wb)
        assert gb["user_id"].dtype == torch.int64 and gb["ctr"].dtype == torch.float64
    item = ds[1]                                                  # over-long history is truncated to 7
    assert item["user_history"].tolist() == [37, 12, 26, 25, 31, 5, 21] and item["user_history_mask"].sum() == 7
    assert "ignored_feature" not in item


def test_text_reader_errors_match_reference(tmp_path):
    with pytest.raises(ValueError, match="Data file path"):
        DataReader(CFG, None)
    with pytest.raises(FileNotFoundError):
        DataReader(CFG, str(tmp_path / "nope.txt"))
    bad = tmp_path / "bad.txt"
    bad.write_text("user_id:1 item_id:2 0\n")                    # no tab
    with pytest.raises(ValueError, match="missing tab"):
        DataReader(CFG, str(bad))[0]
    bad.write_text("user_id:1 item_id\t0\n")                     # no colon
    with pytest.raises(ValueError, match="does not contain"):
        DataReader(CFG, str(bad))[0]


@pytest.fixture(scope="module")
def col_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("col"))
    meta = convert_features_txt(CFG, TXT, d)
    assert meta["n"] == 23 and meta["n_labels"] == 2 and meta["sparse"]["user_id"] == "int32"
    assert meta["array"]["user_history"] == {"dtype": "int32", "max_len": 7}
    return d


def test_columnar_cpu_loader_matches_reference(col_dir):
    n, want = golden_batches()
    ds = ColumnarDataset(col_dir)
    assert len(ds) == n
    loader = ColumnarLoader(ds, 7, "cpu", shuffle=False)
    got = list(loader)
    assert len(got) == len(loader) == 4
    for gb, wb in zip(got, want):
        assert_batch_equal(gb, wb)


def test_oracle_csr_bag_definition_matches_reference_padding(col_dir):
    """oracle.ref_np.csr_bag_to_padded -- the definition of what a CSR bag feature (NRX_FEAT_BAG_CSR) pools -- applied to the
    stored CSR columns reproduces the padded ids + masks of the REFERENCE DataReader (golden batches), truncation included."""
    from oracle import ref_np as R
    n, want = golden_batches()
    ds = ColumnarDataset(col_dir)
    lo = 0
    for wb in want:
        nb = len(wb["label"])
        for k, L in ds.max_len.items():
            off = np.asarray(ds.offsets[k][lo:lo + nb + 1], np.int64)
            vals = np.asarray(ds.values[k][off[0]:off[-1]])
            ids, mask = R.csr_bag_to_padded(vals, off - off[0], L)
            assert np.array_equal(ids.astype(np.int64), wb[k]) and np.array_equal(mask, wb[f"{k}_mask"])
        lo += nb
    assert lo == n
    # longer than L: cut to the first L entries; empty: all padding
    ids, mask = R.csr_bag_to_padded(np.arange(1, 10), np.array([0, 6, 6, 9]), 4)
    assert ids.tolist() == [[1, 2, 3, 4], [0, 0, 0, 0], [7, 8, 9, 0]] and mask.tolist() == [[1, 1, 1, 1], [0, 0, 0, 0], [1, 1, 1, 0]]


def test_columnar_shuffle_is_a_permutation(col_dir):
    ds = ColumnarDataset(col_dir)
    loader = ColumnarLoader(ds, 5, "cpu", shuffle=True, seed=3)
    seen = torch.cat([b["ctr"] for b in loader])
    ref = torch.from_numpy(np.array(ds.dense["ctr"]))
    assert seen.numel() == 23 and torch.equal(seen.sort().values, ref.sort().values)
    again = torch.cat([b["ctr"] for b in loader])                 # next epoch: a different order
    assert not torch.equal(seen, again)
    assert len(list(ColumnarLoader(ds, 5, "cpu", drop_last=True))) == 4


def test_datamodule_mirror(tmp_path):
    import yaml
    cfg = yaml.safe_load(open(CFG))
    cfg["paths"]["out_basedir"] = str(tmp_path)
    cfg["dataset"] = {"batch_size": 7, "num_workers": 0, "pin_memory": False}
    os.makedirs(tmp_path / "extractored_feature")
    for name in ("train_features.txt", "dev_features.txt"):
        shutil.copy(TXT, tmp_path / "extractored_feature" / name)
    cpath = tmp_path / "cfg.yaml"
    cpath.write_text(yaml.safe_dump(cfg))
    dm = MINDDataModule(str(cpath))
    dm.setup("fit")
    _, want = golden_batches()
    for gb, wb in zip(dm.val_dataloader(), want):
        assert_batch_equal(gb, wb)
    for gb, wb in zip(dm.val_loader_columnar("cpu"), want):
        assert_batch_equal(gb, wb)
    assert len(dm.train_dataloader()) == 4
    os.remove(tmp_path / "extractored_feature" / "dev_features.txt")
    with pytest.raises(FileNotFoundError):
        MINDDataModule(str(cpath)).setup("fit")


@pytest.mark.gpu
def test_columnar_device_loader_matches_reference_and_feeds_the_model(col_dir):
    """Device batches (async H2D, CSR expanded on the GPU by nrx_csr_to_padded) == reference batches,
    and they drive the Deep model built from the same YAML."""
    from news_recsys_amd.model.sort.deep.model import Deep
    _, want = golden_batches()
    ds = ColumnarDataset(col_dir)
    got = list(ColumnarLoader(ds, 7, "cuda:0", shuffle=False))
    for gb, wb in zip(got, want):
        assert all(t.is_cuda for t in gb.values())
        assert_batch_equal(gb, wb)
    m = Deep(CFG).to("cuda:0")
    out = m(got[0])
    assert out.shape == (7, 1) and torch.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("shuffle", [False, True])
def test_resident_loader_yields_the_same_batches(col_dir, shuffle):
    """resident=True (dataset kept in HBM, batches gathered on the device) == the streaming loader, bit for
    bit, sequential and shuffled (same permutation, same in-batch order), over two epochs."""
    ds = ColumnarDataset(col_dir)
    a = ColumnarLoader(ds, 5, "cuda:0", shuffle=shuffle, seed=3)
    b = ColumnarLoader(ds, 5, "cuda:0", shuffle=shuffle, seed=3, resident=True)
    for _ in range(2):
        la, lb = list(a), list(b)
        assert len(la) == len(lb) == len(a)
        for x, y in zip(la, lb):
            assert set(x) == set(y)
            for k in x:
                assert x[k].dtype == y[k].dtype and torch.equal(x[k], y[k]), k
    assert b.resident_bytes() > 0
    with pytest.raises(ValueError):
        ColumnarLoader(ds, 5, "cpu", resident=True)


@pytest.mark.gpu
@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("csr", [False, True])
def test_pinned_loader_yields_the_same_batches(col_dir, shuffle, csr):
    """pinned=True (columns page-locked once; a sequential batch = asynchronous copies straight from slices of them, double-buffered on a side
    stream) == the streaming loader over the mmaps, bit for bit, sequential and shuffled, padded and CSR bags, over two epochs -- and the in-memory
    dataset (ColumnarDataset.from_arrays) == the one read from the converted directory."""
    ds = ColumnarDataset(col_dir)
    mem = ColumnarDataset.from_arrays({k: np.array(v) for k, v in ds.sparse.items()}, np.array(ds.label), {k: np.array(v) for k, v in ds.dense.items()},
                                      {k: (np.array(ds.values[k]), np.array(ds.offsets[k]), ds.max_len[k]) for k in ds.values})
    a = ColumnarLoader(ds, 5, "cuda:0", shuffle=shuffle, seed=3, csr_bags=csr)
    b = ColumnarLoader(mem, 5, "cuda:0", shuffle=shuffle, seed=3, csr_bags=csr, pinned=True)
    for _ in range(2):
        la, lb = list(a), list(b)
        torch.cuda.synchronize()
        assert len(la) == len(lb) == len(a)
        for x, y in zip(la, lb):
            assert set(x) == set(y)
            for k in x:
                assert x[k].dtype == y[k].dtype and torch.equal(x[k], y[k]), k
    with pytest.raises(ValueError):
        ColumnarLoader(ds, 5, "cpu", pinned=True)


@pytest.mark.gpu
def test_csr_bag_loader_batches_expand_to_the_reference_batches_and_feed_the_model(col_dir):
    """csr_bags=True: array features travel as stored (ids [nnz] + `name_offsets`); expanded by the oracle's definition of
    DataReader's padding (data_reader.py:96-109) they are the reference's batches, and the model's fused launch on the
    CSR batch gives bit for bit the output it gives on the padded batch."""
    from news_recsys_amd.model.sort.deep.model import Deep
    from oracle import ref_np as R
    _, want = golden_batches()
    ds = ColumnarDataset(col_dir)
    got = list(ColumnarLoader(ds, 7, "cuda:0", shuffle=False, csr_bags=True))
    padded = list(ColumnarLoader(ds, 7, "cuda:0", shuffle=False))
    m = Deep(CFG).to("cuda:0")
    for gb, pb, wb in zip(got, padded, want):
        exp = {k: v for k, v in gb.items() if not k.endswith("_offsets")}
        for k in ds.max_len:
            assert gb[k].dim() == 1 and f"{k}_mask" not in gb
            ids, mask = R.csr_bag_to_padded(gb[k].cpu().numpy(), gb[f"{k}_offsets"].cpu().numpy(), ds.max_len[k])
            exp[k], exp[f"{k}_mask"] = torch.from_numpy(ids), torch.from_numpy(mask)
        assert_batch_equal(exp, wb)
        with torch.no_grad():
            assert torch.equal(m(gb), m(pb))
    with pytest.raises(ValueError):
        ColumnarLoader(ds, 7, "cpu", csr_bags=True)


@pytest.mark.gpu
def test_csr_to_padded_row_selection():
    from news_recsys_amd import ops
    rng = np.random.default_rng(1)
    N, L = 500, 9
    lens = rng.integers(0, 14, N)                       # some longer than L: truncated
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    vals = rng.integers(1, 10 ** 6, off[-1]).astype(np.int32)
    rows = rng.integers(0, N, 77).astype(np.int64)
    ids, mask = ops.csr_to_padded(torch.from_numpy(vals).cuda(), torch.from_numpy(off).cuda(), L, rows=torch.from_numpy(rows).cuda())
    for i, r in enumerate(rows):
        n = min(lens[r], L)
        assert np.array_equal(ids[i, :n].cpu().numpy(), vals[off[r]:off[r] + n]) and not ids[i, n:].any()
        assert mask[i].sum().item() == n


@pytest.mark.gpu
def test_csr_to_padded_kernel_vs_numpy():
    from news_recsys_amd import ops
    rng = np.random.default_rng(0)
    for dt in (np.int32, np.int64):
        B, L = 1000, 50
        lens = rng.integers(0, L + 1, B)
        lens[:3] = [0, L, 1]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        vals = rng.integers(1, 10 ** 6, off[-1]).astype(dt)
        ids, mask = ops.csr_to_padded(torch.from_numpy(vals).cuda(), torch.from_numpy(off).cuda(), L)
        ref = np.zeros((B, L), dt)
        rm = np.zeros((B, L), np.float32)
        for b in range(B):
            ref[b, :lens[b]] = vals[off[b]:off[b + 1]]
            rm[b, :lens[b]] = 1
        assert np.array_equal(ids.cpu().numpy(), ref) and np.array_equal(mask.cpu().numpy(), rm)
