"""Opt-in MLP weight gradient (nrx_linear_wgrad, ops.linear): gradients against torch's own nn.Linear in fp64, and the
drop-in property of the layer class (same parameters and state_dict keys as the reference's MLP, src/model/model_utils/utils.py:6-17)."""
import pytest
import torch

from news_recsys_amd.model.model_utils import utils as mlp_utils


def test_mlp_state_dict_keys_same_with_and_without_switch(monkeypatch):
    plain = mlp_utils.MLP([12, 8, 4, 1])
    monkeypatch.setattr(mlp_utils, "MLP_WGRAD", True)
    fast = mlp_utils.MLP([12, 8, 4, 1])
    assert list(plain.state_dict().keys()) == list(fast.state_dict().keys())
    assert [tuple(v.shape) for v in plain.state_dict().values()] == [tuple(v.shape) for v in fast.state_dict().values()]
    fast.load_state_dict(plain.state_dict())
    x = torch.randn(5, 12)
    assert torch.equal(plain(x), fast(x))            # CPU inputs take nn.Linear's own path


@pytest.mark.gpu
@pytest.mark.parametrize("batch,in_f,out_f", [(1, 4, 4), (257, 416, 128), (4096, 128, 64), (1000, 64, 1), (777, 37, 19), (65536, 128, 128),
                                               (3, 1, 1), (5000, 1000, 130)])
def test_linear_grads_match_fp64(batch, in_f, out_f):
    from news_recsys_amd import ops
    g = torch.Generator(device="cuda").manual_seed(batch + in_f)
    a = torch.randn(batch, in_f, device="cuda", generator=g, requires_grad=True)
    W = (torch.randn(out_f, in_f, device="cuda", generator=g) / in_f ** 0.5).requires_grad_()
    b = torch.randn(out_f, device="cuda", generator=g, requires_grad=True)
    up = torch.randn(batch, out_f, device="cuda", generator=g)
    y = ops.linear(a, W, b)
    y.backward(up)
    a64, W64, b64 = (t.detach().double().requires_grad_() for t in (a, W, b))
    y64 = torch.nn.functional.linear(a64, W64, b64)
    y64.backward(up.double())
    assert torch.equal(y, torch.nn.functional.linear(a, W, b))
    # fp32 sums over the batch: tolerance scaled by sqrt(batch) * eps * magnitude
    tol = 4e-6 * max(1.0, batch ** 0.5)
    assert (W.grad.double() - W64.grad).abs().max().item() <= tol * max(1.0, W64.grad.abs().max().item())
    assert (b.grad.double() - b64.grad).abs().max().item() <= tol * max(1.0, b64.grad.abs().max().item())
    assert torch.allclose(a.grad.double(), a64.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_linear_strided_input_and_no_bias():
    """The input is a column slice of a wider buffer (the concat output's deep columns) and the layer has no bias."""
    from news_recsys_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    buf = torch.randn(2048, 96, device="cuda", generator=g)
    a = buf[:, 8:72].requires_grad_()
    W = torch.randn(32, 64, device="cuda", generator=g).requires_grad_()
    up = torch.randn(2048, 32, device="cuda", generator=g)
    ops.linear(a, W, None).backward(up)
    ref = up.double().t() @ buf[:, 8:72].double()
    assert (W.grad.double() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


@pytest.mark.gpu
def test_mlp_training_step_matches_plain_layers(monkeypatch):
    plain = mlp_utils.MLP([416, 128, 128, 64, 1]).cuda()
    monkeypatch.setattr(mlp_utils, "MLP_WGRAD", True)
    fast = mlp_utils.MLP([416, 128, 128, 64, 1]).cuda()
    fast.load_state_dict(plain.state_dict())
    x = torch.randn(8192, 416, device="cuda")
    lp = torch.sigmoid(plain(x)).mean()
    lf = torch.sigmoid(fast(x)).mean()
    assert torch.equal(lp, lf)
    lp.backward()
    lf.backward()
    for (n, p), (_, q) in zip(plain.named_parameters(), fast.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=2e-4, atol=1e-7), n


@pytest.mark.gpu
@pytest.mark.parametrize("batch,in_f,out_f", [(1, 4, 4), (257, 416, 128), (4096, 128, 64), (1000, 64, 1), (777, 37, 19), (65536, 128, 128),
                                               (20000, 112, 112), (70001, 320, 320), (5000, 1000, 130), (300000, 16, 16)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_ordered_weight_gradient_is_bit_reproducible_and_matches_fp64(batch, in_f, out_f, bias, monkeypatch):
    """ops.WGRAD_ORDERED (nrx_linear_wgrad_ordered): every batch slice's partial tile is STORED and the slices are added in slice order by a second
    launch -- the same bits on every run (three runs compared word for word), where the default mode adds them with float atomics; against fp64
    like the default mode.  Shapes: one slice, many slices, a last slice shorter than a slab, out_features = 1, unaligned widths (the scalar
    load path), tiles past the matrix edge."""
    from news_recsys_amd import ops
    monkeypatch.setattr(ops, "WGRAD_ORDERED", True)
    g = torch.Generator(device="cuda").manual_seed(batch + in_f + out_f)
    a = torch.randn(batch, in_f, device="cuda", generator=g)
    W = (torch.randn(out_f, in_f, device="cuda", generator=g) / in_f ** 0.5).requires_grad_()
    b = torch.randn(out_f, device="cuda", generator=g, requires_grad=True) if bias else None
    up = torch.randn(batch, out_f, device="cuda", generator=g)
    runs = []
    for _ in range(3):
        W.grad = None
        if b is not None:
            b.grad = None
        ops.linear(a, W, b).backward(up)
        torch.cuda.synchronize()
        runs.append((W.grad.clone(), b.grad.clone() if b is not None else None))
    for gw, gb in runs[1:]:
        assert torch.equal(gw.view(torch.int32), runs[0][0].view(torch.int32))
        if b is not None:
            assert torch.equal(gb.view(torch.int32), runs[0][1].view(torch.int32))
    ref_w = up.double().t() @ a.double()
    tol = 4e-6 * max(1.0, batch ** 0.5)
    assert (runs[0][0].double() - ref_w).abs().max().item() <= tol * max(1.0, ref_w.abs().max().item())
    if b is not None:
        ref_b = up.double().sum(0)
        assert (runs[0][1].double() - ref_b).abs().max().item() <= tol * max(1.0, ref_b.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,NL", [(130, 112, 3), (20011, 64, 2), (9000, 320, 2), (65, 37, 2), (30000, 16, 1), (4133, 128, 2)])
@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
def test_dcn_v2_ordered_weight_gradients_are_bit_reproducible(B, D, NL, math, monkeypatch):
    """The DCN-v2 stack's backward with flags bit 2 (ops.WGRAD_ORDERED): g_W and g_b of every layer word for word the same over three runs -- the panel
    form (dim <= 112) and the three-launch form, aligned and unaligned widths -- and g_x unchanged by the switch (it never depended on atomics);
    against the atomic mode within the order of its sums."""
    from news_recsys_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(B + D + NL)
    x = torch.randn(B, D, device="cuda", generator=gen).requires_grad_()
    Ws = [(torch.randn(D, D, device="cuda", generator=gen) / D ** 0.5).requires_grad_() for _ in range(NL)]
    bs = [(torch.randn(D, device="cuda", generator=gen) * 0.1).requires_grad_() for _ in range(NL)]
    up = torch.randn(B, D, device="cuda", generator=gen)

    def grads(ordered):
        monkeypatch.setattr(ops, "WGRAD_ORDERED", ordered)
        out = ops.dcn_v2(x, Ws, bs, math=math)
        gs = torch.autograd.grad(out, [x] + Ws + bs, up)
        torch.cuda.synchronize()
        return [t.clone() for t in gs]
    ref = grads(False)
    runs = [grads(True) for _ in range(3)]
    for r in runs[1:]:
        for t0, t1 in zip(runs[0], r):
            assert torch.equal(t0.view(torch.int32), t1.view(torch.int32))
    assert torch.equal(runs[0][0], ref[0])                       # g_x: no batch-wide sum in it
    for t0, t1 in zip(runs[0][1:], ref[1:]):
        tol = (3e-3 if math == "bf16x3" else 1e-4) * max(1.0, B ** 0.5) * max(1.0, t1.abs().max().item()) * 1e-1
        assert (t0 - t1).abs().max().item() <= tol
