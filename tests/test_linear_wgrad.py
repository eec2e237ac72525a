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
