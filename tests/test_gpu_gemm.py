"""GPU parity of the MFMA GEMM (linear fwd/dgrad/wgrad, conv1d implicit GEMM) against fp64 CPU math."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


TOL = {3: 3e-5, 1: 2e-2}


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("M,N,K", [(200, 256, 256), (333, 81, 256), (128, 1024, 256), (77, 256, 1024), (513, 46, 256),
                                   (40, 256, 80), (3000, 1024, 256), (2500, 1100, 64)])
def test_linear_fwd_dgrad_wgrad(nsplit, M, N, K):
    from unast_amd import ops, config
    config.NSPLIT = nsplit
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    ldn = (N + 3) // 4 * 4
    y = torch.zeros(M, ldn, device=dev())
    ops.linear_fwd(x.to(dev()), W.to(dev()), b.to(dev()), y[:, :N])
    assert relerr(y[:, :N], x.double() @ W.double().t() + b.double()) < TOL[nsplit]
    assert (y[:, N:] == 0).all()
    # relu epilogue + residual
    R = torch.randn(M, N, generator=g)
    y2 = torch.zeros(M, ldn, device=dev())
    Rd = torch.zeros(M, ldn, device=dev()); Rd[:, :N] = R.to(dev())
    ops.linear_fwd(x.to(dev()), W.to(dev()), b.to(dev()), y2[:, :N], act=1, R=Rd[:, :N])
    assert relerr(y2[:, :N], torch.relu(x.double() @ W.double().t() + b.double()) + R.double()) < TOL[nsplit]
    # dgrad (with gate)
    dy = torch.zeros(M, ldn); dy[:, :N] = torch.randn(M, N, generator=g)
    gate = torch.randn(M, K, generator=g)
    dx = torch.empty(M, K, device=dev())
    ops.linear_dgrad(dy.to(dev())[:, :N], W.to(dev()), dx, G=gate.to(dev()), gate_scale=2.0)
    ref = (dy[:, :N].double() @ W.double()) * (gate > 0).double() * 2.0
    assert relerr(dx, ref) < TOL[nsplit]
    # wgrad accumulates
    dW = torch.ones(N, K, device=dev())
    ops.linear_wgrad(dy.to(dev())[:, :N], x.to(dev()), dW)
    assert relerr(dW, 1.0 + dy[:, :N].double().t() @ x.double()) < TOL[nsplit]
    config.NSPLIT = 3


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("B,T,Cin,Cout,pad", [(3, 37, 256, 256, 2), (2, 50, 80, 256, 4), (2, 64, 256, 80, 4), (1, 5, 256, 256, 2)])
def test_conv1d_k5(nsplit, B, T, Cin, Cout, pad):
    from unast_amd import ops, config
    config.NSPLIT = nsplit
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    W = torch.randn(Cout, Cin, 5, generator=g, dtype=torch.float64) * 0.05      # torch layout
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True); Wr = W.clone().requires_grad_(True)
    xp = torch.nn.functional.pad(xr.transpose(1, 2), (pad, 4 - pad))
    yr = torch.nn.functional.conv1d(xp, Wr, b).transpose(1, 2)
    dy = torch.randn(B, T, Cout, generator=g, dtype=torch.float64)
    yr.backward(dy)
    Wp = W.permute(0, 2, 1).contiguous().float().to(dev())                       # [Cout,5,Cin]
    y = torch.empty(B, T, Cout, device=dev())
    ops.conv_fwd(x.float().to(dev()), Wp, b.float().to(dev()), y, pad)
    assert relerr(y, yr.detach()) < TOL[nsplit]
    dx = torch.empty(B, T, Cin, device=dev())
    ops.conv_dgrad(dy.float().to(dev()), Wp, dx, pad)
    assert relerr(dx, xr.grad) < TOL[nsplit]
    dWp = torch.zeros(Cout, 5, Cin, device=dev())
    ops.conv_wgrad(dy.float().to(dev()), x.float().to(dev()), dWp, pad)
    assert relerr(dWp, Wr.grad.permute(0, 2, 1)) < TOL[nsplit]
    config.NSPLIT = 3


def test_gemm_dropout_epilogue_statistics():
    from unast_amd import ops
    M, N, K = 512, 256, 64
    x = torch.ones(M, K, device=dev()); W = torch.ones(N, K, device=dev()) / K
    y = torch.empty(M, N, device=dev())
    ops.linear_fwd(x, W, None, y, drop_p=0.25, seed=123, stream_id=5)
    keep = (y > 0).float().mean().item()
    assert abs(keep - 0.75) < 0.01
    assert torch.allclose(y[y > 0], torch.tensor(1 / 0.75, device=dev()), rtol=1e-4)
    y2 = torch.empty(M, N, device=dev())
    ops.linear_fwd(x, W, None, y2, drop_p=0.25, seed=123, stream_id=5)
    assert torch.equal(y, y2)                      # deterministic in (seed, stream)
    ops.linear_fwd(x, W, None, y2, drop_p=0.25, seed=124, stream_id=5)
    assert not torch.equal(y, y2)


def test_gemm_rejects_bad_arguments():
    from unast_amd import ops
    from unast_amd._lib import UnastHipError
    x = torch.ones(8, 6, device=dev()); W = torch.ones(4, 6, device=dev()); y = torch.empty(8, 4, device=dev())
    with pytest.raises(UnastHipError):
        ops.linear_fwd(x, W, None, y)              # lda = 6 is not a multiple of 4
