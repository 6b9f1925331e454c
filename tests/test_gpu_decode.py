"""GPU parity of the single-position decoding kernels (csrc/decode.hip) against fp64 CPU math and against the general kernels
they stand in for (same dropout streams)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def relerr(a, b):
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(32, 768, 256), (3, 81, 256), (33, 46, 1024), (32, 256, 80), (1, 1024, 256), (70, 256, 1024), (32, 16, 4)])
def test_decode_linear_matches_fp64(M, N, K):
    from unast_amd import ops
    g = torch.Generator().manual_seed(M * 13 + N + K)
    x, W, b, R = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ldn = (N + 3) // 4 * 4
    y = torch.zeros(M, ldn, device=D)
    ops.decode_linear(x.to(D), W.to(D), b.to(D), y)
    assert relerr(y[:, :N], x.double() @ W.double().t() + b.double()) < 3e-5
    assert (y[:, N:] == 0).all(), "columns past N must not be written"
    y2 = torch.zeros(M, ldn, device=D)
    Rd = torch.zeros(M, ldn, device=D)
    Rd[:, :N] = R.to(D)
    ops.decode_linear(x.to(D), W.to(D), b.to(D), y2, act=1, R=Rd)
    assert relerr(y2[:, :N], torch.relu(x.double() @ W.double().t() + b.double()) + R.double()) < 3e-5
    y3 = torch.zeros(M, ldn, device=D)
    ops.decode_linear(x.to(D), W.to(D), None, y3)
    assert relerr(y3[:, :N], x.double() @ W.double().t()) < 3e-5


@pytest.mark.parametrize("M,N,K", [(32, 1024, 256), (5, 256, 256), (40, 768, 128)])
def test_decode_linear_fused_layernorm(M, N, K):
    from unast_amd import ops
    g = torch.Generator().manual_seed(M + N)
    z = torch.randn(M, K, generator=g) * 3 + 1
    gam, bet = torch.rand(K, generator=g) + .5, torch.randn(K, generator=g)
    W, b = torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    xn_ref = torch.nn.functional.layer_norm(z.double(), (K,), gam.double(), bet.double(), 1e-5)
    y = torch.zeros(M, N, device=D)
    xn = torch.zeros(M, K, device=D)
    ops.decode_linear(z.to(D), W.to(D), b.to(D), y, act=1, ln=(gam.to(D), bet.to(D)), xn_out=xn)
    assert relerr(xn, xn_ref) < 1e-5
    assert relerr(y, torch.relu(xn_ref @ W.double().t() + b.double())) < 3e-5


def test_decode_linear_appends_to_cache_at_device_position():
    from unast_amd import ops
    B, E, Tcap = 6, 256, 9
    g = torch.Generator().manual_seed(3)
    x, W, b = torch.randn(B, E, generator=g), torch.randn(3 * E, E, generator=g) * 0.1, torch.randn(3 * E, generator=g)
    ref = x.double() @ W.double().t() + b.double()
    cache = torch.zeros(B, Tcap, 2 * E, device=D)
    q = torch.zeros(B, E, device=D)
    pos = torch.tensor([4], dtype=torch.int64, device=D)
    ops.decode_linear(x.to(D), W.to(D), b.to(D), q, cache=cache, split_col=E, pos=pos)
    assert relerr(q, ref[:, :E]) < 3e-5
    assert relerr(cache[:, 4], ref[:, E:]) < 3e-5
    cache[:, 4] = 0
    assert (cache == 0).all(), "only row `pos` of every sequence may be written"


def test_decode_linear_dropout_uses_the_gemm_streams():
    """Same (seed, stream) => the mask of the general GEMM epilogue."""
    from unast_amd import ops
    M, N, K = 32, 256, 256
    g = torch.Generator().manual_seed(5)
    x, W, b = torch.randn(M, K, generator=g).to(D), (torch.randn(N, K, generator=g) * 0.1).to(D), torch.randn(N, generator=g).to(D)
    y1, y2 = torch.zeros(M, N, device=D), torch.zeros(M, N, device=D)
    ops.decode_linear(x, W, b, y1, drop_p=0.3, seed=11, stream_id=5)
    ops.linear_fwd(x, W, b, y2, drop_p=0.3, seed=11, stream_id=5)
    assert torch.equal(y1 == 0, y2 == 0)
    assert 0.2 < float((y1 == 0).float().mean()) < 0.4
    assert relerr(y1, y2.cpu()) < 3e-5


@pytest.mark.parametrize("B,H,Tcap,drop", [(32, 4, 800, 0.0), (3, 4, 37, 0.0), (5, 2, 300, 0.25)])
def test_decode_attn(B, H, Tcap, drop):
    from unast_amd import ops
    E = 64 * H
    g = torch.Generator().manual_seed(B + Tcap)
    q = torch.randn(B, E, generator=g)
    kv = torch.randn(B, Tcap, 2 * E, generator=g)
    lens = torch.randint(1, Tcap + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = Tcap
    lens[-1] = 1
    kvd = kv.to(D).view(B * Tcap, 2 * E)
    O = torch.zeros(B, E, device=D)
    ops.decode_attn(q.to(D), kvd[:, :E], kvd[:, E:], Tcap, lens.to(D), O, H, drop_p=drop, seed=9, stream_id=2)
    # the general attention kernel with one query per sequence draws the same mask
    O2 = torch.zeros(B, E, device=D)
    lse = torch.zeros(B, H, 1, device=D)
    ops.attn_fwd(q.to(D), kvd[:, :E], kvd[:, E:], O2, lse, lens.to(D), B, H, 1, Tcap, False, drop_p=drop, seed=9, stream_id=2)
    assert relerr(O, O2.cpu()) < 2e-5
    if drop == 0.0:
        ref = torch.zeros(B, E, dtype=torch.float64)
        for b in range(B):
            n = int(lens[b])
            for h in range(H):
                qq = q[b, 64 * h:64 * h + 64].double()
                kk = kv[b, :n, 64 * h:64 * h + 64].double()
                vv = kv[b, :n, E + 64 * h:E + 64 * h + 64].double()
                ref[b, 64 * h:64 * h + 64] = torch.softmax(kk @ qq * 0.125, 0) @ vv
        assert relerr(O, ref) < 1e-5
