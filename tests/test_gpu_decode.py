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
    ops.decode_attn(q.to(D), kvd[:, :E], kvd[:, E:], Tcap, O, H, lens=lens.to(D), drop_p=drop, seed=9, stream_id=2)
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
    # valid length from the device-resident position and stop lengths: min(stop + 1, pos + 1)
    pos = torch.tensor([min(20, Tcap - 1)], dtype=torch.int64, device=D)
    stop = torch.full((B,), Tcap, dtype=torch.int64)
    stop[-1] = 3
    O3, O4 = torch.zeros(B, E, device=D), torch.zeros(B, E, device=D)
    ops.decode_attn(q.to(D), kvd[:, :E], kvd[:, E:], Tcap, O3, H, stop_lens=stop.to(D), pos=pos)
    ops.decode_attn(q.to(D), kvd[:, :E], kvd[:, E:], Tcap, O4, H, lens=torch.minimum(stop + 1, pos.cpu() + 1).to(torch.int32).to(D))
    assert torch.equal(O3, O4)


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_decode_linear_row_producers_match_the_standalone_kernels(drop):
    """Embedding + positional encoding, positional encoding alone and LayerNorm + dropout produced inside the contraction give
    what embed_fwd / posenc_fwd / layernorm_fwd / leaky_dropout followed by the plain contraction give (same dropout streams)."""
    import math
    from unast_amd import ops
    B, E, N, T, V = 7, 256, 96, 12, 46
    g = torch.Generator().manual_seed(21)
    tokens = torch.randint(0, V, (B, T), generator=g).to(D)
    emb, pe = torch.randn(V, E, generator=g).to(D), torch.randn(40, E, generator=g).to(D)
    W, b = (torch.randn(N, E, generator=g) * 0.1).to(D), torch.randn(N, generator=g).to(D)
    pos = torch.tensor([5], dtype=torch.int64, device=D)
    sc = math.sqrt(E)
    # embedding + positional encoding
    y, xn = torch.zeros(B, N, device=D), torch.zeros(B, E, device=D)
    ops.decode_linear(None, W, b, y, embed=(tokens, emb, pe, sc, (drop, 3), (drop, 4)), xn_out=xn, seed=17, pos=pos)
    x0, x1, y_ref = torch.zeros(B, E, device=D), torch.zeros(B, E, device=D), torch.zeros(B, N, device=D)
    ops.embed_fwd(tokens[:, 5].contiguous(), emb, x0, 1, drop_p=drop, seed=17, stream_id=3)
    ops.posenc_fwd(x0, pe[5:6], x1, 1, sc, drop_p=drop, seed=17, stream_id=4)
    ops.decode_linear(x1, W, b, y_ref)
    assert relerr(xn, x1.cpu()) < 1e-6 and relerr(y, y_ref.cpu()) < 1e-5
    # positional encoding of given rows, rows taken from a [B, T, K] buffer at the position
    frames = torch.randn(B, T, E, generator=g).to(D)
    y2, xn2 = torch.zeros(B, N, device=D), torch.zeros(B, E, device=D)
    ops.decode_linear(frames[:, 5].contiguous(), W, b, y2, posenc=(pe, sc, (drop, 6)), xn_out=xn2, seed=17, pos=pos)
    ops.posenc_fwd(frames[:, 5].contiguous(), pe[5:6], x1, 1, sc, drop_p=drop, seed=17, stream_id=6)
    ops.decode_linear(x1, W, b, y_ref)
    assert relerr(xn2, x1.cpu()) < 1e-6 and relerr(y2, y_ref.cpu()) < 1e-5
    y3 = torch.zeros(B, N, device=D)
    ops.decode_linear(None, W, b, y3, x_frames=frames, pos=pos)
    ops.decode_linear(frames[:, 5].contiguous(), W, b, y_ref)
    assert torch.equal(y3, y_ref)
    # LayerNorm + dropout
    z = (torch.randn(B, E, generator=g) * 2 + .5).to(D)
    gam, bet = (torch.rand(E, generator=g) + .5).to(D), torch.randn(E, generator=g).to(D)
    y4 = torch.zeros(B, N, device=D)
    ops.decode_linear(z, W, b, y4, ln=(gam, bet), ln_drop=(drop, 8), seed=17)
    xl, mean, rstd, xd = torch.zeros(B, E, device=D), torch.zeros(B, device=D), torch.zeros(B, device=D), torch.zeros(B, E, device=D)
    ops.layernorm_fwd(z, gam, bet, xl, mean, rstd)
    ops.leaky_dropout(xl, None, xd, 1.0, drop_p=drop, seed=17, stream_id=8)
    ops.decode_linear(xd, W, b, y_ref)
    assert relerr(y4, y_ref.cpu()) < 1e-5
    if drop > 0:
        assert torch.equal(y4 == 0, y_ref == 0) or relerr(y4, y_ref.cpu()) < 1e-5
