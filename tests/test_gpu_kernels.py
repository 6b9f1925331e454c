"""GPU parity of each HIP kernel against fp64 CPU math / torch CPU autograd of the same op."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def relerr(a, b):
    b = b.double()
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def naive_attention(q, k, v, lens_k, causal, H):
    B, Tq, E = q.shape
    Tk = k.shape[1]
    hd = E // H
    qh = q.view(B, Tq, H, hd).transpose(1, 2) / math.sqrt(hd)
    kh = k.view(B, Tk, H, hd).transpose(1, 2)
    vh = v.view(B, Tk, H, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    neg = ~(torch.arange(Tk)[None, :] < lens_k[:, None])[:, None, None, :]
    if causal:
        neg = neg | (torch.arange(Tk)[None, :] > torch.arange(Tq)[:, None])[None, None]
    s = s.masked_fill(neg, float("-inf"))
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, E), torch.logsumexp(s, -1)


@pytest.fixture(params=[1, 0], ids=["mfma32x32", "mfma16x16"])
def fwd_variant(request):
    """Both forward kernels (csrc/attention.hip: attn_fwd32_kernel, the default, and attn_q_kernel) serve the same call."""
    from unast_amd._lib import lib
    old = lib().unast_attn_fwd_variant(request.param)
    yield request.param
    lib().unast_attn_fwd_variant(old)


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("B,Tq,Tk,causal", [(2, 40, 40, False), (3, 200, 200, True), (2, 150, 37, False), (2, 33, 190, False),
                                            (1, 257, 257, True), (2, 128, 64, False)])
def test_attention_fwd_bwd(nsplit, B, Tq, Tk, causal, fwd_variant):
    from unast_amd import ops
    H, E = 4, 256
    g = torch.Generator().manual_seed(B * 1000 + Tq + Tk)
    qkv = torch.randn(B, Tq, 3 * E, generator=g, dtype=torch.float64)
    kvsrc = qkv if Tq == Tk else torch.randn(B, Tk, 3 * E, generator=g, dtype=torch.float64)
    lens = torch.randint(max(1, Tk // 2), Tk + 1, (B,), generator=g)
    lens[0] = Tk
    q = qkv[..., :E].clone().requires_grad_(True)
    k = kvsrc[..., E:2 * E].clone().requires_grad_(True)
    v = kvsrc[..., 2 * E:].clone().requires_grad_(True)
    o_ref, lse_ref = naive_attention(q, k, v, lens, causal, H)
    do = torch.randn(B, Tq, E, generator=g, dtype=torch.float64)
    o_ref.backward(do)

    qd = qkv.float().to(D).view(B * Tq, 3 * E)
    kd = kvsrc.float().to(D).view(B * Tk, 3 * E)
    O = torch.empty(B * Tq, E, device=D)
    LSE = torch.empty(B, H, Tq, device=D)
    lens_d = lens.to(torch.int32).to(D)
    ops.attn_fwd(qd[:, :E], kd[:, E:2 * E], kd[:, 2 * E:], O, LSE, lens_d, B, H, Tq, Tk, causal, nsplit=nsplit)
    tol = 5e-5 if nsplit == 3 else 2e-2
    assert relerr(O.view(B, Tq, E), o_ref.detach()) < tol
    assert relerr(LSE, lse_ref.detach()) < tol
    dQ = torch.full((B * Tq, E), float("nan"), device=D)
    dKV = torch.full((B * Tk, 2 * E), float("nan"), device=D)
    ws = torch.empty(B, H, Tq, device=D)
    ops.attn_bwd(qd[:, :E], kd[:, E:2 * E], kd[:, 2 * E:], O, do.float().to(D).view(B * Tq, E), LSE, ws, dQ, dKV[:, :E], dKV[:, E:],
                 lens_d, B, H, Tq, Tk, causal, nsplit=nsplit)
    assert relerr(dQ.view(B, Tq, E), q.grad) < tol
    assert relerr(dKV[:, :E].reshape(B, Tk, E), k.grad) < tol
    assert relerr(dKV[:, E:].reshape(B, Tk, E), v.grad) < tol


@pytest.mark.parametrize("B,Tq,Tk,causal,p", [(2, 200, 200, True, 0.0), (3, 150, 37, False, 0.1), (2, 300, 300, False, 0.1), (1, 257, 257, True, 0.1)])
def test_attention_forward_kernels_agree(B, Tq, Tk, causal, p):
    """attn_fwd32_kernel (32x32x16 MFMAs, running maximum raised only when it grows by more than 8) against attn_q_kernel (16x16x32)
    on the same call: same masks and, with dropout on, the SAME keep decisions (an element dropped by one and kept by the other would
    show as a difference of its whole probability), results equal to accumulation order."""
    from unast_amd import ops
    from unast_amd._lib import lib
    H, E = 4, 256
    g = torch.Generator().manual_seed(B * 31 + Tq + Tk)
    q = (torch.randn(B * Tq, E, generator=g) * 2).to(D); k = (torch.randn(B * Tk, E, generator=g) * 2).to(D); v = torch.randn(B * Tk, E, generator=g).to(D)
    lens = torch.randint(max(1, Tk // 2), Tk + 1, (B,), generator=g).to(torch.int32).to(D)
    res = []
    old = lib().unast_attn_fwd_variant(-1)
    try:
        for m32 in (0, 1):
            lib().unast_attn_fwd_variant(m32)
            o = torch.full((B * Tq, E), float("nan"), device=D); lse = torch.full((B, H, Tq), float("nan"), device=D)
            ops.attn_fwd(q, k, v, o, lse, lens, B, H, Tq, Tk, causal, drop_p=p, seed=5, stream_id=9)
            res.append((o, lse))
    finally:
        lib().unast_attn_fwd_variant(old)
    assert relerr(res[1][0], res[0][0].cpu()) < 2e-5 and relerr(res[1][1], res[0][1].cpu()) < 2e-6


@pytest.mark.parametrize("B,Tq,Tk,causal", [(2, 40, 40, False), (3, 200, 200, True), (2, 150, 37, False)])
def test_attention_backward_two_term_mode_stays_within_its_bound(B, Tq, Tk, causal, monkeypatch):
    """config.ATTN_BWD_TERMS = 2 (opt-in): P and dS as one bf16 part in the dV / dK / dQ products.  One call's gradients are then
    accurate to the bf16 rounding of P / dS -- bounded here at 4e-3 of the largest element and 3e-3 in norm -- where the default
    three-term form holds 5e-5 (test_attention_fwd_bwd)."""
    from unast_amd import ops, config
    monkeypatch.setattr(config, "ATTN_BWD_TERMS", 2)
    H, E = 4, 256
    g = torch.Generator().manual_seed(B * 1000 + Tq + Tk)
    qkv = torch.randn(B, Tq, 3 * E, generator=g, dtype=torch.float64)
    kvsrc = qkv if Tq == Tk else torch.randn(B, Tk, 3 * E, generator=g, dtype=torch.float64)
    lens = torch.randint(max(1, Tk // 2), Tk + 1, (B,), generator=g)
    lens[0] = Tk
    q = qkv[..., :E].clone().requires_grad_(True); k = kvsrc[..., E:2 * E].clone().requires_grad_(True); v = kvsrc[..., 2 * E:].clone().requires_grad_(True)
    o_ref, _ = naive_attention(q, k, v, lens, causal, H)
    do = torch.randn(B, Tq, E, generator=g, dtype=torch.float64)
    o_ref.backward(do)
    qd = qkv.float().to(D).view(B * Tq, 3 * E); kd = kvsrc.float().to(D).view(B * Tk, 3 * E)
    O = torch.empty(B * Tq, E, device=D); LSE = torch.empty(B, H, Tq, device=D); lens_d = lens.to(torch.int32).to(D)
    ops.attn_fwd(qd[:, :E], kd[:, E:2 * E], kd[:, 2 * E:], O, LSE, lens_d, B, H, Tq, Tk, causal)
    dQ = torch.empty(B * Tq, E, device=D); dKV = torch.empty(B * Tk, 2 * E, device=D); ws = torch.empty(B, H, Tq, device=D)
    ops.attn_bwd(qd[:, :E], kd[:, E:2 * E], kd[:, 2 * E:], O, do.float().to(D).view(B * Tq, E), LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens_d, B, H, Tq, Tk, causal)
    for got, ref in ((dQ.view(B, Tq, E), q.grad), (dKV[:, :E].reshape(B, Tk, E), k.grad), (dKV[:, E:].reshape(B, Tk, E), v.grad)):
        assert relerr(got, ref) < 4e-3
        assert ((got.double().cpu() - ref).norm() / ref.norm()).item() < 3e-3


@pytest.mark.parametrize("B,Tq,Tk,causal,p", [(2, 200, 200, True, 0.0), (3, 150, 37, False, 0.1), (2, 33, 300, False, 0.1), (2, 257, 257, True, 0.1), (1, 5, 5, False, 0.0)])
def test_attention_backward_one_pass_equals_two_kernels(B, Tq, Tk, causal, p):
    """The fused backward (dK, dV and dQ from one sweep; dQ summed over key blocks with fp32 atomics) against the dQ-kernel +
    dK/dV-kernel pair on the same inputs and the same dropout stream: dK / dV bit-identical (same arithmetic), dQ to summation
    order; ragged key lengths, rows beyond Tq untouched-then-zeroed, strided dQ view (the self-attention layout)."""
    from unast_amd import ops, config
    H, E = 4, 256
    g = torch.Generator().manual_seed(B * 1000 + Tq + Tk)
    q = torch.randn(B * Tq, E, generator=g).to(D); k = torch.randn(B * Tk, E, generator=g).to(D); v = torch.randn(B * Tk, E, generator=g).to(D)
    lens = torch.randint(max(1, Tk // 2), Tk + 1, (B,), generator=g).to(torch.int32).to(D)
    o = torch.empty(B * Tq, E, device=D); lse = torch.empty(B, H, Tq, device=D)
    ops.attn_fwd(q, k, v, o, lse, lens, B, H, Tq, Tk, causal, drop_p=p, seed=7, stream_id=3)
    dO = torch.randn(B * Tq, E, generator=g).to(D)
    res = []
    for fused in (False, True):
        config.ATTN_FUSED_BWD = fused
        ws = torch.empty(B, H, Tq, device=D)
        dqkv = torch.full((B * Tq, 3 * E), float("nan"), device=D)          # dQ is a column slice, as in self-attention
        dk = torch.empty(B * Tk, E, device=D); dv = torch.empty(B * Tk, E, device=D)
        ops.attn_bwd(q, k, v, o, dO, lse, ws, dqkv[:, :E], dk, dv, lens, B, H, Tq, Tk, causal, drop_p=p, seed=7, stream_id=3)
        assert torch.isnan(dqkv[:, E:]).all()                                # nothing written outside the dQ columns
        res.append((dqkv[:, :E].clone(), dk, dv))
    config.ATTN_FUSED_BWD = True
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert relerr(res[1][0], res[0][0].cpu()) < 2e-5


@pytest.mark.parametrize("B,Tq,Tk,causal,p", [(2, 200, 200, True, 0.1), (2, 150, 37, False, 0.0), (1, 33, 300, False, 0.1)])
def test_attention_on_presplit_operands(B, Tq, Tk, causal, p):
    """Q / K / V / dO handed over in the pre-split format (what the projections' out_split epilogue stores; produced here by
    unast_split_f32) give the same O, LSE, dQ, dK, dV as the fp32 operands: the same hi/lo values enter the MFMAs, only the
    place of the score scale differs (in the exponent's fma instead of in Q)."""
    from unast_amd import ops
    H, E = 4, 256
    g = torch.Generator().manual_seed(B * 77 + Tq)
    q = torch.randn(B * Tq, E, generator=g).to(D); k = torch.randn(B * Tk, E, generator=g).to(D); v = torch.randn(B * Tk, E, generator=g).to(D)
    dO = torch.randn(B * Tq, E, generator=g).to(D)
    lens = torch.randint(max(1, Tk // 2), Tk + 1, (B,), generator=g).to(torch.int32).to(D)
    sp = [torch.empty_like(t) for t in (q, k, v, dO)]
    for src, dst in zip((q, k, v, dO), sp):
        ops.split_f32(src.view(-1), dst.view(-1))
    res = []
    for split in (False, True):
        Q_, K_, V_, dO_ = sp if split else (q, k, v, dO)
        o = torch.empty(B * Tq, E, device=D); lse = torch.empty(B, H, Tq, device=D)
        ops.attn_fwd(Q_, K_, V_, o, lse, lens, B, H, Tq, Tk, causal, drop_p=p, seed=11, stream_id=2, qkv_split=split)
        ws = torch.empty(B, H, Tq, device=D)
        dq = torch.empty(B * Tq, E, device=D); dk = torch.empty(B * Tk, E, device=D); dv = torch.empty(B * Tk, E, device=D)
        ops.attn_bwd(Q_, K_, V_, o, dO_, lse, ws, dq, dk, dv, lens, B, H, Tq, Tk, causal, drop_p=p, seed=11, stream_id=2, qkv_split=split)
        res.append((o, lse, dq, dk, dv))
    for a, b, name in zip(res[1], res[0], ("O", "LSE", "dQ", "dK", "dV")):
        assert relerr(a, b.cpu()) < 3e-5, name


def test_gemm_out_split_equals_split_of_fp32_output():
    """unast_gemm out_split = 1 stores exactly split_f32(fp32 result): bias / residual epilogue, column slices of a wider buffer."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(4)
    M, N, K = 640, 768, 256
    x = torch.randn(M, K, generator=g).to(D); W = (torch.randn(N, K, generator=g) * 0.1).to(D); b = torch.randn(N, generator=g).to(D)
    y = torch.empty(M, N, device=D); ys = torch.empty(M, N, device=D); ref = torch.empty(M, N, device=D)
    ops.linear_fwd(x, W, b, y)
    ops.linear_fwd(x, W, b, ys, out_split=True)
    ops.split_f32(y.view(-1), ref.view(-1))
    assert torch.equal(ys.view(torch.int32), ref.view(torch.int32))
    dx = torch.empty(M, K, device=D); dxs = torch.empty(M, K, device=D); refx = torch.empty(M, K, device=D)
    ops.linear_dgrad(y, W, dx)
    ops.linear_dgrad(y, W, dxs, out_split=True)
    ops.split_f32(dx.view(-1), refx.view(-1))
    assert torch.equal(dxs.view(torch.int32), refx.view(torch.int32))


def test_attention_dropout_consistency():
    """Dropout on P: forward keep-rate, determinism, and backward uses the same mask (finite-difference check on V)."""
    from unast_amd import ops
    B, H, T, E = 2, 4, 96, 256
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * T, 3 * E, generator=g).to(D)
    lens = torch.tensor([T, T - 10], dtype=torch.int32, device=D)
    O1 = torch.empty(B * T, E, device=D); O2 = torch.empty_like(O1); LSE = torch.empty(B, H, T, device=D)
    ops.attn_fwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], O1, LSE, lens, B, H, T, T, False, drop_p=0.3, seed=11, stream_id=3)
    ops.attn_fwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], O2, LSE, lens, B, H, T, T, False, drop_p=0.3, seed=11, stream_id=3)
    assert torch.equal(O1, O2)
    # O is linear in V for a fixed mask: dV from the backward must equal the adjoint of that linear map
    dO = torch.randn(B * T, E, generator=g).to(D)
    ws = torch.empty(B, H, T, device=D)
    dQ = torch.empty(B * T, E, device=D); dKV = torch.empty(B * T, 2 * E, device=D)
    ops.attn_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], O1, dO, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, T, T, False,
                 drop_p=0.3, seed=11, stream_id=3)
    dVdir = torch.randn(B * T, E, generator=g).to(D)
    qkv2 = qkv.clone(); qkv2[:, 2 * E:] += dVdir
    ops.attn_fwd(qkv2[:, :E], qkv2[:, E:2 * E], qkv2[:, 2 * E:], O2, LSE, lens, B, H, T, T, False, drop_p=0.3, seed=11, stream_id=3)
    lhs = ((O2 - O1).double() * dO.double()).sum().item()
    rhs = (dKV[:, E:].double() * dVdir.double()).sum().item()
    assert abs(lhs - rhs) < 2e-3 * max(abs(lhs), 1.0)
    ops.attn_fwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], O2, LSE, lens, B, H, T, T, False, drop_p=0.0)
    assert not torch.allclose(O1, O2)


@pytest.mark.parametrize("rows,C", [(333, 256), (1, 256), (37, 80), (130, 512), (65, 1024), (4099, 256)])
def test_layernorm_fwd_bwd(rows, C):
    """Every width class of the backward kernel (<= 256, <= 512, <= 1024 columns) and row counts that leave the four-row unroll a tail."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(1)
    z = (torch.randn(rows, C, generator=g, dtype=torch.float64) * 2 + 0.3).requires_grad_(True)
    w = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(z, (C,), w, b, 1e-5)
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    y = torch.empty(rows, C, device=D); mean = torch.empty(rows, device=D); rstd = torch.empty(rows, device=D)
    ops.layernorm_fwd(z.detach().float().to(D), w.detach().float().to(D), b.detach().float().to(D), y, mean, rstd)
    assert relerr(y, y_ref.detach()) < 1e-5
    dz = torch.empty(rows, C, device=D); dg = torch.ones(C, device=D); db = torch.ones(C, device=D)
    ops.layernorm_bwd(dy.float().to(D), z.detach().float().to(D), w.detach().float().to(D), mean, rstd, dz, None, dg, db)
    assert relerr(dz, z.grad) < 1e-5
    assert relerr(dg, w.grad + 1) < 1e-5 and relerr(db, b.grad + 1) < 1e-5
    # dropped second output: keep-rate and values
    dzd = torch.empty(rows, C, device=D)
    ops.layernorm_bwd(dy.float().to(D), z.detach().float().to(D), w.detach().float().to(D), mean, rstd, dz, dzd, None, None, drop_p=0.1, seed=3, stream_id=9)
    keep = dzd != 0
    assert abs(keep.float().mean().item() - 0.9) < (0.01 if rows * C > 50000 else 0.08)
    assert torch.allclose(dzd[keep], dz[keep] / 0.9, rtol=1e-5)
    # the same (seed, stream) mask as the GEMM epilogue dropout
    x = torch.ones(rows, 64, device=D); W = torch.ones(C, 64, device=D)
    yy = torch.empty(rows, C, device=D)
    ops.linear_fwd(x, W, None, yy, drop_p=0.1, seed=3, stream_id=9)
    assert torch.equal(yy != 0, keep)


@pytest.mark.parametrize("act,C", [(1, 256), (2, 256), (0, 80)])
def test_batchnorm_fwd_bwd(act, C):
    from unast_amd import ops
    rows = 517
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(rows, C, generator=g, dtype=torch.float64) * 1.5 + 0.7).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm = torch.zeros(C, dtype=torch.float64); rv = torch.ones(C, dtype=torch.float64)
    pre = torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
    y_ref = torch.relu(pre) if act == 1 else (torch.tanh(pre) if act == 2 else pre)
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    xd = x.detach().float().to(D); wd = w.detach().float().to(D); bd = b.detach().float().to(D)
    y = torch.empty(rows, C, device=D); mean = torch.empty(C, device=D); rstd = torch.empty(C, device=D)
    rmd = torch.zeros(C, device=D); rvd = torch.ones(C, device=D); ws = torch.empty(2 * C, dtype=torch.float64, device=D)
    ops.bn_fwd(xd, wd, bd, y, mean, rstd, rmd, rvd, ws, act)
    assert relerr(y, y_ref.detach()) < 1e-5
    assert relerr(rmd, rm) < 1e-5 and relerr(rvd, rv) < 1e-5
    dyd = dy.float().to(D).clone(); dx = torch.empty(rows, C, device=D); dg = torch.zeros(C, device=D); db = torch.zeros(C, device=D)
    ops.bn_bwd(dyd, xd, mean, rstd, wd, bd, dx, dg, db, ws, act)
    assert relerr(dx, x.grad) < 2e-5
    assert relerr(dg, w.grad) < 2e-5 and relerr(db, b.grad) < 2e-5


def test_embed_posenc_rowmask():
    from unast_amd import ops
    from unast_amd.portable import positional_table
    B, T, V, Dm = 3, 17, 46, 256
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (B, T), generator=g)
    E = torch.randn(V, Dm, generator=g)
    out = torch.empty(B * T, Dm, device=D)
    ops.embed_fwd(ids.to(D), E.to(D), out, T)
    assert torch.equal(out.cpu(), E[ids.view(-1)])
    ops.embed_fwd(ids.to(D), E.to(D), out, T, shift_sos=1)
    sh = torch.cat([torch.ones(B, 1, dtype=torch.long), ids[:, :-1]], 1)
    assert torch.equal(out.cpu(), E[sh.view(-1)])
    dout = torch.randn(B * T, Dm, generator=g)
    dE = torch.zeros(V, Dm, device=D)
    ops.embed_bwd(ids.to(D), dout.to(D), dE, T)
    ref = torch.zeros(V, Dm, dtype=torch.float64).index_add_(0, ids.view(-1), dout.double())
    ref[0] = 0
    assert relerr(dE, ref) < 1e-6
    pe = torch.from_numpy(positional_table(64, Dm))
    x = torch.randn(B * T, Dm, generator=g)
    y = torch.empty(B * T, Dm, device=D)
    ops.posenc_fwd(x.to(D), pe.to(D), y, T, 16.0)
    ref = (x.view(B, T, Dm) * 16.0 + pe[None, :T]).view(B * T, Dm)
    assert relerr(y, ref) < 1e-6
    dx = torch.empty(B * T, Dm, device=D)
    ops.posenc_bwd(dout.to(D), x.to(D), dx, 16.0)
    assert relerr(dx, dout.double() * 16.0 * (x > 0).double()) < 1e-6
    big = torch.ones(4096, 80, device=D); m = torch.empty_like(big)
    ops.rowmask(big, m, 0.3, 7, 2)
    rows_kept = (m.sum(1) == 80).float().mean().item()
    assert abs(rows_kept - 0.7) < 0.03 and ((m.sum(1) == 0) | (m.sum(1) == 80)).all()
    a = torch.randn(1001, generator=g); b = torch.randn(1001, generator=g)
    ad = a.to(D); ops.add_inplace(ad, b.to(D))
    assert torch.allclose(ad.cpu(), a + b)


def test_losses_match_oracle():
    from unast_amd import ops
    from oracle import unast_ref as R
    B, T, M, V = 3, 21, 80, 46
    g = torch.Generator().manual_seed(4)
    lens = torch.tensor([21, 9, 14])
    gold = torch.rand(B, T, M, generator=g)
    head = torch.zeros(B, T, 84); head[..., :81] = torch.randn(B, T, 81, generator=g)
    post = torch.randn(B, T, M, generator=g)
    hp = head[..., :80].clone().double().requires_grad_(True); st = head[..., 80].clone().double().requires_grad_(True)
    pp = post.clone().double().requires_grad_(True)
    gold_stop = torch.nn.functional.one_hot(lens - 1, T).double()
    ref = R.speech_loss(gold.double(), gold_stop, hp, pp, lens, st, 5.0)
    (ref * 0.5).backward()
    ws = torch.zeros(4, dtype=torch.float64, device=D); loss = torch.empty(1, device=D)       # zero on entry, left zero on exit
    li = lens.to(torch.int32).to(D)
    for _ in range(2):
        loss.fill_(-1.0)
        ops.speech_loss_fwd(gold.to(D), head.to(D), post.to(D), li, 5.0, ws, loss)
        assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
        assert bool((ws.view(torch.int64) == 0).all())
    gs = torch.tensor([0.5], device=D); dh = torch.empty(B, T, 84, device=D); dp = torch.empty(B, T, M, device=D)
    ops.speech_loss_bwd(gold.to(D), head.to(D), post.to(D), li, 5.0, gs, dh, dp)
    assert relerr(dh[..., :80], hp.grad) < 1e-5 and relerr(dh[..., 80], st.grad) < 1e-5 and relerr(dp, pp.grad) < 1e-5
    assert (dh[..., 81:] == 0).all()
    logits = torch.zeros(B * T, 48); logits[:, :V] = torch.randn(B * T, V, generator=g) * 2
    text = torch.randint(3, V, (B, T), generator=g)
    for b in range(B):
        text[b, lens[b] - 1] = 2; text[b, lens[b]:] = 0
    lr = logits[:, :V].clone().double().view(B, T, V).requires_grad_(True)
    ref = R.text_loss(text, lr, 3.0)
    (ref * 0.25).backward()
    ws2 = torch.zeros(8, dtype=torch.float64, device=D)
    for _ in range(2):
        loss.fill_(-1.0)
        ops.text_loss_fwd(logits.to(D), text.to(D).view(-1), V, 3.0, ws2, loss)
        assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
        assert bool((ws2[:4].view(torch.int64) == 0).all()) and float(ws2[4]) > 0
    ops.speech_loss_fwd(gold.to(D), head.to(D), post.to(D), li, 5.0, ws2, loss)          # a slot the text loss has used serves the speech loss
    assert abs(loss.item() - R.speech_loss(gold.double(), gold_stop, hp.detach(), pp.detach(), lens, st.detach(), 5.0).item()) < 1e-5 * abs(loss.item())
    ops.text_loss_fwd(logits.to(D), text.to(D).view(-1), V, 3.0, ws2, loss)
    gs = torch.tensor([0.25], device=D); dl = torch.empty(B * T, 48, device=D)
    ops.text_loss_bwd(logits.to(D), text.to(D).view(-1), V, 3.0, ws2, gs, dl)
    assert relerr(dl[:, :V].reshape(B, T, V), lr.grad) < 1e-5 and (dl[:, V:] == 0).all()
    x = torch.randn(8, generator=g); perm = torch.randperm(8, generator=g)
    for flip in (0, 1):
        tgt = torch.where(perm < 4, torch.tensor(0.9), torch.tensor(0.1))
        if flip:
            tgt = 1 - tgt
        xr = x.clone().double().requires_grad_(True)
        ref = R.bce_logits_mean(xr, tgt.double()); (ref * 0.5).backward()
        dlg = torch.zeros(8, 4, device=D)
        tg = torch.empty(8, device=D)
        ops.disc_targets(perm.to(D), 4, flip, tg)
        assert torch.allclose(tg.cpu(), tgt)
        xs = torch.zeros(8, 4, device=D); xs[:, 0] = x.to(D)
        ops.bce_logits(xs, 4, tg, 8, loss, torch.tensor([0.5], device=D), dlg, 4)
        assert abs(loss.item() - ref.item()) < 1e-5 and relerr(dlg[:, 0], xr.grad) < 1e-5


def test_lstm_fwd_bwd_matches_torch_packed():
    from unast_amd import ops
    Bd, T, H, ndir = 5, 23, 64, 2
    g = torch.Generator().manual_seed(6)
    lens = torch.tensor([23, 5, 17, 1, 12])
    lstm = torch.nn.LSTM(128, H, num_layers=1, bidirectional=True, batch_first=True).double()
    x = torch.randn(Bd, T, 128, generator=g, dtype=torch.float64, requires_grad=True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    out, (hn, cn) = lstm(packed)
    out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    dy = torch.randn(Bd, T, 2 * H, generator=g, dtype=torch.float64)
    dhf = torch.randn(Bd, 2 * H, generator=g, dtype=torch.float64)
    hcat = torch.cat([hn[0], hn[1]], -1)
    ((out * dy).sum() + (hcat * dhf).sum()).backward()
    wih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]).detach()
    whh = torch.cat([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]).detach().float().to(D).contiguous()
    bih = torch.cat([lstm.bias_ih_l0, lstm.bias_ih_l0_reverse]).detach().float().to(D)
    bhh = torch.cat([lstm.bias_hh_l0, lstm.bias_hh_l0_reverse]).detach().float().to(D)
    xproj = (x.detach() @ wih.t()).float().to(D).contiguous()
    li = lens.to(torch.int32).to(D)
    y = torch.zeros(Bd, T, 2 * H, device=D); gates = torch.empty(Bd, T, ndir, 4 * H, device=D); cs = torch.empty(Bd, T, ndir, H, device=D)
    hprev = torch.zeros(Bd, T, ndir, H, device=D); hfin = torch.empty(Bd, 2 * H, device=D)
    ops.lstm_fwd(xproj, whh, bih, bhh, li, y, gates, cs, hprev, hfin, ndir, 4 * H * H, 4 * H)
    assert relerr(y, out.detach()) < 2e-5
    assert relerr(hfin, hcat.detach()) < 2e-5
    dg = torch.zeros(Bd, T, ndir, 4 * H, device=D)
    ops.lstm_bwd(dy.float().to(D), dhf.float().to(D), whh, gates, cs, li, dg, ndir, 4 * H * H)
    dg2 = dg.view(Bd * T, ndir * 4 * H).double().cpu()
    dx = dg2 @ wih
    assert relerr(dx.view(Bd, T, 128), x.grad) < 5e-5
    dwih = dg2.t() @ x.detach().view(Bd * T, 128)
    assert relerr(dwih, torch.cat([lstm.weight_ih_l0.grad, lstm.weight_ih_l0_reverse.grad])) < 5e-5
    hp = hprev.view(Bd * T, ndir, H).double().cpu()
    dwhh_f = dg2[:, :4 * H].t() @ hp[:, 0]
    dwhh_r = dg2[:, 4 * H:].t() @ hp[:, 1]
    assert relerr(dwhh_f, lstm.weight_hh_l0.grad) < 5e-5 and relerr(dwhh_r, lstm.weight_hh_l0_reverse.grad) < 5e-5
    assert relerr(dg2.sum(0), torch.cat([lstm.bias_ih_l0.grad, lstm.bias_ih_l0_reverse.grad])) < 5e-5


def test_order_fixed_column_sums_and_embedding_gradient():
    """unast_colsum_det / unast_embed_bwd_det (config.DETERMINISTIC_SUMS): the same sums as the atomic kernels, to fp32 rounding against
    fp64, and bit-identical from run to run."""
    from unast_amd import config, ops
    g = torch.Generator().manual_seed(3)
    for rows, C in ((5000, 256), (777, 46), (1, 81), (25600, 1024)):
        x = torch.randn(rows, C, generator=g).to(D)
        ref = x.double().sum(0).cpu()
        outs = []
        for rep in range(2):
            out = torch.full((C,), 0.5, device=D)
            config.DETERMINISTIC_SUMS = True
            try:
                ops.colsum(x, out)
            finally:
                config.DETERMINISTIC_SUMS = False
            outs.append(out.clone())
            assert relerr(out - 0.5, ref) < 2e-6, (rows, C)
        assert torch.equal(outs[0], outs[1])
        a = torch.full((C,), 0.5, device=D); ops.colsum(x, a)
        assert relerr(a, outs[0].cpu()) < 2e-6
    V, Dm, B, T = 46, 256, 6, 37
    ids = torch.randint(0, V, (B, T), generator=g).to(D)
    dout = torch.randn(B * T, Dm, generator=g).to(D)
    for shift, kw in ((-1, {}), (1, {}), (-1, dict(drop_p=0.2, seed=5, stream_id=3, noise_p=0.1, noise_stream=4))):
        a = torch.zeros(V, Dm, device=D); b = torch.zeros(V, Dm, device=D); c = torch.zeros(V, Dm, device=D)
        ops.embed_bwd(ids, dout, a, T, shift_sos=shift, **kw)
        config.DETERMINISTIC_SUMS = True
        try:
            ops.embed_bwd(ids, dout, b, T, shift_sos=shift, **kw)
            ops.embed_bwd(ids, dout, c, T, shift_sos=shift, **kw)
        finally:
            config.DETERMINISTIC_SUMS = False
        assert torch.equal(b, c) and relerr(a, b.cpu()) < 2e-6 and float(b[0].abs().max()) == 0.0, (shift, kw)


def test_leaky_dropout_and_disc_gather():
    from unast_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(64, 64, generator=g)
    y = torch.empty(64, 64, device=D)
    ops.leaky_dropout(x.to(D), None, y, 0.2)
    assert relerr(y, torch.nn.functional.leaky_relu(x.double(), 0.2)) < 1e-6
    dy = torch.randn(64, 64, generator=g); dx = torch.empty(64, 64, device=D)
    ops.leaky_dropout(x.to(D), dy.to(D), dx, 0.2)
    assert relerr(dx, dy.double() * torch.where(x > 0, 1.0, 0.2).double()) < 1e-6
    B, Tt, Ts, Dm = 3, 7, 19, 256
    th = torch.randn(B, Tt, Dm, generator=g); sh = torch.randn(B, Ts, Dm, generator=g)
    tl = torch.tensor([7, 4, 2], dtype=torch.int32); sl = torch.tensor([19, 8, 11], dtype=torch.int32)
    perm = torch.randperm(2 * B, generator=g)
    out = torch.empty(2 * B, Ts, Dm, device=D); ol = torch.empty(2 * B, dtype=torch.int32, device=D)
    ops.disc_gather(th.to(D), sh.to(D), tl.to(D), sl.to(D), perm.to(D), out, ol)
    full = torch.cat([torch.nn.functional.pad(th, (0, 0, 0, Ts - Tt)), sh])[perm]
    assert torch.equal(out.cpu(), full) and torch.equal(ol.cpu(), torch.cat([tl, sl])[perm])
    dth = torch.empty(B, Tt, Dm, device=D); dsh = torch.empty(B, Ts, Dm, device=D)
    ops.disc_scatter(out, perm.to(D), dth, dsh)
    assert torch.equal(dth.cpu(), th) and torch.equal(dsh.cpu(), sh)


def test_adamw_matches_oracle():
    from unast_amd import ops
    from oracle import unast_ref as R
    g = torch.Generator().manual_seed(9)
    n = 10007
    p0 = torch.randn(n, generator=g); grads = [torch.randn(n, generator=g) * s for s in (3.0, 0.01)]
    pr = {"w": p0.clone().requires_grad_(True)}
    opt = R.AdamW(pr, lr=1e-3, weight_decay=1e-2)
    p = p0.to(D).clone(); m = torch.zeros(n, device=D); v = torch.zeros(n, device=D)
    ss = torch.zeros(1, dtype=torch.float64, device=D)
    for step, gr in enumerate(grads, 1):
        pr["w"].grad = gr.clone()
        total = opt.step(1.0)
        ss.zero_(); ops.sumsq(gr.to(D), ss)
        assert abs(math.sqrt(ss.item()) - total.item()) < 1e-4 * total.item()
        ops.adamw(p, gr.to(D), m, v, ss, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
        assert relerr(p, pr["w"].detach()) < 1e-6


def test_adam_l2_form_and_device_hyper_parameters():
    """optim_type 'adam' (src/train.py:929-930: torch.optim.Adam, weight decay added to the gradient) against torch's own
    Adam in fp64 behind clip_grad_norm_, and the device-resident {lr, 1-b1^t, sqrt(1-b2^t)} block a captured step reads."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(10)
    n = 4099
    p0 = torch.randn(n, generator=g); grads = [torch.randn(n, generator=g) * s for s in (2.0, 0.05, 1.0)]
    ref = torch.nn.Parameter(p0.double().clone())
    opt = torch.optim.Adam([ref], lr=2e-3, weight_decay=1e-2)
    p = p0.to(D).clone(); m = torch.zeros(n, device=D); v = torch.zeros(n, device=D)
    p2 = p.clone(); m2 = m.clone(); v2 = v.clone()
    ss = torch.zeros(1, dtype=torch.float64, device=D)
    hyper = torch.zeros(3, device=D)
    for step, gr in enumerate(grads, 1):
        ref.grad = gr.double().clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        ss.zero_(); ops.sumsq(gr.to(D), ss)
        ops.adamw(p, gr.to(D), m, v, ss, 1.0, 2e-3, 0.9, 0.999, 1e-8, 1e-2, step, decoupled=False)
        assert relerr(p, ref.detach()) < 2e-6, step
        hyper.copy_(torch.tensor(ops.adam_hyper(2e-3, 0.9, 0.999, step), dtype=torch.float64).float())
        ops.adamw(p2, gr.to(D), m2, v2, ss, 1.0, 123.0, 0.9, 0.999, 1e-8, 1e-2, 0, decoupled=False, dev_hyper=hyper)
        assert torch.equal(p2, p) and torch.equal(m2, m) and torch.equal(v2, v), step


def test_specaugment_spans():
    from unast_amd import ops
    B, T, M = 4, 300, 80
    g = torch.Generator().manual_seed(10)
    mel = torch.rand(B, T, M, generator=g)
    lens = torch.tensor([300, 150, 40, 10], dtype=torch.int32)
    out = torch.empty(B, T, M, device=D)
    ops.specaugment(mel.to(D), lens.to(D), out, 1, 2)
    o = out.cpu()
    for b in range(B):
        changed = (o[b] != mel[b]).any(1)
        assert changed.sum() <= 119 and not changed[lens[b]:].any()      # two time spans of widths < 20 and < 100
        if changed.any():
            assert torch.allclose(o[b][changed], mel[b].mean().expand(int(changed.sum()), M), atol=1e-5)
        assert ((o[b] != mel[b]).sum(1)[changed] >= M - 1).all()         # whole rows, never frequency columns


@pytest.mark.parametrize("rows,V", [(300, 46), (128, 46), (5760, 46), (77, 12)])
def test_text_head_and_loss_in_one_launch(rows, V):
    """unast_text_head_loss (head GEMM + weighted cross-entropy + d loss / d logits from the GEMM's accumulators) against fp64 torch:
    logits, the loss (ignore_index 0, EOS weight), and gscale * dlogits; and against the three separate launches it replaces."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(rows + V)
    x = torch.randn(rows, 256, generator=g); W = torch.randn(V, 256, generator=g) * 0.1; b = torch.randn(V, generator=g)
    gold = torch.randint(0, V, (rows,), generator=g)
    gold[::7] = 0; gold[3::11] = 2
    eos_w, gs = 5.0, 0.5
    ldl = (V + 3) // 4 * 4
    xd, Wd, bd, gd = x.to(D), W.to(D), b.to(D), gold.to(D)
    logits = torch.full((rows, ldl), float("nan"), device=D); dl = torch.full((rows, ldl), float("nan"), device=D)
    ws = torch.zeros(8, dtype=torch.float64, device=D); loss = torch.empty(1, device=D)
    ops.text_head_loss(xd, Wd, bd, gd, V, eos_w, gs, logits, dl, ws, loss)
    x64 = x.double().requires_grad_(False)
    lg = (x64 @ W.double().t() + b.double()).requires_grad_(True)
    wvec = torch.ones(V, dtype=torch.float64); wvec[2] = eos_w
    ref = torch.nn.functional.cross_entropy(lg, gold, weight=wvec, ignore_index=0)
    (ref * gs).backward()
    assert relerr(logits[:, :V], lg.detach()) < 2e-5
    assert abs(float(loss) - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert relerr(dl[:, :V], lg.grad) < 2e-5
    assert not torch.isnan(logits).any() and bool((dl[:, V:] == 0).all()) and bool((ws[:4] == 0).all())
    # the three launches it replaces
    l2 = torch.zeros(rows, ldl, device=D); ops.linear_fwd(xd, Wd, bd, l2[:, :V])
    loss2 = torch.empty(1, device=D); ops.text_loss_fwd(l2, gd, V, eos_w, ws, loss2)
    dl2 = torch.empty(rows, ldl, device=D); ops.text_loss_bwd(l2, gd, V, eos_w, ws, torch.tensor([gs], device=D), dl2)
    assert relerr(logits[:, :V], l2[:, :V].cpu()) < 1e-5 and abs(float(loss) - float(loss2)) < 2e-6 and relerr(dl, dl2.cpu()) < 2e-5


@pytest.mark.parametrize("B,T", [(3, 50), (2, 128), (32, 800)])
def test_speech_heads_and_loss_terms_in_two_launches(B, T):
    """unast_speech_head_loss (head GEMM + pre-net MSE + stop BCE + d loss / d head) followed by unast_speech_post_loss (post-net MSE,
    d loss / d post, the scalar) against the three launches they replace (head GEMM, speech_loss_fwd, speech_loss_bwd) and, at the small
    sizes, against fp64 torch."""
    from unast_amd import ops
    M, ldh = 80, 84
    g = torch.Generator().manual_seed(B * T)
    N = B * T
    x = torch.randn(N, 256, generator=g); W = torch.randn(M + 1, 256, generator=g) * 0.1; b = torch.randn(M + 1, generator=g)
    gold = torch.rand(B, T, M, generator=g); post = torch.rand(B, T, M, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32); lens[0] = T
    eos_w, gs = 5.0, 0.5
    xd, Wd, bd, gd, pd, ld = x.to(D), W.to(D), b.to(D), gold.to(D), post.to(D), lens.to(D)
    head = torch.full((N, ldh), float("nan"), device=D); dh = torch.full((N, ldh), float("nan"), device=D)
    ws = torch.zeros(8, dtype=torch.float64, device=D); loss = torch.empty(1, device=D); dp = torch.full((B, T, M), float("nan"), device=D)
    ops.speech_head_loss(xd, Wd, bd, gd.view(N, M), ld, B, T, M, eos_w, gs, head, dh, ws)
    ops.speech_post_loss(gd.view(N, M), pd.view(N, M), ld, B, T, M, gs, dp, ws, loss)
    h2 = torch.zeros(N, ldh, device=D); ops.linear_fwd(xd, Wd, bd, h2[:, :M + 1])
    loss2 = torch.empty(1, device=D); ops.speech_loss_fwd(gd, h2.view(B, T, ldh), pd, ld, eos_w, ws, loss2)
    dh2 = torch.empty(B, T, ldh, device=D); dp2 = torch.empty(B, T, M, device=D)
    ops.speech_loss_bwd(gd, h2.view(B, T, ldh), pd, ld, eos_w, torch.tensor([gs], device=D), dh2, dp2)
    assert relerr(head[:, :M + 1], h2[:, :M + 1].cpu()) < 1e-5 and bool((head[:, M + 1:] == 0).all())
    assert abs(float(loss) - float(loss2)) < 3e-6 * max(1.0, abs(float(loss2)))
    assert relerr(dh, dh2.view(N, ldh).cpu()) < 2e-5 and relerr(dp, dp2.cpu()) < 1e-6 and bool((ws[:4] == 0).all())
    if N <= 300:
        x64, W64 = x.double(), W.double()
        hh = (x64 @ W64.t() + b.double()).requires_grad_(True)
        p64 = post.double().requires_grad_(True)
        mask = (torch.arange(T)[None, :] < lens[:, None]).double()[..., None]
        pre = hh[:, :M].view(B, T, M)
        denom = mask.sum() * M
        y = torch.nn.functional.one_hot(lens.long() - 1, T).double()
        pw = 1.0 + (eos_w - 1.0) * y
        st = hh[:, M].view(B, T)
        bce = ((1 - y) * st + pw * (torch.log1p(torch.exp(-st.abs())) + torch.clamp(-st, min=0))).mean()
        ref = ((gold.double() - pre) ** 2 * mask).sum() / denom + ((gold.double() - p64) ** 2 * mask).sum() / denom + bce
        (ref * gs).backward()
        assert abs(float(loss) - ref.item()) < 3e-6 * max(1.0, abs(ref.item()))
        assert relerr(dh[:, :M + 1], hh.grad) < 3e-5 and relerr(dp, p64.grad) < 1e-5
