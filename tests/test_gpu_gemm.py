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
                                   (40, 256, 80), (3000, 1024, 256), (2500, 1100, 64), (1024, 256, 512), (2048, 1024, 256), (256, 128, 96), (96, 256, 256), (32, 256, 128)])
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
    # wgrad accumulates (and carries the fused bias gradient)
    dW = torch.ones(N, K, device=dev())
    db = torch.ones(N, device=dev())
    ops.linear_wgrad(dy.to(dev())[:, :N], x.to(dev()), dW, db=db)
    assert relerr(dW, 1.0 + dy[:, :N].double().t() @ x.double()) < TOL[nsplit]
    assert relerr(db, 1.0 + dy[:, :N].double().sum(0)) < 1e-5
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


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("M,N,K", [(1024, 256, 512), (384, 768, 256), (200, 81, 256), (130, 256, 80)])
def test_presplit_weights_are_bit_identical(nsplit, M, N, K):
    """Weights read from the pre-split copy (written by unast_split_f32 / the AdamW kernel) give the same bits as weights
    split inside the GEMM, on the interior fast path and on the general path, forward and dgrad; conv forms as well."""
    from unast_amd import ops, config
    config.NSPLIT = nsplit
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev())
    W = (torch.randn(N, K, generator=g) * 0.1).to(dev())
    Ws = torch.empty_like(W)
    ops.split_f32(W, Ws)
    dy = torch.randn(M, N + (-N) % 4, generator=g).to(dev())[:, :N]
    outs = []
    for presplit in (False, True):
        if presplit:
            ops.register_weight_span(W.data_ptr(), W.numel() * 4, Ws.data_ptr())
        try:
            y = torch.empty(M, N + (-N) % 4, device=dev())
            ops.linear_fwd(x, W, None, y[:, :N])
            dx = torch.empty(M, K, device=dev())
            ops.linear_dgrad(dy, W, dx)
            outs.append((y[:, :N].clone(), dx))
        finally:
            ops.unregister_weight_span(W.data_ptr())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert relerr(outs[1][0], x.double().cpu() @ W.double().cpu().t()) < TOL[nsplit]
    config.NSPLIT = 3


def test_presplit_conv_and_adamw_refresh():
    from unast_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, Cin, Cout = 2, 40, 256, 256
    x = torch.randn(B, T, Cin, generator=g).to(dev())
    Wp = (torch.randn(Cout, 5, Cin, generator=g) * 0.05).to(dev())
    dy = torch.randn(B, T, Cout, generator=g).to(dev())
    ref_y = torch.empty(B, T, Cout, device=dev()); ops.conv_fwd(x, Wp, None, ref_y, 2)
    ref_dx = torch.empty(B, T, Cin, device=dev()); ops.conv_dgrad(dy, Wp, ref_dx, 2)
    Ws = torch.empty_like(Wp); ops.split_f32(Wp, Ws)
    ops.register_weight_span(Wp.data_ptr(), Wp.numel() * 4, Ws.data_ptr())
    try:
        y = torch.empty(B, T, Cout, device=dev()); ops.conv_fwd(x, Wp, None, y, 2)
        dx = torch.empty(B, T, Cin, device=dev()); ops.conv_dgrad(dy, Wp, dx, 2)
    finally:
        ops.unregister_weight_span(Wp.data_ptr())
    assert torch.equal(y, ref_y) and torch.equal(dx, ref_dx)
    # the AdamW kernel refreshes the split copy of what it updates
    n = 4096
    p = torch.randn(n, generator=g).to(dev()); gr = torch.randn(n, generator=g).to(dev())
    m = torch.zeros(n, device=dev()); v = torch.zeros(n, device=dev()); ss = torch.zeros(1, dtype=torch.float64, device=dev())
    sp = torch.zeros(n, device=dev())
    ops.sumsq(gr, ss)
    ops.adamw(p, gr, m, v, ss, 1.0, 1e-2, 0.9, 0.999, 1e-8, 1e-6, 1, split_out=sp)
    want = torch.empty(n, device=dev()); ops.split_f32(p, want)
    assert torch.equal(sp.view(torch.int32), want.view(torch.int32))
    # format: per 4 values [hi x4 | lo x4] bf16, hi + lo ~ value to ~2^-16
    chunks = want.view(torch.int16).view(-1, 8).cpu()
    hi = (chunks[:, :4].to(torch.int32) << 16).view(torch.float32)
    lo = (chunks[:, 4:].to(torch.int32) << 16).view(torch.float32)
    assert ((hi + lo).view(-1) - p.cpu()).abs().max() <= p.abs().max().item() * 2 ** -15


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("M,N,K", [(1024, 256, 512), (200, 81, 256), (130, 256, 80), (513, 1024, 256), (96, 384, 64), (256, 128, 32)])
def test_tile_variants_agree_bitwise(nsplit, M, N, K):
    """The 8-wave and 4-wave tilings (interior fast path or general path) give the same bits: same products, same k order."""
    from unast_amd import ops, config
    config.NSPLIT = nsplit
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, generator=g).to(dev()); W = (torch.randn(N, K, generator=g) * 0.1).to(dev()); b = torch.randn(N, generator=g).to(dev())
    ldn = (N + 3) // 4 * 4
    dy = torch.zeros(M, ldn, device=dev()); dy[:, :N] = torch.randn(M, N, generator=g).to(dev())
    outs = []
    for wn in (8, 2):
        y = torch.zeros(M, ldn, device=dev())
        ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, ldn, M, N, K, bias=b, act=1, tile_wn=wn)
        dx = torch.empty(M, K, device=dev())
        ops.gemm(ops.OP_KC, ops.OP_RC, dy, ldn, W, K, dx, K, M, K, ldn, kb_valid=N, tile_wn=wn)
        dW = torch.zeros(N, K, device=dev()); db = torch.zeros(N, device=dev())
        ops.gemm(ops.OP_RC, ops.OP_RC, dy, ldn, x, K, dW, K, N, K, M, beta=1, rowsum_a=db, tile_wn=wn)
        outs.append((y, dx, dW, db))
    assert relerr(outs[0][0][:, :N], torch.relu(x.double().cpu() @ W.double().cpu().t() + b.double().cpu())) < TOL[nsplit]
    assert relerr(outs[0][2], dy[:, :N].double().cpu().t() @ x.double().cpu()) < TOL[nsplit]
    assert relerr(outs[0][3], dy[:, :N].double().cpu().sum(0)) < 1e-5
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2])
    config.NSPLIT = 3


@pytest.mark.parametrize("nsplit", [3, 1])
def test_grouped_weight_gradients_match_fp64_and_single_launches(nsplit):
    """unast_wgrad_group: the weight (and bias) gradients of several linears -- different shapes, different token counts, one
    of them accumulating into a non-zero buffer -- from ONE launch, against fp64 and against the one-by-one split-K path."""
    from unast_amd import ops, config
    config.NSPLIT = nsplit
    g = torch.Generator().manual_seed(21)
    probs = [(25600, 256, 256), (25600, 768, 256), (5760, 512, 256), (4096, 256, 1024), (4096, 1024, 256)]      # (tokens, out, in)
    data = []
    for (M, N, K) in probs:
        dy = torch.randn(M, N, generator=g).to(dev()); x = torch.randn(M, K, generator=g).to(dev())
        dW0 = torch.randn(N, K, generator=g).to(dev()); db0 = torch.randn(N, generator=g).to(dev())
        data.append((dy, x, dW0, db0))
    outs = []
    for grouped in (True, False):
        config.WGRAD_GROUP = grouped
        res = []
        with ops.wgrad_batch():
            for dy, x, dW0, db0 in data:
                dW, db = dW0.clone(), db0.clone()
                ops.linear_wgrad(dy, x, dW, db=db)
                res.append((dW, db))
            if grouped:
                assert ops._BATCH is not None and len(ops._BATCH) == len(data)       # all five deferred
        torch.cuda.synchronize()
        outs.append(res)
    config.WGRAD_GROUP = True
    for (dy, x, dW0, db0), (gW, gb), (sW, sb) in zip(data, outs[0], outs[1]):
        ref = dW0.double().cpu() + dy.double().cpu().t() @ x.double().cpu()
        assert relerr(gW, ref) < TOL[nsplit] and relerr(sW, ref) < TOL[nsplit]
        assert relerr(gb, db0.double().cpu() + dy.double().cpu().sum(0)) < 1e-5
    config.NSPLIT = 3


def test_grouped_weight_gradients_general_shapes():
    """The group entry point on shapes off the interior fast path (rows / columns not multiples of 128, a token count that is
    not a multiple of 32, strided operand views) -- straight through the C ABI."""
    import ctypes
    from unast_amd import ops
    from unast_amd._lib import lib, check
    g = torch.Generator().manual_seed(22)
    probs = [(1000, 192, 320), (777, 46, 256), (2048, 256, 80)]
    arr = (ops._WgradItem * len(probs))()
    keep, refs = [], []
    for j, (M, N, K) in enumerate(probs):
        ldn, ldk = (N + 3) // 4 * 4 + 8, K + 4
        dyb = torch.zeros(M, ldn, device=dev()); xb = torch.zeros(M, ldk, device=dev())
        dy = torch.randn(M, N, generator=g); x = torch.randn(M, K, generator=g)
        dyb[:, :N] = dy.to(dev()); xb[:, :K] = x.to(dev())
        dW = torch.zeros(N, K, device=dev()); db = torch.zeros(N, device=dev())
        a = arr[j]
        a.A, a.lda, a.B, a.ldb, a.C, a.ldc = dyb.data_ptr(), ldn, xb.data_ptr(), ldk, dW.data_ptr(), K
        a.rowsum_a, a.M, a.N, a.K = db.data_ptr(), N, K, M
        keep.append((dyb, xb, dW, db)); refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    ptr = ctypes.cast(arr, ctypes.c_void_p)
    n = lib().unast_wgrad_group_ws_floats(len(probs), ptr, 0)
    ws = torch.empty(n, device=dev())
    check(lib().unast_wgrad_group(3, len(probs), ptr, ws.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream), "unast_wgrad_group")
    for (dyb, xb, dW, db), (rW, rb) in zip(keep, refs):
        assert relerr(dW, rW) < 3e-5 and relerr(db, rb) < 1e-5
    # too small a workspace is refused, not overrun
    assert lib().unast_wgrad_group(3, len(probs), ptr, ws.data_ptr(), n - 1, 0, torch.cuda.current_stream().cuda_stream) != 0


def test_dgrad_through_transposed_presplit_weights():
    """Input gradients of stored weights read W^T from the transposed pre-split copy (engine.FlatStore.dgrad_T): same results as
    the untransposed form for whole matrices, row slices (cross-attention's q / kv halves of in_proj), padded heads (81 -> 84,
    46 -> 48 rows) and the LSTM's combined [fwd | reverse] input weights -- before and after an optimizer step refreshed it."""
    from unast_amd import ops, config, train, utils
    from unast_amd.configs import make_args
    train.DEVICE = dev()
    utils.set_seed(3)
    args = make_args(num_layers=1, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
    config.DGRAD_TRANSPOSED = True            # opt-in feature (config.py): the store builds the transposed copies when it is on
    try:
        _, _, model, opt, _ = train.initialize_model(args)
        st = model._store()
        _dgrad_T_body(st, opt)
    finally:
        config.DGRAD_TRANSPOSED = False


def _dgrad_T_body(st, opt):
    from unast_amd import ops, config
    g = None
    st.sync_split()
    g = torch.Generator().manual_seed(5)
    P = st.phys
    inp = P["speech_m.decoder.transformer_decoder.layers.0.multihead_attn.in_proj_weight"]
    cases = [("whole", P["text_m.encoder.transformer_encoder.layers.0.linear1.weight"]), ("q-slice", inp[:256]), ("kv-slice", inp[256:]),
             ("head81", st.span("speech_m.postnet.linear_project.weight", "speech_m.postnet.stop_linear.weight", (81, 256))),
             ("logits46", P["text_m.postnet.fc1.weight"]),
             ("lstm", st.span("discriminator.rnn.rnn.weight_ih_l0", "discriminator.rnn.rnn.weight_ih_l0_reverse", (512, 256))),
             ("fc2", P["discriminator.fc2.weight"])]

    def check_all():
        for name, W in cases:
            N, K = W.shape
            assert st.dgrad_T(W) is not None, name
            Np = (N + 3) // 4 * 4
            M = 384
            dyb = torch.zeros(M, Np, device=dev()); dyb[:, :N] = torch.randn(M, N, generator=g).to(dev())
            R = torch.randn(M, K, generator=g).to(dev()); G = torch.randn(M, K, generator=g).to(dev())
            outs = []
            for flag in (True, False):
                config.DGRAD_TRANSPOSED = flag
                dx = torch.empty(M, K, device=dev())
                ops.linear_dgrad(dyb[:, :N], W, dx, R=R, G=G, gate_scale=1.25)
                outs.append(dx)
            config.DGRAD_TRANSPOSED = True
            ref = (dyb[:, :N].double().cpu() @ W.double().cpu()) * (G.cpu() > 0).double() * 1.25 + R.double().cpu()
            assert relerr(outs[0], ref) < 3e-5, name
            assert relerr(outs[0], outs[1].double().cpu()) < 1e-6, name
    check_all()
    # one optimizer step over both regions moves the weights: the transposed copies follow
    st.grad.normal_(generator=None)
    st.touched = {"gen", "disc"}
    before = st.flat.clone()
    opt.param_groups[0]["lr"] = 1e-2
    opt.step(max_norm=1.0)
    opt.zero_grad()
    assert not torch.equal(before, st.flat)
    check_all()


@pytest.mark.parametrize("B,T,Cin,Cout", [(3, 37, 80, 256), (32, 800, 256, 512), (2, 5, 256, 256), (5, 130, 512, 80)])
def test_conv_forward_epilogue_column_statistics(B, T, Cin, Cout):
    """unast_gemm colstats (the conv forward's epilogue) = per-channel sum and sum of squares of the conv output incl. bias, over
    all B*T positions -- ragged row / column tiles included -- and BatchNorm from those sums equals BatchNorm from its own pass."""
    from unast_amd import ops, config
    config.NSPLIT = 3
    D = dev()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, T, Cin, generator=g).to(D); Wp = (torch.randn(Cout, 5, Cin, generator=g) / (5 * Cin) ** 0.5).to(D); b = torch.randn(Cout, generator=g).to(D)
    y = torch.empty(B, T, Cout, device=D); ws = torch.zeros(2 * Cout, dtype=torch.float64, device=D)
    ops.conv_fwd(x, Wp, b, y, 2, colstats=ws)
    y0 = torch.empty(B, T, Cout, device=D)
    ops.conv_fwd(x, Wp, b, y0, 2)
    assert torch.equal(y, y0)                                   # the statistics do not touch the output
    yd = y.view(-1, Cout).double()
    s1, s2 = yd.sum(0), (yd * yd).sum(0)
    assert float((ws[:Cout] - s1).abs().max()) < 1e-4 * max(1.0, float(s1.abs().max()))
    assert float(((ws[Cout:] - s2) / s2).abs().max()) < 1e-5
    N = B * T
    gamma = torch.randn(Cout, generator=g).to(D); beta = torch.randn(Cout, generator=g).to(D)
    outs = []
    for have in (True, False):
        out = torch.empty(N, Cout, device=D); mean = torch.empty(Cout, device=D); rstd = torch.empty(Cout, device=D)
        rm = torch.zeros(Cout, device=D); rv = torch.ones(Cout, device=D)
        w = ws.clone() if have else torch.empty(2 * Cout, dtype=torch.float64, device=D)
        ops.bn_fwd(y.view(N, Cout), gamma, beta, out, mean, rstd, rm, rv, w, 1, have_sums=have)
        outs.append((out, mean, rstd, rm, rv))
    for a, c in zip(outs[0], outs[1]):
        assert relerr(a, c.cpu()) < 2e-5          # fp32 partial sums in a different order (variance = E[x^2] - E[x]^2 cancels at small B*T)

