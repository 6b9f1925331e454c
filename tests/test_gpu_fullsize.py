"""Parity at BASELINE.json's full sizes (config 3: B=32, T_text=180, T_mel=800, d=256, 4 heads, L=4) through size-independent
properties, and against the pinned oracle at full sequence length with a batch the CPU finishes in seconds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
B, TT, TM, H, E = 32, 180, 800, 4, 256


def relerr(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build(L, lr):
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
    train.DEVICE = D
    utils.set_seed(0)
    utils.set_deterministic(True)
    _, _, model, opt, _ = train.initialize_model(args)
    sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()}
    model.load_state_dict(sd)
    opt.param_groups[0]["lr"] = lr
    return args, model, opt, sd


def test_gemm_full_size_linearity_and_row_samples():
    """25600x256x1024 (FFN2 forward of config 3): 64 sampled rows against fp64, linearity in A, and the wgrad of the same
    size against an fp64 reference built from a sparse operand (only 512 token rows are non-zero)."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(0)
    M, N, K = B * TM, 256, 1024
    x1 = torch.randn(M, K, generator=g).to(D); x2 = torch.randn(M, K, generator=g).to(D)
    W = (torch.randn(N, K, generator=g) * 0.05).to(D); b = torch.randn(N, generator=g).to(D)
    y1 = torch.empty(M, N, device=D); y2 = torch.empty(M, N, device=D); y12 = torch.empty(M, N, device=D)
    ops.linear_fwd(x1, W, None, y1); ops.linear_fwd(x2, W, None, y2); ops.linear_fwd(x1 + x2, W, None, y12)
    assert relerr(y12, y1 + y2) < 2e-5
    rows = torch.randint(0, M, (64,), generator=g)
    ref = x1[rows].double().cpu() @ W.double().cpu().t()
    assert relerr(y1[rows], ref) < 3e-5
    yb = torch.empty(M, N, device=D); ops.linear_fwd(x1, W, b, yb)
    assert relerr(yb - y1, b.expand(M, N)) < 1e-4
    # weight gradient over all 25600 tokens, reference from the 512 non-zero rows
    dy = torch.zeros(M, N, device=D); nz = torch.randperm(M, generator=g)[:512]
    dy[nz] = torch.randn(512, N, generator=g).to(D)
    dW = torch.zeros(N, K, device=D); db = torch.zeros(N, device=D)
    ops.linear_wgrad(dy, x1, dW, db=db)
    assert relerr(dW, dy[nz].double().cpu().t() @ x1[nz].double().cpu()) < 3e-5
    assert relerr(db, dy[nz].double().cpu().sum(0)) < 1e-5
    # dgrad: 64 sampled rows
    dx = torch.empty(M, K, device=D); dyf = torch.randn(M, N, generator=g).to(D)
    ops.linear_dgrad(dyf, W, dx)
    assert relerr(dx[rows], dyf[rows].double().cpu() @ W.double().cpu()) < 3e-5


@pytest.mark.parametrize("Tq,Tk,causal", [(TM, TM, True), (TM, TM, False), (TM, TT, False), (TT, TM, False),
                                          (2000, 2000, True), (2000, 2000, False), (2000, 300, False), (300, 2000, False)])
def test_attention_full_size_properties(Tq, Tk, causal):
    """The four (Tq,Tk) shapes of config 3 and of config 5 (T_mel=2000, T_text=300) with ragged key lengths: softmax rows sum to one (V = 1 -> O = 1), linearity in V,
    sampled (batch, head) slices against fp64, masked keys receive exactly zero gradient, dO = 0 -> all gradients zero."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(Tq + Tk)
    lens = torch.randint(Tk // 2, Tk + 1, (B,), generator=g); lens[0] = Tk
    li = lens.to(torch.int32).to(D)
    q = (torch.randn(B * Tq, E, generator=g) * 0.7).to(D)
    k = (torch.randn(B * Tk, E, generator=g) * 0.7).to(D)
    v1 = torch.randn(B * Tk, E, generator=g).to(D); v2 = torch.randn(B * Tk, E, generator=g).to(D)
    lse = torch.empty(B, H, Tq, device=D)

    def fwd(v):
        o = torch.empty(B * Tq, E, device=D)
        ops.attn_fwd(q, k, v, o, lse, li, B, H, Tq, Tk, causal)
        return o
    ones = fwd(torch.ones(B * Tk, E, device=D))
    assert float((ones - 1).abs().max()) < 2e-5
    o1, o2, o12 = fwd(v1), fwd(v2), fwd(v1 + v2)
    assert relerr(o12, o1 + o2) < 3e-5
    # sampled slices vs fp64
    for (b, h) in ((0, 0), (B - 1, H - 1), (7, 2)):
        L = int(lens[b])
        qq = q.view(B, Tq, H, 64)[b, :, h].double().cpu(); kk = k.view(B, Tk, H, 64)[b, :L, h].double().cpu()
        vv = v1.view(B, Tk, H, 64)[b, :L, h].double().cpu()
        s = qq @ kk.t() / 8.0
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(Tq, L, dtype=torch.bool), 1), float("-inf"))
        ref = torch.softmax(s, -1) @ vv
        assert relerr(o1.view(B, Tq, H, 64)[b, :, h], ref) < 5e-5
        assert relerr(lse[b, h], torch.logsumexp(s, -1)) < 2e-5
    # backward
    dO = torch.randn(B * Tq, E, generator=g).to(D)
    ws = torch.empty(B, H, Tq, device=D)
    dq = torch.empty(B * Tq, E, device=D); dk = torch.empty(B * Tk, E, device=D); dv = torch.empty(B * Tk, E, device=D)
    fwd(v1)
    o = fwd(v1)
    ops.attn_bwd(q, k, v1, o, dO, lse, ws, dq, dk, dv, li, B, H, Tq, Tk, causal)
    key_pos = torch.arange(Tk)[None, :].expand(B, Tk)
    pad = (key_pos >= lens[:, None]).reshape(B * Tk).to(D)
    assert float(dk[pad].abs().max() if pad.any() else 0) == 0.0 and float(dv[pad].abs().max() if pad.any() else 0) == 0.0
    b, h = 7, 2
    L = int(lens[b])
    qq = q.view(B, Tq, H, 64)[b, :, h].double().cpu().requires_grad_(True)
    kk = k.view(B, Tk, H, 64)[b, :L, h].double().cpu().requires_grad_(True)
    vv = v1.view(B, Tk, H, 64)[b, :L, h].double().cpu().requires_grad_(True)
    s = qq @ kk.t() / 8.0
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Tq, L, dtype=torch.bool), 1), float("-inf"))
    (torch.softmax(s, -1) @ vv * dO.view(B, Tq, H, 64)[b, :, h].double().cpu()).sum().backward()
    assert relerr(dq.view(B, Tq, H, 64)[b, :, h], qq.grad) < 1e-4
    assert relerr(dk.view(B, Tk, H, 64)[b, :L, h], kk.grad) < 1e-4
    assert relerr(dv.view(B, Tk, H, 64)[b, :L, h], vv.grad) < 1e-4
    ops.attn_bwd(q, k, v1, o, torch.zeros_like(dO), lse, ws, dq, dk, dv, li, B, H, Tq, Tk, causal)
    assert float(dq.abs().max()) == 0.0 and float(dk.abs().max()) == 0.0 and float(dv.abs().max()) == 0.0


@pytest.mark.parametrize("T", [TM, 2000])
def test_lstm_full_size_properties(T):
    """Discriminator batch of config 3 (64 sequences x 800 steps, ragged) and of config 5 (x 2000 steps): padded steps stay zero, the reverse direction equals
    the forward direction run on the time-reversed sequence, and a sampled sequence matches torch's fp64 LSTM."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(11)
    Bd, Hh = 2 * B, 64
    lens = torch.randint(T // 4, T + 1, (Bd,), generator=g); lens[0] = T; lens[1] = 1
    li = lens.to(torch.int32).to(D)
    lstm = torch.nn.LSTM(128, Hh, num_layers=1, bidirectional=True, batch_first=True).double()
    with torch.no_grad():
        lstm.weight_ih_l0_reverse.copy_(lstm.weight_ih_l0); lstm.weight_hh_l0_reverse.copy_(lstm.weight_hh_l0)
        lstm.bias_ih_l0_reverse.copy_(lstm.bias_ih_l0); lstm.bias_hh_l0_reverse.copy_(lstm.bias_hh_l0)
    x = torch.randn(Bd, T, 128, generator=g, dtype=torch.float64) * 0.5
    for i in range(Bd):
        x[i, lens[i]:] = 0
    wih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]).detach()
    whh = torch.cat([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]).detach().float().to(D).contiguous()
    bih = torch.cat([lstm.bias_ih_l0, lstm.bias_ih_l0_reverse]).detach().float().to(D)
    bhh = torch.cat([lstm.bias_hh_l0, lstm.bias_hh_l0_reverse]).detach().float().to(D)

    def run(xin):
        xproj = (xin @ wih.t()).float().to(D).contiguous()
        y = torch.zeros(Bd, T, 2 * Hh, device=D); gates = torch.empty(Bd, T, 2, 4 * Hh, device=D); cs = torch.empty(Bd, T, 2, Hh, device=D)
        hprev = torch.zeros(Bd, T, 2, Hh, device=D); hfin = torch.empty(Bd, 2 * Hh, device=D)
        ops.lstm_fwd(xproj, whh, bih, bhh, li, y, gates, cs, hprev, hfin, 2, 4 * Hh * Hh, 4 * Hh)
        return y, hfin
    y, hfin = run(x)
    for i in (0, 1, 5, Bd - 1):
        assert float(y[i, lens[i]:].abs().max() if lens[i] < T else 0) == 0.0
    # time reversal (within each sequence's valid length): with identical weights in both directions, fwd(x_rev) == rev(bwd(x))
    xr = torch.zeros_like(x)
    for i in range(Bd):
        xr[i, :lens[i]] = x[i, :lens[i]].flip(0)
    yr, hfr = run(xr)
    for i in (0, 1, 9, Bd - 1):
        n = int(lens[i])
        assert relerr(yr[i, :n, :Hh], y[i, :n, Hh:].flip(0)) < 2e-5
    assert relerr(hfr[:, :Hh], hfin[:, Hh:]) < 2e-5
    # one sequence against torch fp64
    i = 9
    n = int(lens[i])
    out, (hn, cn) = lstm(x[i:i + 1, :n])
    assert relerr(y[i, :n], out[0]) < 3e-5


def test_layernorm_full_size_properties():
    from unast_amd import ops
    g = torch.Generator().manual_seed(2)
    N = B * TM
    z = (torch.randn(N, E, generator=g) * 3 + 1).to(D)
    gamma = torch.ones(E, device=D); beta = torch.zeros(E, device=D)
    y = torch.empty_like(z); mean = torch.empty(N, device=D); rstd = torch.empty(N, device=D)
    ops.layernorm_fwd(z, gamma, beta, y, mean, rstd)
    assert float(y.mean(1).abs().max()) < 1e-5 and float((y.var(1, unbiased=False) - 1).abs().max()) < 1e-3
    dy = torch.randn(N, E, generator=g).to(D)
    dz = torch.empty_like(z); dg = torch.zeros(E, device=D); db = torch.zeros(E, device=D)
    ops.layernorm_bwd(dy, z, gamma, mean, rstd, dz, dgamma=dg, dbeta=db)
    # the gradient of LayerNorm is orthogonal to the constant vector and to the normalised input
    assert float(dz.sum(1).abs().max()) < 2e-3 and float((dz * y).sum(1).abs().max()) < 2e-2
    assert relerr(db, dy.double().cpu().sum(0)) < 1e-5 and relerr(dg, (dy * y).double().cpu().sum(0)) < 1e-4


def test_full_length_step_vs_oracle_b2():
    """Full sequence lengths of config 3 (T_text=180, T_mel=800, L=4) with B=2 so the CPU oracle finishes in seconds: AE and SP
    sub-step losses and gradients against the pinned oracle."""
    from collections import defaultdict
    from oracle import unast_ref as R
    from unast_amd import train
    from unast_amd.portable import synth_batch
    L = 4
    args, model, opt, sd = build(L, 0.0)
    batch = tuple(torch.from_numpy(x) for x in synth_batch(2, TT, TM, seed=5, ragged=True))
    m = R.Model({k: v.clone() for k, v in sd.items()}, L)
    m.packed_lstm = True                 # torch's own packed-sequence LSTM for the recurrence (same values, seconds instead of minutes)
    for n, p in m.P.items():
        if n.startswith("discriminator."):
            p.requires_grad_(False)
    torch.set_num_threads(16)
    ae = R.generator_losses(m, batch)
    ae.pop("_ae_out")
    (sum(ae.values()) / 2).backward()
    sp = R.supervised_losses(m, batch)
    (sum(sp.values()) / 2).backward()
    losses = defaultdict(list)
    model.train()
    train.freeze_model_parameters(model.discriminator)
    train.train_ae_step(losses, model, batch, 0, 2, args)
    train.train_sp_step(losses, model, batch, 0, 2, args)
    for k, v in list(ae.items()) + list(sp.items()):
        assert abs(float(losses[k][0]) - v.item()) < 3e-4 * max(1.0, abs(v.item())), (k, float(losses[k][0]), v.item())
    model.expose_grads()
    tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.P.values() if p.grad is not None)))
    errs = []
    for n, p in model.named_parameters():
        r = m.P[n].grad
        if r is None or r.double().norm().item() < 1e-5 * tot:
            continue
        nrel = (p.grad.cpu().double() - r.double()).norm().item() / r.double().norm().item()
        errs.append(nrel)
        hot = n.startswith("text_m.prenet.") or n.startswith("text_m.encoder.transformer_encoder.layers.0.self_attn.in_proj")
        assert nrel < (3e-2 if hot else 1e-2), (n, nrel)
    assert np.median(errs) < 2e-3, np.median(errs)


def test_full_batch_step_is_reproducible_and_finite():
    """The complete config-3 step (B=32) twice from the same state and seed: identical losses (to accumulation-order noise),
    finite parameters, gradients of padded-only tokens untouched."""
    from collections import defaultdict
    from unast_amd import train, utils
    from unast_amd.portable import synth_batch
    res = []
    for rep in range(2):
        args, model, opt, sd = build(4, 1e-3)
        utils.set_deterministic(False)
        try:
            utils.set_seed(99)
            batch = tuple(torch.from_numpy(x) for x in synth_batch(B, TT, TM, seed=1, ragged=True))
            batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
            losses = defaultdict(list)
            train.train_step(losses, model, opt, None, batches, 1, args)
            res.append({k: float(v[-1]) for k, v in losses.items()})
            for n, p in model.named_parameters():
                assert torch.isfinite(p).all(), n
        finally:
            utils.set_deterministic(True)
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) <= 2e-5 * max(1.0, abs(res[0][k])), (k, res[0][k], res[1][k])


def test_config5_full_batch_step_is_reproducible_and_finite():
    """BASELINE.json config 5 as a step, not only per kernel: B=32, T_text=300, T_mel=2000 (64 000 rows per speech GEMM, 9 600 per
    text GEMM, 2000x2000 causal attention, 2000x300 / 300x2000 cross-attention, 2000-step recurrences), the complete step twice
    from the same state and seed."""
    from collections import defaultdict
    from unast_amd import train, utils
    from unast_amd.portable import synth_batch
    res = []
    for rep in range(2):
        args, model, opt, sd = build(4, 1e-3)
        utils.set_deterministic(False)
        try:
            utils.set_seed(5)
            batch = tuple(torch.from_numpy(x) for x in synth_batch(B, 300, 2000, seed=2, ragged=True))
            batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
            losses = defaultdict(list)
            train.train_step(losses, model, opt, None, batches, 1, args)
            res.append({k: float(v[-1]) for k, v in losses.items()})
            assert all(v == v and abs(v) < 1e4 for v in res[-1].values()), res[-1]
            assert bool(torch.isfinite(model._store().flat).all())
        finally:
            utils.set_deterministic(True)
        del model, opt
        torch.cuda.empty_cache()
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) <= 2e-5 * max(1.0, abs(res[0][k])), (k, res[0][k], res[1][k])


def test_deferred_discriminator_phase_matches_joined():
    """train_step(defer_d_phase=True) (the D phase stays on its own stream and overlaps the next step's generator forward, as
    train() and bench.py run it) reaches the same losses as the default, fully joined, form over several consecutive steps.
    The learning rate is small enough (2e-4) that Adam does not amplify accumulation-order noise within four steps, yet one
    missed or stale update would move the next step's losses by ~1e-2 relative (checked below by skipping one on purpose)."""
    from collections import defaultdict
    from unast_amd import train
    from unast_amd.engine import join_streams
    from unast_amd.portable import synth_batch
    res = []
    for mode in ("joined", "deferred", "stale"):
        args, model, opt, sd = build(2, 2e-4)
        batch = tuple(torch.from_numpy(x) for x in synth_batch(8, 60, 256, seed=2, ragged=True))
        batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
        losses = defaultdict(list)
        for i in range(4):
            if mode == "stale" and i == 2:      # what an ordering bug would look like: the next step reads D without this step's update
                from unast_amd import ops
                st = model._store()
                saved = st.flat.clone()
                train.train_step(losses, model, opt, None, batches, i + 1, args, defer_d_phase=False)
                a, b = st.regions["disc"]
                st.flat[a:b].copy_(saved[a:b])
                ops.split_f32(st.flat, st.flat_split)
                continue
            train.train_step(losses, model, opt, None, batches, i + 1, args, defer_d_phase=(mode == "deferred"))
        join_streams()
        torch.cuda.synchronize()
        res.append({k: [float(x) for x in v] for k, v in losses.items()})
    gap = 0.0
    for k in res[0]:
        for i, (a, b) in enumerate(zip(res[0][k], res[1][k])):
            gap = max(gap, abs(a - b) / max(1.0, abs(a)))
            assert abs(a - b) <= 1e-3 * max(1.0, abs(a)), (k, i, a, b)
    # that tolerance is meaningful: the deliberately stale run differs by clearly more in the D-dependent losses of the step after
    stale_gap = max(abs(a - b) / max(1.0, abs(a)) for k in ("d", "d_ae", "sp_d") for a, b in zip(res[0][k][3:], res[2][k][3:]))
    assert stale_gap > 3 * max(gap, 1e-4), (stale_gap, gap)


def test_replayed_soak_over_two_ragged_shapes_stays_finite_and_learns():
    """The run that caught the loss-workspace race of round 3 (NaN parameters at B=32 / T_text=300 / T_mel=2000 while
    every other test was green), inside the suite: replayed steps (capture at the second meeting of a shape, csrc/graph_exec.cpp
    afterwards) alternating between config 5's shape and a B=16 / 180 / 800 one, ragged lengths, random sites on.  Parameters
    and losses stay finite and the speech auto-encoder loss falls (reference step: /root/reference/src/train.py:199-259, 358-363)."""
    from collections import defaultdict
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.graphed import GraphedTrainStep
    from unast_amd.portable import synth_batch
    train.DEVICE = D
    args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, warmup_steps=200)      # lr reaches 2e-4 by step 10
    utils.set_seed(0)
    utils.set_deterministic(False)
    try:
        _, _, model, opt, sched = train.initialize_model(args)
        stepper = GraphedTrainStep(model, opt, sched, args)
        losses = defaultdict(list)
        shapes = ((32, 300, 2000), (16, 180, 800))
        data = {s: [tuple(torch.from_numpy(x).to(D) for x in synth_batch(*s, seed=k, ragged=True)) for k in range(2)] for s in shapes}
        for i in range(14):
            s = shapes[(i // 3) % 2]             # three steps of one shape, then three of the other: eager, captured, replayed ... and back
            batch = data[s][i % 2]
            stepper(losses, dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]), i)
        stepper.flush(losses)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(model._store().flat).all()), "non-finite parameters after the soak"
        for k, v in losses.items():
            vals = [float(x) for x in v]
            assert all(x == x and abs(x) < 1e4 for x in vals), (k, vals)
        s_ae = [float(x) for x in losses["s_ae"]]
        assert len(s_ae) == 14
        assert s_ae[12] < s_ae[0] and s_ae[10] < s_ae[4], s_ae          # the same batch of either shape, 12 / 6 updates later
        assert stepper.stats["replays"] >= 4, stepper.stats
    finally:
        utils.set_deterministic(True)


def test_full_size_replayed_steps_equal_eager_steps_to_the_bit():
    """Eight steps alternating between config 5's shape (B=32 / T_text=300 / T_mel=2000) and B=16 / 180 / 800, ragged lengths, once through
    train.train_step and once through GraphedTrainStep (eager, captured, stream-replayed), with every fp32 sum in a fixed order
    (utils.set_deterministic(True, fixed_sums=True)): the seven losses of every step and the parameters agree to the BIT.  The race of
    DESIGN 5c-8b opened its window only at this size; with a fixed summation order a dependency that holds by timing alone cannot hide
    under a noise bound here either."""
    from collections import defaultdict
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.engine import join_streams
    from unast_amd.graphed import GraphedTrainStep
    from unast_amd.portable import synth_batch
    train.DEVICE = D
    shapes = ((32, 300, 2000), (16, 180, 800))
    data = {s: [tuple(torch.from_numpy(x).to(D) for x in synth_batch(*s, seed=k, ragged=True)) for k in range(2)] for s in shapes}
    utils.set_deterministic(True, fixed_sums=True)
    try:
        res = []
        for graphed in (False, True):
            args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, warmup_steps=200)
            utils.set_seed(0)
            _, _, model, opt, sched = train.initialize_model(args)
            stepper = GraphedTrainStep(model, opt, sched, args) if graphed else None
            losses = defaultdict(list)
            for i in range(8):
                batch = data[shapes[(i // 3) % 2]][i % 2]
                b = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
                if graphed:
                    stepper(losses, b, i)
                else:
                    train.train_step(losses, model, opt, sched, b, i, args, defer_d_phase=True)
            if graphed:
                stepper.flush(losses)
                assert stepper.stats["replays"] >= 1, stepper.stats
            join_streams()
            torch.cuda.synchronize()
            res.append(({k: [float(x) for x in v] for k, v in losses.items()}, model._store().flat.detach().clone()))
        (la, pa), (lb, pb) = res
        assert bool(torch.isfinite(pa).all())
        assert la == lb, {k: (la[k], lb[k]) for k in la if la[k] != lb[k]}
        assert torch.equal(pa, pb), float((pa - pb).abs().max())
    finally:
        utils.set_deterministic(True, fixed_sums=False)

