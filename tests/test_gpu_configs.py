"""GPU parity for the BASELINE.json configurations that had no -m gpu coverage in round 1:
  config 2 (B=8, T_text=128, T_mel=512, num_layers=3, use_discriminator=False -- the generator-only branch,
            /root/reference/src/train.py:365-416 `else` arms, :631-638 skipped),
  config 5 (B=32, T_text=300, T_mel=2000 -- long-form: attention / causal conv postnet / BatchNorm / LSTM at those lengths),
plus the MLP `Discriminator` (/root/reference/src/network.py:154-170) and the conv row-gather edge case fixed in b7609a0."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
REL = 1e-3          # north_star tolerance on mel / logits


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def nrel(a, b):
    """Norm-relative error: robust to an isolated LeakyReLU / ReLU gate whose pre-activation is within rounding of zero."""
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def build(L, lr, use_discriminator=True):
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, use_discriminator=use_discriminator)
    train.DEVICE = D
    utils.set_seed(0)
    utils.set_deterministic(True)
    _, _, model, opt, _ = train.initialize_model(args)
    spec = state_dict_spec(L, use_discriminator=use_discriminator)
    assert list(model.state_dict().keys()) == list(spec.keys())
    sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in spec.items()}
    model.load_state_dict(sd)
    opt.param_groups[0]["lr"] = lr
    return args, model, opt, sd


# ---------------------------------------------------------------------------------------------------------------
# config 2: three layers, no discriminator
# ---------------------------------------------------------------------------------------------------------------
def test_config2_generator_only_step_matches_reference_golden(golden_dir, panel_rows):
    """num_layers=3, use_discriminator=False against the fixture the reference itself produced (tools/gen_golden.py
    step_b3_t20_m56_l3_nodisc): outputs within 1e-3, the four losses, every gradient norm, clip + AdamW deltas."""
    from unast_amd import train
    g = np.load(os.path.join(golden_dir, "step_b3_t20_m56_l3_nodisc.npz"))
    B, Tt, Tm, L, _ = [int(v) for v in g["meta"]]
    assert L == 3
    lr = float(g["lr"])
    args, model, opt, sd = build(L, lr, use_discriminator=False)
    assert model.discriminator is None
    batch = tuple(torch.from_numpy(g[k]) for k in ("text", "mel", "text_len", "mel_len"))
    names = [str(n) for n in g["param_names"]]
    params = dict(model.named_parameters())
    assert list(params.keys()) == names
    model.train()
    (text, mel, tl, ml), _ = train.process_batch(batch)
    bn = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    with torch.no_grad():
        logits, t_enc = model.text_ae(text, tl, ret_enc_hid=True)
        pre, post, stop, s_enc = model.speech_ae(mel, ml, ret_enc_hid=True)
        pre2, post2, stop2, _, _ = model.tts(text, tl, mel, ml, ret_enc_hid=True)
        logits2, _ = model.asr(text, tl, mel, ml, ret_enc_hid=True)
    for got, key in ((logits, "ae_logits"), (t_enc, "ae_t_enc"), (pre, "ae_pre"), (post, "ae_post"), (stop, "ae_stop"),
                     (s_enc, "ae_s_enc"), (pre2, "tts_pre"), (post2, "tts_post"), (stop2, "tts_stop"), (logits2, "asr_logits")):
        assert tuple(got.shape) == g[key].shape, key
        assert rel(got, g[key]) < REL, (key, rel(got, g[key]))
    safe = g["ae_logit_margin"] > 10 * REL * np.abs(g["ae_logits"]).max()
    assert np.array_equal(logits.argmax(-1).cpu().numpy()[safe], g["ae_logits"].argmax(-1)[safe])
    model.load_state_dict(bn, strict=False)          # undo the BN running-stat updates of the forward-only pass
    losses = defaultdict(list)
    train.train_ae_step(losses, model, batch, 0, 2, args)
    train.train_sp_step(losses, model, batch, 0, 2, args)
    assert sorted(losses.keys()) == ["asr_", "s_ae", "t_ae", "tts_"]
    for k in losses:
        assert abs(float(losses[k][0]) - g["loss/" + k]) < 2e-4 * max(1.0, abs(g["loss/" + k])), (k, float(losses[k][0]), g["loss/" + k])
    model.expose_grads()
    gn = np.array([params[n].grad.double().norm().item() if params[n].grad is not None else -1.0 for n in names])
    ref = g["gen_grad_norms"]
    assert np.array_equal(gn < 0, ref < 0)
    tot = float(g["gen_grad_norm"])

    def gtol(n):    # same carve-out as tests/test_gpu_parity.py (x16-scaled text prenet -> first text self-attention), not wider
        hot = n.startswith("text_m.prenet.") or n.startswith("text_m.encoder.transformer_encoder.layers.0.self_attn.in_proj")
        return 1e-2 if hot else 1e-3
    bad = [(n, a, b) for n, a, b in zip(names, gn, ref) if b >= 0 and abs(a - b) > gtol(n) * b + 2e-5 * tot]
    assert not bad, bad[:8]
    for key in g.files:
        if key.startswith("gen_grad/"):
            n = key[len("gen_grad/"):]
            assert np.abs(params[n].grad.cpu().numpy() - g[key]).max() < 2 * gtol(n) * np.abs(g[key]).max() + 2e-6 * tot, key
    before = {n: p.detach().clone() for n, p in params.items()}
    train.optimizer_step(model, opt, args)
    assert abs(opt.grad_norm() - tot) < 3e-3 * tot
    dd = np.array([(params[n].detach() - before[n]).double().norm().item() for n in names])
    well = g["gen_grad_norms"] > 1e-3 * tot
    assert np.allclose(dd[well], g["gen_delta_norms"][well], rtol=3e-2, atol=1e-7)
    sdn = model.state_dict()
    for key in g.files:
        if key.startswith("bn/"):
            assert np.abs(sdn[key[3:]].cpu().numpy() - g[key]).max() < 2e-4 * np.abs(g[key]).max() + 1.0 * lr, key


def test_config2_full_length_generator_only_vs_oracle_b2(panel_rows):
    """Config 2's sequence lengths (T_text=128, T_mel=512) and depth (L=3) with B=2, generator-only: the train_step surface
    (AE + SP + clip/AdamW, no D phase) against the pinned oracle."""
    from oracle import unast_ref as R
    from unast_amd import train
    from unast_amd.portable import synth_batch
    L = 3
    args, model, opt, sd = build(L, 1e-3, use_discriminator=False)
    batch = tuple(torch.from_numpy(x) for x in synth_batch(2, 128, 512, seed=8, ragged=True))
    m = R.Model({k: v.clone() for k, v in sd.items()}, L)
    ropt = R.AdamW(m.P, lr=1e-3, weight_decay=args.weight_decay)
    grads = {}
    orig = ropt.step

    def spy(clip):
        grads.update({n: p.grad.clone() for n, p in m.P.items() if p.grad is not None})
        return orig(clip)
    ropt.step = spy
    torch.set_num_threads(16)
    before_ref = {n: p.detach().clone() for n, p in m.P.items()}
    rec = R.full_step(m, ropt, batch, use_discriminator=False)
    losses = defaultdict(list)
    model.train()
    store = model._store()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    # the step surface, with the gradients looked at before the optimizer consumes them
    train.train_ae_step(losses, model, batch, 0, 2, args)
    train.train_sp_step(losses, model, batch, 0, 2, args)
    model.expose_grads()
    tot = float(torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads.values())))
    errs = []
    for n, p in model.named_parameters():
        r = grads.get(n)
        if r is None:
            assert p.grad is None, n
            continue
        if r.double().norm().item() < 1e-5 * tot:
            continue
        nrel = (p.grad.cpu().double() - r.double()).norm().item() / r.double().norm().item()
        errs.append(nrel)
        hot = n.startswith("text_m.prenet.") or n.startswith("text_m.encoder.transformer_encoder.layers.0.self_attn.in_proj")
        assert nrel < (3e-2 if hot else 1e-2), (n, nrel)
    assert np.median(errs) < 2e-3, np.median(errs)
    train.optimizer_step(model, opt, args)
    assert store.touched == set()
    for k in ("t_ae", "s_ae", "asr_", "tts_"):
        assert abs(float(losses[k][0]) - rec[k]) < 3e-4 * max(1.0, abs(rec[k])), (k, float(losses[k][0]), rec[k])
    assert "d" not in losses and "d_ae" not in losses and "sp_d" not in losses
    assert abs(opt.grad_norm() - rec["gen_grad_norm"]) < 5e-3 * rec["gen_grad_norm"]
    # AdamW moved every parameter that has a well-conditioned gradient by the same amount as the oracle's
    for n, p in model.named_parameters():
        r = grads.get(n)
        if r is None or r.double().norm().item() < 1e-3 * tot:
            continue
        d_hip = (p.detach().cpu().double() - before[n].cpu().double()).norm().item()
        d_ref = (m.P[n].detach().double() - before_ref[n].double()).norm().item()
        assert abs(d_hip - d_ref) < 5e-2 * d_ref + 1e-9, (n, d_hip, d_ref)


# ---------------------------------------------------------------------------------------------------------------
# config 5: T_mel = 2000, T_text = 300
# ---------------------------------------------------------------------------------------------------------------
def test_config5_causal_conv_and_batchnorm_properties():
    """SpeechPostnet stage at config-5 size (B=32, T=2000): the causal k=5 convolution never looks ahead (src/module.py:155-168:
    pad 4, drop the last 4), sampled rows match fp64, and train-mode BatchNorm over all B*T = 64000 positions normalises every
    channel and updates the running statistics with the unbiased variance."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, C = 32, 2000, 256
    x = torch.randn(B, T, C, generator=g)
    W = torch.randn(C, C, 5, generator=g) * 0.03
    b = torch.randn(C, generator=g)
    Wp = W.permute(0, 2, 1).contiguous().to(D)
    xd = x.to(D)
    y = torch.empty(B, T, C, device=D)
    ops.conv_fwd(xd, Wp, b.to(D), y, 4)
    # sampled (sequence, time) positions against fp64, including t < 4 where the window hangs over the left edge
    for (bi, t) in ((0, 0), (0, 3), (5, 4), (17, 999), (31, 1999), (12, 1280)):
        acc = b.double().clone()
        for j in range(5):
            ts = t + j - 4
            if ts >= 0:
                acc += W[:, :, j].double() @ x[bi, ts].double()
        assert rel(y[bi, t], acc) < 3e-5, (bi, t)
    # causality: changing inputs at t >= 1000 leaves outputs at t < 1000 bit-identical
    x2 = xd.clone()
    x2[:, 1000:] += 1.0
    y2 = torch.empty(B, T, C, device=D)
    ops.conv_fwd(x2, Wp, b.to(D), y2, 4)
    assert torch.equal(y[:, :1000], y2[:, :1000]) and not torch.equal(y[:, 1000], y2[:, 1000])
    # BatchNorm (train) over the conv output
    N = B * T
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(D); beta = (0.1 * torch.randn(C, generator=g)).to(D)
    out = torch.empty(N, C, device=D); mean = torch.empty(C, device=D); rstd = torch.empty(C, device=D)
    rm = torch.zeros(C, device=D); rv = torch.ones(C, device=D); ws = torch.empty(2 * C, dtype=torch.float64, device=D)
    ops.bn_fwd(y.view(N, C), gamma, beta, out, mean, rstd, rm, rv, ws, 0)
    yd = y.view(N, C).double()
    assert rel(mean, yd.mean(0)) < 1e-5 and rel(rstd, 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5)) < 1e-5
    assert rel(out, (yd - yd.mean(0)) / torch.sqrt(yd.var(0, unbiased=False) + 1e-5) * gamma.double() + beta.double()) < 2e-5
    assert rel(rm, 0.1 * yd.mean(0)) < 1e-5 and rel(rv, 0.9 + 0.1 * yd.var(0, unbiased=True)) < 1e-5
    # gradient of train-mode BN is orthogonal to constants and to the normalised input, per channel
    dy = torch.randn(N, C, generator=g).to(D)
    dx = torch.empty(N, C, device=D); dg = torch.zeros(C, device=D); db = torch.zeros(C, device=D)
    ops.bn_bwd(dy.clone(), y.view(N, C), mean, rstd, gamma, beta, dx, dg, db, ws, 0)
    xhat = (yd - yd.mean(0)) * rstd.double()
    assert float(dx.double().sum(0).abs().max()) < 5e-2 and float((dx.double() * xhat).sum(0).abs().max()) < 5e-1
    assert rel(db, dy.double().sum(0)) < 1e-5 and rel(dg, (dy.double() * xhat).sum(0)) < 1e-4


def test_config5_full_length_step_vs_oracle_b1():
    """One full-length utterance of config 5 (T_text=300, T_mel=2000, L=4), full gen+disc sub-steps: AE and SP losses and
    gradients plus the D-phase loss against the pinned oracle (B=1 so the CPU side finishes in about a minute)."""
    from oracle import unast_ref as R
    from unast_amd import train
    from unast_amd.portable import synth_batch
    L = 4
    args, model, opt, sd = build(L, 0.0)
    batch = tuple(torch.from_numpy(x) for x in synth_batch(1, 300, 2000, seed=9, ragged=False))
    m = R.Model({k: v.clone() for k, v in sd.items()}, L)
    m.packed_lstm = True
    for n, p in m.P.items():
        if n.startswith("discriminator."):
            p.requires_grad_(False)
    torch.set_num_threads(16)
    ae = R.generator_losses(m, batch)
    ref_out = ae.pop("_ae_out")
    (sum(ae.values()) / 2).backward()
    sp = R.supervised_losses(m, batch)
    (sum(sp.values()) / 2).backward()
    model.train()
    # forward outputs at this length first (mel / logits / stop within 1e-3)
    (text, mel, tl, ml), _ = train.process_batch(batch)
    bn = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    with torch.no_grad():
        logits, _ = model.text_ae(text, tl, ret_enc_hid=True)
        pre, post, stop, _ = model.speech_ae(mel, ml, ret_enc_hid=True)
    for got, want, key in ((logits, ref_out[0], "logits"), (pre, ref_out[1], "pre"), (post, ref_out[2], "post"), (stop, ref_out[3], "stop")):
        assert rel(got, want.detach()) < REL, (key, rel(got, want.detach()))
    model.load_state_dict(bn, strict=False)
    losses = defaultdict(list)
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
    train.optimizer_step(model, opt, args)                 # lr = 0: parameters stay, the gradient range is cleared
    train.unfreeze_model_parameters(model.discriminator)
    train.train_discriminator_step(losses, model, batch, 0, 1, args)
    for p in m.P.values():
        p.grad = None
    for n, p in m.P.items():
        p.requires_grad_(n.startswith("discriminator."))
    d = R.discriminator_loss_step(m, batch)
    assert abs(float(losses["d"][0]) - d.item()) < 3e-4 * max(1.0, abs(d.item()))


# ---------------------------------------------------------------------------------------------------------------
# MLP Discriminator (row D3)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(6, 37, 256), (64, 256)])
def test_mlp_discriminator_fwd_bwd_vs_fp64(shape):
    """Discriminator (src/network.py:154-170): 3 x {Linear -> LeakyReLU(0.2) -> Dropout} -> Linear -> squeeze(-1); forward,
    input gradient and all eight parameter gradients against fp64 torch with dropout off, then the dropout statistics."""
    from unast_amd import utils
    from unast_amd.network import Discriminator
    utils.set_seed(3)
    utils.set_deterministic(True)
    g = torch.Generator().manual_seed(11)
    disc = Discriminator(shape[-1], hidden=1024).to(D)
    ref = torch.nn.Sequential()
    lins = []
    for i, name in enumerate(("fc1", "fc2", "fc3", "fc4")):
        src = getattr(disc, name)
        lin = torch.nn.Linear(src.in_features, src.out_features).double()
        with torch.no_grad():
            w = torch.randn(src.weight.shape, generator=g) * (1.5 / src.in_features ** 0.5)
            b = torch.randn(src.bias.shape, generator=g) * 0.1
            src.weight.copy_(w); src.bias.copy_(b)
            lin.weight.copy_(w.double()); lin.bias.copy_(b.double())
        lins.append(lin)
    x = torch.randn(*shape, generator=g)
    xr = x.double().requires_grad_(True)
    h = xr
    for lin in lins[:3]:
        h = torch.nn.functional.leaky_relu(lin(h), 0.2)
    yr = lins[3](h).squeeze(-1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    disc.eval()                                     # dropout off (model.eval()), as the fp64 chain above
    xd = x.to(D).requires_grad_(True)
    y = disc(xd)
    assert tuple(y.shape) == tuple(yr.shape)
    assert rel(y, yr.detach()) < 5e-5
    y.backward(dy.to(D))
    disc._store().expose_grads()
    # A pre-activation within rounding of zero (about two of the 3 x 1024 units of a few hundred rows) takes the other
    # LeakyReLU slope than in fp64: that row's gradient then differs by ~1/sqrt(1024) of its norm.  Rows are independent, so
    # the MEDIAN row error is the kernel's arithmetic error (1e-4 bar) and the whole-tensor norm error gets a looser bar.
    gx, gr = xd.grad.reshape(-1, shape[-1]).double().cpu(), xr.grad.reshape(-1, shape[-1])
    row_err = (gx - gr).norm(dim=1) / gr.norm(dim=1).clamp_min(1e-30)
    assert float(row_err.median()) < 1e-4 and float((row_err > 1e-3).float().mean()) < 0.1, (float(row_err.median()), float(row_err.max()))
    assert nrel(xd.grad, xr.grad) < 5e-3
    for name, lin in zip(("fc1", "fc2", "fc3", "fc4"), lins):
        src = getattr(disc, name)
        assert nrel(src.weight.grad, lin.weight.grad) < 5e-3, name
        assert nrel(src.bias.grad, lin.bias.grad) < 5e-3, name
    disc._store().zero_grad()
    # train mode: dropout p = 0.2 after each activation; the forward/backward masks agree (backward regenerates them)
    utils.set_deterministic(False)
    try:
        disc.train()
        xs = torch.ones(4096, shape[-1], device=D, requires_grad=True)
        with torch.no_grad():
            for name in ("fc1", "fc2", "fc3"):
                getattr(disc, name).weight.fill_(0.0); getattr(disc, name).bias.fill_(1.0)
            disc.fc4.weight.fill_(1.0 / 1024); disc.fc4.bias.fill_(0.0)
        ys = disc(xs)                                # every hidden unit is 1 before dropout: y = mean of the kept/0.8 units of layer 3
        assert abs(float(ys.mean()) - 1.0) < 0.01 and float(ys.std()) > 1e-3
        ys.sum().backward()
        disc._store().expose_grads()
        # d/d(fc3.bias[j]) = sum_rows keep3[row, j] / 0.8 / 1024: the mean over j recovers the keep rate
        assert abs(float(disc.fc3.bias.grad.mean()) * 1024 / 4096 - 1.0) < 0.01
    finally:
        utils.set_deterministic(True)


def test_mlp_discriminator_in_the_adversarial_loss():
    """The 1-argument MLP discriminator behind the reference's loss (BCE-with-logits on smoothed targets, src/train.py:147-164):
    per-token logits of an encoder output, gradient back to the encoder output, against fp64."""
    from unast_amd import train, utils
    from unast_amd.network import Discriminator
    utils.set_seed(4)
    utils.set_deterministic(True)
    g = torch.Generator().manual_seed(2)
    disc = Discriminator(256, hidden=1024).to(D)
    disc.eval()
    enc = torch.randn(3, 21, 256, generator=g)
    tgt = torch.full((3, 21), 0.9)
    e = enc.to(D).requires_grad_(True)
    out = disc(e)
    loss = train.discriminator_loss(out.reshape(-1), tgt.reshape(-1).to(D))
    loss.backward()
    lins = [(getattr(disc, n).weight.detach().double().cpu(), getattr(disc, n).bias.detach().double().cpu()) for n in ("fc1", "fc2", "fc3", "fc4")]
    er = enc.double().requires_grad_(True)
    h = er
    for w, b in lins[:3]:
        h = torch.nn.functional.leaky_relu(h @ w.t() + b, 0.2)
    lr_ = torch.nn.functional.binary_cross_entropy_with_logits((h @ lins[3][0].t() + lins[3][1]).squeeze(-1), tgt.double())
    lr_.backward()
    assert abs(float(loss.detach()) - lr_.item()) < 1e-5 * max(1.0, abs(lr_.item()))
    assert nrel(e.grad, er.grad) < 5e-3


# ---------------------------------------------------------------------------------------------------------------
# conv1d row gather at the end of the activation buffer (the out-of-bounds row base fixed in b7609a0)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T", [(1, 1), (1, 3), (3, 4), (2, 2), (1, 129), (5, 51)])
@pytest.mark.parametrize("pad", [2, 4])
def test_conv_gather_row_base_beyond_last_sequence(B, T, pad):
    """Tiles whose rows run past M = B*T (M < 128 or M % 128 != 0) and sequences shorter than the kernel (T < 5): the row
    base of the time-shifted gather used to be computed from the unclamped tile row, i.e. from a sequence that does not
    exist (GPU memory fault in round 1, gpurun_out/t11.log).  The activation sits at the very end of its allocation and is
    framed by NaN guards, so a read that is not clamped-and-zeroed either faults or poisons the result."""
    from unast_amd import ops
    g = torch.Generator().manual_seed(B * 131 + T)
    Cin = Cout = 256
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    W = torch.randn(Cout, Cin, 5, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True); Wr = W.clone().requires_grad_(True)
    yr = torch.nn.functional.conv1d(torch.nn.functional.pad(xr.transpose(1, 2), (pad, 4 - pad)), Wr, b).transpose(1, 2)
    dy = torch.randn(B, T, Cout, generator=g, dtype=torch.float64)
    yr.backward(dy)
    n = B * T * Cin
    guard = 4096
    buf = torch.full((guard + n,), float("nan"), device=D)          # [NaN guard | x]: x ends where the allocation ends
    xd = buf[guard:].view(B, T, Cin)
    xd.copy_(x.float())
    dbuf = torch.full((guard + B * T * Cout,), float("nan"), device=D)
    dyd = dbuf[guard:].view(B, T, Cout)
    dyd.copy_(dy.float())
    Wp = W.permute(0, 2, 1).contiguous().float().to(D)
    y = torch.empty(B, T, Cout, device=D)
    ops.conv_fwd(xd, Wp, b.float().to(D), y, pad)
    assert torch.isfinite(y).all() and rel(y, yr.detach()) < 3e-5
    dx = torch.empty(B, T, Cin, device=D)
    ops.conv_dgrad(dyd, Wp, dx, pad)
    assert torch.isfinite(dx).all() and rel(dx, xr.grad) < 3e-5
    dWp = torch.zeros(Cout, 5, Cin, device=D); db = torch.zeros(Cout, device=D)
    ops.conv_wgrad(dyd, xd, dWp, pad, db=db)
    assert torch.isfinite(dWp).all() and rel(dWp, Wr.grad.permute(0, 2, 1)) < 3e-5
    assert rel(db, dy.sum((0, 1))) < 1e-5
