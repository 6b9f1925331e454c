"""Pins the oracle (oracle/unast_ref.py) against golden vectors produced by the reference itself
(tools/gen_golden.py ran /root/reference/src's own step functions)."""
import os

import numpy as np
import pytest
import torch

from oracle import unast_ref as R
from unast_amd.portable import portable_tensor

CASES = ["step_b1_t40_m200_l4", "step_b4_t24_m64_l2", "step_b4_t24_m64_l4_lr0"]


def state_dict_for(L):
    from unast_amd.spec import state_dict_spec
    return {k: portable_tensor(k, shp, 1234) for k, shp in state_dict_spec(L).items()}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_unit_vectors(golden_dir):
    g = load(golden_dir, "unit")
    lens = torch.from_numpy(g["lens"])
    assert np.array_equal(R.lens_mask(lens, 7).numpy(), g["sent_lens_to_mask"])
    T = 6
    causal = (torch.arange(T)[None, :] > torch.arange(T)[:, None]).numpy()
    assert np.array_equal(causal, g["causal_mask"])
    sl = R.speech_loss(torch.from_numpy(g["sl_gold"]), torch.from_numpy(g["sl_gold_stop"]), torch.from_numpy(g["sl_pre"]),
                       torch.from_numpy(g["sl_post"]), torch.from_numpy(g["sl_len"]), torch.from_numpy(g["sl_stop"]), 5.0)
    assert abs(sl.item() - g["speech_loss"]) < 2e-6 * abs(g["speech_loss"])
    for w, key in ((1.0, "text_loss_w1"), (3.0, "text_loss_w3")):
        tl = R.text_loss(torch.from_numpy(g["tl_text"]), torch.from_numpy(g["tl_logits"]), w)
        assert abs(tl.item() - g[key]) < 2e-6 * abs(g[key])
    dl = R.bce_logits_mean(torch.from_numpy(g["dl_out"]), torch.from_numpy(g["dl_tgt"]))
    assert abs(dl.item() - g["disc_loss"]) < 2e-6
    assert np.allclose(g["disc_target_text"], 0.9) and np.allclose(g["disc_target_speech"], 0.1, atol=1e-7)
    from unast_amd.portable import positional_table
    pe = torch.from_numpy(positional_table(5000, 256))[None]
    assert np.array_equal(pe[0, :16].numpy(), g["pe_buf"])
    assert rel(R.pos_enc(torch.from_numpy(g["pe_x"]), pe).numpy(), g["pe_y"]) < 1e-6
    assert np.allclose([0.0625 * R.transformer_schedule(i, 2000) for i in range(5)], g["sched_transformer_first5"], rtol=1e-12)
    assert np.allclose([R.linear_schedule(i, 3, 10) for i in range(11)], g["sched_linear_11"], rtol=1e-12)


@pytest.mark.parametrize("name", CASES)
def test_full_step_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    B, Tt, Tm, L, ragged = [int(v) for v in g["meta"]]
    sd = state_dict_for(L)
    names = [str(n) for n in g["param_names"]]
    assert [k for k in sd if k in set(names)] == names, "state_dict order/keys differ from the reference"
    batch = tuple(torch.from_numpy(g[k]) for k in ("text", "mel", "text_len", "mel_len"))
    torch.manual_seed(0)

    # forward parity (train mode, RNG off): mel / logits / stop / encoder outputs
    m = R.Model(sd, L, requires_grad=False)
    m.update_bn = False
    with torch.no_grad():
        text, mel, tl, ml = batch
        logits, t_enc = m.text_ae(text, tl)
        pre, post, stop, s_enc = m.speech_ae(mel, ml)
        pre2, post2, stop2, _ = m.tts(text, tl, mel, ml)
        logits2, _ = m.asr(text, tl, mel, ml)
    for got, key in ((logits, "ae_logits"), (t_enc, "ae_t_enc"), (pre, "ae_pre"), (post, "ae_post"), (stop, "ae_stop"),
                     (s_enc, "ae_s_enc"), (pre2, "tts_pre"), (post2, "tts_post"), (stop2, "tts_stop"), (logits2, "asr_logits")):
        assert rel(got.numpy(), g[key]) < 2e-5, key
    assert np.array_equal(logits.argmax(-1).numpy()[g["ae_logit_margin"] > 1e-3],
                          g["ae_logits"].argmax(-1)[g["ae_logit_margin"] > 1e-3])

    # full step: losses, gradients, AdamW deltas, BN running stats
    m = R.Model(sd, L)
    opt = R.AdamW(m.P, lr=float(g["lr"]), weight_decay=1e-6)
    grads = {}
    orig_step = opt.step

    def spy(clip):
        grads[len(grads)] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.P.items()}
        return orig_step(clip)
    opt.step = spy
    before = {n: p.detach().clone() for n, p in m.P.items()}
    rec = R.full_step(m, opt, batch)
    # NOTE on post-optimizer quantities ('d' loss, BN running stats, deltas): the first AdamW step moves every
    # parameter by lr*g/(|g|+eps) ~ +-lr, so parameters whose gradient is analytically ZERO (conv biases feeding a
    # train-mode BatchNorm, attention key biases) move by an amount set by fp32 rounding noise in g.  Those moves do
    # not change any model output but do shift BN running means by <= momentum*lr; tolerances below allow for that.
    lr = float(g["lr"])
    for k in ["t_ae", "s_ae", "d_ae", "asr_", "tts_", "sp_d", "d"]:
        tol = 5e-6 if (k != "d" or lr == 0) else 5e-5
        assert abs(rec[k] - g["loss/" + k]) < tol * max(1.0, abs(g["loss/" + k])), (k, rec[k], g["loss/" + k])
    assert abs(rec["gen_grad_norm"] - g["gen_grad_norm"]) < 2e-4 * g["gen_grad_norm"]
    assert abs(rec["d_grad_norm"] - g["d_grad_norm"]) < 2e-4 * g["d_grad_norm"]
    gn = np.array([grads[0][n].double().norm().item() if grads[0][n] is not None else -1.0 for n in names])
    ref = g["gen_grad_norms"]
    assert np.array_equal(gn < 0, ref < 0), "set of parameters without gradient differs"
    assert np.allclose(gn, ref, rtol=2e-3, atol=1e-6 * g["gen_grad_norm"])
    dn = np.array([grads[1][n].double().norm().item() if grads[1][n] is not None else -1.0 for n in names])
    assert np.array_equal(dn < 0, g["d_grad_norms"] < 0)
    assert np.allclose(dn, g["d_grad_norms"], rtol=2e-3, atol=1e-6)
    for key in g.files:
        if key.startswith("gen_grad/"):
            n = key[len("gen_grad/"):]
            # (conv biases feeding a train-mode BatchNorm have an analytically zero gradient: ~1e-8 noise)
            assert np.abs(grads[0][n].numpy() - g[key]).max() < 5e-4 * np.abs(g[key]).max() + 1e-6, key
        if key.startswith("d_grad/"):
            n = key[len("d_grad/"):]
            assert rel(grads[1][n].numpy(), g[key]) < 5e-4, key
        if key.startswith("bn/"):
            assert np.abs(m.buf[key[3:]].numpy() - g[key]).max() < 1e-5 * np.abs(g[key]).max() + 0.2 * lr, key
    psum = sum(p.double().sum().item() for p in m.P.values())
    assert abs(psum - g["param_sum_final"]) < (1e-3 if lr == 0 else 0.5)   # see NOTE above
    if float(g["lr"]) > 0:
        dd = np.array([(m.P[n].detach() - before[n]).double().norm().item() for n in names])
        tot = g["gen_delta_norms"] + g["d_delta_norms"]
        well = (g["gen_grad_norms"] > 1e-4 * g["gen_grad_norm"]) | (g["d_grad_norms"] > 1e-4 * g["d_grad_norm"])
        assert np.allclose(dd[well], tot[well], rtol=1e-2, atol=1e-7)
        assert np.array_equal(dd == 0, tot == 0), "set of untouched parameters differs (reduce_c_W must not move)"


def test_generator_only_step_matches_reference(golden_dir):
    """BASELINE.json config 2's branch (num_layers=3, use_discriminator=False: src/train.py:365-416 `else` arms, no D phase):
    outputs, the four losses, gradients and AdamW deltas of the oracle against the reference-generated fixture."""
    g = load(golden_dir, "step_b3_t20_m56_l3_nodisc")
    B, Tt, Tm, L, ragged = [int(v) for v in g["meta"]]
    assert L == 3 and "loss/d" not in g.files and "loss/d_ae" not in g.files
    from unast_amd.spec import state_dict_spec
    sd = {k: portable_tensor(k, shp, 1234) for k, shp in state_dict_spec(L, use_discriminator=False).items()}
    names = [str(n) for n in g["param_names"]]
    assert [k for k in sd if k in set(names)] == names and not any(n.startswith("discriminator.") for n in names)
    batch = tuple(torch.from_numpy(g[k]) for k in ("text", "mel", "text_len", "mel_len"))
    m = R.Model(sd, L, requires_grad=False)
    m.update_bn = False
    with torch.no_grad():
        text, mel, tl, ml = batch
        logits, t_enc = m.text_ae(text, tl)
        pre, post, stop, s_enc = m.speech_ae(mel, ml)
        pre2, post2, stop2, _ = m.tts(text, tl, mel, ml)
        logits2, _ = m.asr(text, tl, mel, ml)
    for got, key in ((logits, "ae_logits"), (t_enc, "ae_t_enc"), (pre, "ae_pre"), (post, "ae_post"), (stop, "ae_stop"),
                     (s_enc, "ae_s_enc"), (pre2, "tts_pre"), (post2, "tts_post"), (stop2, "tts_stop"), (logits2, "asr_logits")):
        assert rel(got.numpy(), g[key]) < 5e-5, key      # fp32 accumulation-order noise of the x16-scaled text side: 2.4e-5
    m = R.Model(sd, L)
    opt = R.AdamW(m.P, lr=float(g["lr"]), weight_decay=1e-6)
    grads = {}
    orig_step = opt.step

    def spy(clip):
        grads[len(grads)] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.P.items()}
        return orig_step(clip)
    opt.step = spy
    before = {n: p.detach().clone() for n, p in m.P.items()}
    rec = R.full_step(m, opt, batch, use_discriminator=False)
    assert sorted(k for k in rec if not k.endswith("norm")) == ["asr_", "s_ae", "t_ae", "tts_"] and len(grads) == 1
    for k in ["t_ae", "s_ae", "asr_", "tts_"]:
        assert abs(rec[k] - g["loss/" + k]) < 5e-6 * max(1.0, abs(g["loss/" + k])), (k, rec[k], g["loss/" + k])
    assert abs(rec["gen_grad_norm"] - g["gen_grad_norm"]) < 2e-4 * g["gen_grad_norm"]
    gn = np.array([grads[0][n].double().norm().item() if grads[0][n] is not None else -1.0 for n in names])
    assert np.array_equal(gn < 0, g["gen_grad_norms"] < 0)
    assert np.allclose(gn, g["gen_grad_norms"], rtol=2e-3, atol=1e-6 * g["gen_grad_norm"])
    for key in g.files:
        if key.startswith("gen_grad/"):
            assert np.abs(grads[0][key[len("gen_grad/"):]].numpy() - g[key]).max() < 5e-4 * np.abs(g[key]).max() + 1e-6, key
    dd = np.array([(m.P[n].detach() - before[n]).double().norm().item() for n in names])
    well = g["gen_grad_norms"] > 1e-4 * g["gen_grad_norm"]
    assert np.allclose(dd[well], g["gen_delta_norms"][well], rtol=1e-2, atol=1e-7)


def test_packed_lstm_form_equals_time_loop():
    """The oracle's two forms of the LSTM discriminator (explicit loop / torch packed-sequence LSTM) agree, values and grads."""
    torch.manual_seed(1)
    sd = state_dict_for(1)
    x = torch.randn(5, 19, 256)
    lens = torch.tensor([19, 3, 11, 1, 7])
    outs, grads = [], []
    for packed in (False, True):
        m = R.Model(sd, 1)
        m.packed_lstm = packed
        y = m.lstm_discriminator(x, lens)
        (y * torch.arange(1, 6)).sum().backward()
        outs.append(y.detach())
        grads.append({k: p.grad.clone() for k, p in m.P.items() if p.grad is not None})
    assert torch.allclose(outs[0], outs[1], atol=1e-6)
    assert grads[0].keys() == grads[1].keys()
    for k in grads[0]:
        assert torch.allclose(grads[0][k], grads[1][k], atol=2e-6, rtol=1e-4), k
