"""Autoregressive inference (K/V-cached) and the cross-model back-translation step vs. golden vectors produced by the
reference's own infer_sequence / train_cm_step (tools/gen_golden_cm.py; src/network.py:103-123, 219-252, 455-481,
src/train.py:261-294, 418-444).  Tolerance: north_star's 1e-3 on mel / stop logits; tokens and lengths bit-exact."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
REL = 1e-3
CASES = ["cm_b3_t12_m40_l2_s1234", "cm_b3_t12_m40_l2_s77"]


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def build(g):
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    B, Tt, Tm, L, seed, text_cap, mel_cap = [int(v) for v in g["meta"]]
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=1)
    train.DEVICE = D
    utils.set_seed(0)
    utils.set_deterministic(True)
    _, _, model, opt, _ = train.initialize_model(args)
    spec = state_dict_spec(L)
    model.load_state_dict({k: torch.from_numpy(portable_tensor(k, shp, seed)) for k, shp in spec.items()})
    # the same smaller generation caps the fixture was made with: plain arguments of infer_sequence
    orig_s, orig_t = model.speech_m.infer_sequence, model.text_m.infer_sequence
    model.speech_m.infer_sequence = lambda memory, masks, max_len=mel_cap: orig_s(memory, masks, max_len)
    model.text_m.infer_sequence = lambda memory, masks, max_len=text_cap: orig_t(memory, masks, max_len)
    model.speech_m.infer_max_len, model.text_m.infer_max_len = mel_cap, text_cap        # ... and of the cross-model paths
    model.train()
    return args, model, opt


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    batch = tuple(torch.from_numpy(g[k]) for k in ("text", "mel", "text_len", "mel_len"))
    return g, batch


@pytest.mark.parametrize("name", CASES)
def test_infer_sequence_matches_reference(golden_dir, name):
    from unast_amd import train
    g, batch = load(golden_dir, name)
    args, model, opt = build(g)
    (text, mel, tl, ml), _ = train.process_batch(batch)
    with torch.no_grad():
        t_enc, t_masks = model.text_m.encode(text, tl)
        pre, post, stops, slens = model.speech_m.infer_sequence(t_enc, t_masks)
        s_enc, s_masks = model.speech_m.encode(mel, ml)
        tokens, tlens = model.text_m.infer_sequence(s_enc, s_masks)
    assert slens.cpu().tolist() == g["inf_slens"].tolist()
    assert tlens.cpu().tolist() == g["inf_tlens"].tolist()
    assert tuple(pre.shape) == g["inf_pre"].shape and tuple(post.shape) == g["inf_post"].shape
    assert tuple(stops.shape) == g["inf_stop"].shape and tuple(tokens.shape) == g["inf_text"].shape
    assert tokens.dtype == torch.int64 and slens.dtype == torch.int64
    assert np.array_equal(tokens.cpu().numpy(), g["inf_text"]), "generated tokens differ"
    assert rel(pre, g["inf_pre"]) < REL, rel(pre, g["inf_pre"])
    assert rel(post, g["inf_post"]) < REL, rel(post, g["inf_post"])
    assert rel(stops, g["inf_stop"]) < REL, rel(stops, g["inf_stop"])
    # frames at and after each stop length are exactly zero (src/network.py:249-251)
    for b, n in enumerate(g["inf_slens"].tolist()):
        assert float(post[b, n:].abs().max() if n < post.shape[1] else 0) == 0.0


@pytest.mark.parametrize("name", CASES)
def test_cm_step_matches_reference(golden_dir, name):
    from unast_amd import train
    g, batch = load(golden_dir, name)
    args, model, opt = build(g)
    losses = defaultdict(list)
    train.freeze_model_parameters(model.discriminator)
    train.train_cm_step(losses, model, batch, 0, 3, args)
    for k in ("s_cm", "t_cm", "d_cm"):
        got, want = float(losses[k][0]), float(g["loss/" + k])
        assert abs(got - want) <= 2e-3 * abs(want) + 1e-5, (k, got, want)
    model.expose_grads()
    names = [str(n) for n in g["param_names"]]
    params = dict(model.named_parameters())
    assert list(params.keys()) == names
    gn = g["grad_norms"]
    scale = float(gn[gn > 0].max())
    bad = []
    for (n, p), want in zip(params.items(), gn):
        if want < 0:
            assert p.grad is None, n + " must have no gradient (frozen / unused)"
            continue
        assert p.grad is not None, n
        got = float(p.grad.double().norm())
        if abs(got - want) > 2e-2 * want + 1e-4 * scale:
            bad.append((n, got, float(want)))
    assert not bad, bad[:8]


def test_train_step_with_cm_runs(golden_dir):
    """The hot loop with cm_steps=1 (ae + cm + sp + optimizer + discriminator) runs and produces finite losses."""
    from unast_amd import train
    g, batch = load(golden_dir, CASES[1])
    args, model, opt = build(g)
    batches = {"unsup": [batch], "cm": [batch], "sup": [batch], "disc": [batch]}
    losses = defaultdict(list)
    train.train_step(losses, model, opt, None, batches, 0, args)
    for k in ("s_cm", "t_cm", "d_cm", "s_ae", "t_ae"):
        assert k in losses and np.isfinite(float(losses[k][-1])), k
    for n, p in model.named_parameters():
        assert torch.isfinite(p).all(), n


def test_graph_and_eager_decoding_agree(golden_dir):
    """The captured-graph decode loop and the launch-by-launch loop produce the same tokens / frames / lengths."""
    from unast_amd import train, config
    g, batch = load(golden_dir, CASES[1])
    args, model, opt = build(g)
    (text, mel, tl, ml), _ = train.process_batch(batch)
    outs = []
    for use_graph in (True, False):
        config.DECODE_GRAPH = use_graph
        try:
            with torch.no_grad():
                t_enc, t_masks = model.text_m.encode(text, tl)
                pre, post, stops, slens = model.speech_m.infer_sequence(t_enc, t_masks)
                s_enc, s_masks = model.speech_m.encode(mel, ml)
                tokens, tlens = model.text_m.infer_sequence(s_enc, s_masks)
        finally:
            config.DECODE_GRAPH = True
        outs.append((pre, post, stops, slens, tokens, tlens))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    from unast_amd import ops
    assert int(ops.rng_epoch_counter().item()) == 0, "the RNG epoch counter must be back at 0 after generation"


def test_graph_decoding_draws_fresh_dropout_per_position(golden_dir):
    """With dropout active, replayed launches must not reuse one mask at every position: the device-side epoch counter varies
    the streams.  Two generations with the same seed agree with each other (determinism) and differ from a dropout-free one."""
    from unast_amd import train, utils
    g, batch = load(golden_dir, CASES[1])
    args, model, opt = build(g)
    (text, mel, tl, ml), _ = train.process_batch(batch)
    utils.set_deterministic(False)
    try:
        runs = []
        for _ in range(2):
            utils.set_seed(7)
            with torch.no_grad():
                t_enc, t_masks = model.text_m.encode(text, tl)
                pre, post, stops, slens = model.speech_m.infer_sequence(t_enc, t_masks)
            runs.append(pre.clone())
        assert runs[0].shape[0] == runs[1].shape[0] and torch.isfinite(runs[0]).all()
        if runs[0].shape == runs[1].shape:
            assert torch.equal(runs[0], runs[1]), "same seed => same generation"
        # frames at consecutive positions are not produced with one frozen mask: their prenet-dropout patterns differ
        d = (runs[0][:, 1:] - runs[0][:, :-1]).abs().amax(dim=(0, 2)) if runs[0].shape[1] > 2 else torch.ones(1)
        assert float(d.min()) > 0
    finally:
        utils.set_deterministic(True)


def test_lockstep_generation_matches_one_after_the_other(golden_dir):
    """cm_both_in (the two directions decoded in lock-step, one graph with two branches per position, then the longer one on
    its own) returns exactly what cm_speech_in followed by cm_text_in return; different caps so that one finishes first."""
    from unast_amd import train, ops
    g, batch = load(golden_dir, CASES[0])
    args, model, opt = build(g)
    (text, mel, tl, ml), _ = train.process_batch(batch)
    for caps in ((40, 24), (16, 48)):
        model.speech_m.infer_max_len, model.text_m.infer_max_len = caps
        sp = model.cm_speech_in(mel, ml, ret_enc_hid=True)
        tx = model.cm_text_in(text, tl, ret_enc_hid=True)
        sp2, tx2 = model.cm_both_in(text, tl, mel, ml, ret_enc_hid=True)
        for a, b in zip(list(sp) + list(tx), list(sp2) + list(tx2)):
            assert a.shape == b.shape and torch.equal(a, b)
        assert int(ops.rng_epoch_counter().item()) == 0
