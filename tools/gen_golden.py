#!/usr/bin/env python3
"""Generate golden input/output vectors for the UNAST train-step hot path.

Runs ONLY in the build container: it imports the reference from /root/reference/src
(stubbing the six absent, hot-path-irrelevant third-party packages),
loads portable seeded weights (tools/portable_init.py, keyed by state_dict name), disables
every RNG site, runs the reference's own step functions
    freeze(D) -> train_ae_step -> train_sp_step -> optimizer_step -> unfreeze(D)
    -> train_discriminator_step -> optimizer_step            (src/train.py:602-638)
and writes inputs + expected outputs as small .npz fixtures under tests/golden/.

No reference source or bytecode is copied: fixtures are data only.
Usage: python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import collections
import collections.abc
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from portable_init import portable_state_dict, synth_batch  # noqa: E402

REF_SRC = "/root/reference/src"


def import_reference():
    sys.dont_write_bytecode = True
    for name in ["librosa", "librosa.effects", "librosa.filters", "jiwer", "eng_to_ipa",
                 "unidecode", "inflect", "soundfile", "tensorboard"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["jiwer"].wer = lambda a, b: 0.0
    sys.modules["unidecode"].unidecode = lambda s: s
    sys.modules["inflect"].engine = lambda: None
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    torch.utils.tensorboard = tb
    collections.Mapping = collections.abc.Mapping
    sys.path.insert(0, REF_SRC)
    import module, network, utils, train  # noqa: F401
    return module, network, utils, train


def make_args(num_layers):
    cfg = json.load(open(os.path.join(REF_SRC, "configs", "transformer_d_trans.json")))
    args = SimpleNamespace(**cfg)
    args.load_path = None
    args.num_layers = num_layers
    return args


def deterministic_mode(model, network, train):
    """switch every RNG site off."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if isinstance(m, torch.nn.LSTM):
            m.dropout = 0.0
    network.noise_fn = lambda x, *a, **k: x
    train.specaugment = lambda mel, mel_len, *a, **k: mel.detach().clone()
    train.torch.randperm = lambda n, *a, **k: torch.arange(n)


def run_case(mods, name, B, Tt, Tm, L, ragged, out_dir, lr_warm_steps, use_discriminator=True):
    """use_discriminator=False is BASELINE.json config 2's branch (src/train.py:922-924 builds no discriminator, the
    `else` arms of train_ae_step / train_sp_step src/train.py:365-416 run, the D phase is skipped src/train.py:631-638)."""
    module, network, utils, train = mods
    args = make_args(L)
    args.use_discriminator = use_discriminator
    train.DEVICE = torch.device("cpu")
    train.WRITER = None
    utils.set_seed(0)
    s_epoch, best, model, opt, sched = train.initialize_model(args)
    sd = portable_state_dict(model.state_dict(), seed=1234)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    deterministic_mode(model, network, train)
    model.train()
    for _ in range(lr_warm_steps):
        sched.step()
    lr = opt.param_groups[0]["lr"]

    text, mel, text_len, mel_len = synth_batch(B, Tt, Tm, seed=0, ragged=ragged)
    batch = (torch.from_numpy(text), torch.from_numpy(mel),
             torch.from_numpy(text_len), torch.from_numpy(mel_len))
    out = {"text": text, "mel": mel, "text_len": text_len, "mel_len": mel_len,
           "meta": np.array([B, Tt, Tm, L, int(ragged)], np.int64), "lr": np.float64(lr)}
    names = [n for n, _ in model.named_parameters()]

    # ---- plain forward outputs (train mode, RNG off) for parity of mel/logits/stop ----
    with torch.no_grad():
        bn_backup = {k: v.clone() for k, v in model.state_dict().items()
                     if "running" in k or "num_batches" in k}
        (t, m, tl, ml), (gc, gm, gs) = train.process_batch(batch)
        logits, t_enc = model.text_ae(t, tl, ret_enc_hid=True)
        pre, post, stop, s_enc = model.speech_ae(m, ml, ret_enc_hid=True)
        pre2, post2, stop2, _, t_enc2 = model.tts(t, tl, m, ml, ret_enc_hid=True)
        logits2, s_enc2 = model.asr(t, tl, m, ml, ret_enc_hid=True)
        out.update(ae_logits=logits.numpy(), ae_t_enc=t_enc.numpy(), ae_pre=pre.numpy(),
                   ae_post=post.numpy(), ae_stop=stop.numpy(), ae_s_enc=s_enc.numpy(),
                   tts_pre=pre2.numpy(), tts_post=post2.numpy(), tts_stop=stop2.numpy(),
                   asr_logits=logits2.numpy(), gold_stop=gs.numpy())
        # argmax margins: bit-exact argmax parity is only required where the top-2 margin is
        # far above the numeric tolerance; the fixture records the margin.
        top2 = torch.topk(logits, 2, dim=-1).values
        out["ae_logit_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
        model.load_state_dict(bn_backup, strict=False)   # undo BN running-stat updates

    # ---- generator phase ----
    losses = collections.defaultdict(list)
    if use_discriminator:
        train.freeze_model_parameters(model.discriminator)
    train.train_ae_step(losses, model, batch, 0, 2, args)
    train.train_sp_step(losses, model, batch, 0, 2, args)
    grads = {n: (p.grad.detach().clone() if p.grad is not None else None)
             for n, p in model.named_parameters()}
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None))
    out["gen_grad_norm"] = np.float64(gnorm.item())
    out["gen_grad_norms"] = np.array([grads[n].double().norm().item() if grads[n] is not None else -1.0
                                      for n in names])
    out["gen_grad_sums"] = np.array([grads[n].double().sum().item() if grads[n] is not None else 0.0
                                     for n in names])
    keep_full = ["text_m.postnet.fc1.bias", "text_m.prenet.batch_norm1.weight", "text_m.prenet.conv1.conv.bias",
                 "speech_m.postnet.stop_linear.weight", "speech_m.postnet.linear_project.bias",
                 "speech_m.postnet.pre_batchnorm.bias", "speech_m.postnet.conv2.conv.bias",
                 "text_m.encoder.transformer_encoder.layers.0.norm1.weight",
                 "speech_m.decoder.transformer_decoder.layers.%d.multihead_attn.in_proj_bias" % (L - 1),
                 "speech_m.encoder.transformer_encoder.layers.0.self_attn.out_proj.bias",
                 "text_m.prenet.embed.weight", "speech_m.prenet.layer.fc1.linear_layer.bias"]
    for n in keep_full:
        out["gen_grad/" + n] = grads[n].numpy()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    train.optimizer_step(model, opt, args)
    after = {n: p.detach().clone() for n, p in model.named_parameters()}
    out["gen_delta_norms"] = np.array([(after[n] - before[n]).double().norm().item() for n in names])
    for n in keep_full:
        out["gen_delta/" + n] = (after[n] - before[n]).numpy()

    # ---- discriminator phase ----
    dnorm = torch.zeros(())
    if use_discriminator:
        train.unfreeze_model_parameters(model.discriminator)
        train.train_discriminator_step(losses, model, batch, 0, 1, args)
        dgrads = {n: (p.grad.detach().clone() if p.grad is not None else None)
                  for n, p in model.named_parameters()}
        dnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in dgrads.values() if g is not None))
        out["d_grad_norm"] = np.float64(dnorm.item())
        out["d_grad_norms"] = np.array([dgrads[n].double().norm().item() if dgrads[n] is not None else -1.0
                                        for n in names])
        for n in ["discriminator.fc2.weight", "discriminator.rnn.reduce_h_W.bias",
                  "discriminator.rnn.rnn.bias_hh_l1_reverse", "discriminator.rnn.rnn.bias_ih_l0"]:
            out["d_grad/" + n] = dgrads[n].numpy()
        before = after
        train.optimizer_step(model, opt, args)
        after = {n: p.detach().clone() for n, p in model.named_parameters()}
        out["d_delta_norms"] = np.array([(after[n] - before[n]).double().norm().item() for n in names])

    for k in (["t_ae", "s_ae", "d_ae", "asr_", "tts_", "sp_d", "d"] if use_discriminator else ["t_ae", "s_ae", "asr_", "tts_"]):
        out["loss/" + k] = np.float64(losses[k][0])
    assert sorted(losses.keys()) == sorted(k[5:] for k in out if k.startswith("loss/"))
    fsd = model.state_dict()
    for k, v in fsd.items():
        if "running_" in k:
            out["bn/" + k] = v.numpy()
    out["param_sum_final"] = np.float64(sum(v.double().sum().item() for n, v in after.items()))
    out["param_names"] = np.array(names)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, {k: round(float(losses[k][0]), 6) for k in losses}, "gnorm", float(gnorm), "dnorm", float(dnorm),
          "lr", lr, "->", path, os.path.getsize(path) // 1024, "KiB")


def unit_vectors(mods, out_dir):
    """Small known-answer vectors for single reference functions (masks, losses, targets)."""
    module, network, utils, train = mods
    train.DEVICE = torch.device("cpu")
    g = torch.Generator().manual_seed(7)
    out = {}
    lens = torch.tensor([5, 1, 3, 7])
    out["lens"] = lens.numpy()
    out["sent_lens_to_mask"] = utils.sent_lens_to_mask(lens, 7).numpy()
    out["causal_mask"] = network.generate_square_subsequent_mask(6, "cpu").numpy()
    # losses
    B, T, V = 3, 9, 46
    mel_len = torch.tensor([9, 4, 6])
    gold = torch.rand(B, T, 80, generator=g)
    pre = torch.randn(B, T, 80, generator=g)
    post = torch.randn(B, T, 80, generator=g)
    stop = torch.randn(B, T, generator=g)
    gold_stop = torch.nn.functional.one_hot(mel_len - 1, T).float()
    out.update(sl_gold=gold.numpy(), sl_pre=pre.numpy(), sl_post=post.numpy(), sl_stop=stop.numpy(),
               sl_len=mel_len.numpy(), sl_gold_stop=gold_stop.numpy())
    out["speech_loss"] = np.float64(train.speech_loss(gold, gold_stop, pre, post, mel_len, stop, 5.0).item())
    logits = torch.randn(B, T, V, generator=g)
    text = torch.randint(3, V, (B, T), generator=g)
    tl = torch.tensor([9, 5, 2])
    for b in range(B):
        text[b, tl[b] - 1] = 2
        text[b, tl[b]:] = 0
    out.update(tl_logits=logits.numpy(), tl_text=text.numpy())
    out["text_loss_w1"] = np.float64(train.text_loss(text, logits.permute(0, 2, 1), 1.0).item())
    out["text_loss_w3"] = np.float64(train.text_loss(text, logits.permute(0, 2, 1), 3.0).item())
    out["disc_target_text"] = train.discriminator_target(4, "text").numpy()
    out["disc_target_speech"] = train.discriminator_target(4, "speech").numpy()
    d_out = torch.randn(8, generator=g)
    d_tgt = torch.cat([train.discriminator_target(4, "text"), train.discriminator_target(4, "speech")])
    out.update(dl_out=d_out.numpy(), dl_tgt=d_tgt.numpy())
    out["disc_loss"] = np.float64(train.discriminator_loss(d_out, d_tgt).item())
    # positional encoding buffer slice + a forward with dropout off
    pe = module.PositionalEncoding(256)
    pe.dropout.p = 0.0
    x = torch.randn(2, 5, 256, generator=g)
    out.update(pe_x=x.numpy(), pe_y=pe(x).numpy(), pe_buf=pe.pe[0, :16].numpy())
    # LR schedules (src/train.py:858-907)
    p = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.SGD([p], lr=0.0625)
    s = train.get_transformer_paper_schedule(o, 2000)
    lrs = []
    for i in range(5):
        lrs.append(o.param_groups[0]["lr"])
        o.step()
        s.step()
    out["sched_transformer_first5"] = np.array(lrs)
    o = torch.optim.SGD([p], lr=1.0)
    s = train.get_linear_schedule_with_warmup(o, 3, 10)
    lrs = []
    for i in range(11):
        lrs.append(o.param_groups[0]["lr"])
        o.step()
        s.step()
    out["sched_linear_11"] = np.array(lrs)
    path = os.path.join(out_dir, "unit.npz")
    np.savez_compressed(path, **out)
    print("unit ->", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--only", default=None, help="generate just this case (name of the .npz without extension)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    mods = import_reference()
    # BASELINE.json config 2's branch at fixture size: 3 layers, no discriminator (generator-only AE + SP + optimizer step)
    if a.only in (None, "step_b3_t20_m56_l3_nodisc"):
        run_case(mods, "step_b3_t20_m56_l3_nodisc", 3, 20, 56, 3, True, a.out, lr_warm_steps=2000, use_discriminator=False)
    if a.only is not None:
        return
    unit_vectors(mods, a.out)
    # config 1 of BASELINE.json (1 utterance, Tt=40, Tm=200, L=4), LR at schedule peak
    run_case(mods, "step_b1_t40_m200_l4", 1, 40, 200, 4, False, a.out, lr_warm_steps=2000)
    # ragged small batch, 2 layers: exercises padding masks, BN over pads, packed LSTM
    run_case(mods, "step_b4_t24_m64_l2", 4, 24, 64, 2, True, a.out, lr_warm_steps=2000)
    # survey's known-answer tuple (lr = 0 at step 0) for cross-check with SURVEY.md section 8c
    run_case(mods, "step_b4_t24_m64_l4_lr0", 4, 24, 64, 4, True, a.out, lr_warm_steps=0)


if __name__ == "__main__":
    main()
