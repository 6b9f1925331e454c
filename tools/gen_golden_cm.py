#!/usr/bin/env python3
"""Golden vectors for the cross-model (back-translation) step and autoregressive inference.
Runs the reference's own infer_sequence / crossmodel_step (src/network.py:103-123, 219-252, 455-481; src/train.py:261-294)
with RNG sites off and portable weights; writes tests/golden/cm_*.npz.  Build container only (imports /root/reference)."""
import os, sys, collections
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE)
import gen_golden as G
from portable_init import portable_state_dict, synth_batch


def run(mods, name, B, Tt, Tm, L, seed, out_dir, text_cap, mel_cap):
    module, network, utils, train = mods
    args = G.make_args(L)
    train.DEVICE = torch.device("cpu"); train.WRITER = None
    utils.set_seed(0)
    _, _, model, opt, sched = train.initialize_model(args)
    sd = portable_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    G.deterministic_mode(model, network, train)
    model.train()
    text, mel, text_len, mel_len = synth_batch(B, Tt, Tm, seed=0, ragged=True)
    batch = tuple(torch.from_numpy(x) for x in (text, mel, text_len, mel_len))
    out = {"text": text, "mel": mel, "text_len": text_len, "mel_len": mel_len, "meta": np.array([B, Tt, Tm, L, seed, text_cap, mel_cap], np.int64)}
    # smaller generation caps keep the fixture small; they are plain arguments of infer_sequence
    orig_s, orig_t = model.speech_m.infer_sequence, model.text_m.infer_sequence
    model.speech_m.infer_sequence = lambda memory, masks, max_len=mel_cap: orig_s(memory, masks, max_len)
    model.text_m.infer_sequence = lambda memory, masks, max_len=text_cap: orig_t(memory, masks, max_len)
    with torch.no_grad():
        (t, m, tl, ml), _ = train.process_batch(batch)
        t_enc, t_masks = model.text_m.encode(t, tl)
        pre, post, stops, slens = model.speech_m.infer_sequence(t_enc, t_masks)
        s_enc, s_masks = model.speech_m.encode(m, ml)
        tp, tplens = model.text_m.infer_sequence(s_enc, s_masks)
        out.update(inf_pre=pre.numpy(), inf_post=post.numpy(), inf_stop=stops.numpy(), inf_slens=slens.numpy(),
                   inf_text=tp.numpy(), inf_tlens=tplens.numpy())
    losses = collections.defaultdict(list)
    train.freeze_model_parameters(model.discriminator)
    train.train_cm_step(losses, model, batch, 0, 3, args)
    for k in ("s_cm", "t_cm", "d_cm"):
        out["loss/" + k] = np.float64(losses[k][0])
    names = [n for n, _ in model.named_parameters()]
    out["grad_norms"] = np.array([p.grad.double().norm().item() if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    out["param_names"] = np.array(names)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, {k: round(float(v[0]), 5) for k, v in losses.items()}, "speech lens", slens.tolist(), "text lens", tplens.tolist(),
          "shapes", pre.shape, tp.shape, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    mods = G.import_reference()
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    torch.set_num_threads(8)
    for seed in (1234, 77):
        run(mods, "cm_b3_t12_m40_l2_s%d" % seed, 3, 12, 40, 2, seed, out_dir, text_cap=20, mel_cap=30)
