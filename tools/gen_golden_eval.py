#!/usr/bin/env python3
"""Golden vectors for the evaluation path: runs the reference's own evaluate()
(src/train.py:474-565) in eval mode (BatchNorm running statistics, dropout off) on two synthetic paired batches, with
RNG sites off and portable weights; writes tests/golden/eval_*.npz.  jiwer is absent from this image, so compute_per is
replaced by a recorder: the fixture holds the (truth, hypothesis, lengths) it was called with, not a PER value.
Build container only (imports /root/reference)."""
import os, sys
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
    orig_s, orig_t = model.speech_m.infer_sequence, model.text_m.infer_sequence
    model.speech_m.infer_sequence = lambda memory, masks, max_len=mel_cap: orig_s(memory, masks, max_len)
    model.text_m.infer_sequence = lambda memory, masks, max_len=text_cap: orig_t(memory, masks, max_len)
    batches = [synth_batch(B, Tt, Tm, seed=s, ragged=True) for s in (0, 1)]
    out = {"meta": np.array([B, Tt, Tm, L, seed, text_cap, mel_cap, len(batches)], np.int64)}
    for i, b in enumerate(batches):
        for k, v in zip(("text", "mel", "text_len", "mel_len"), b):
            out["b%d/%s" % (i, k)] = v
    calls = []
    train.compute_per = lambda gt, hyp, gl, hl: (calls.append([x.detach().cpu().numpy().copy() for x in (gt, hyp, gl, hl)]), 0.0)[1]
    train.compare_outputs = lambda *a, **k: None
    loader = [tuple(torch.from_numpy(x) for x in b) for b in batches]
    per, losses = train.evaluate(model, loader, 0, args)
    for k, v in losses.items():
        out["loss/" + k] = np.array(v, np.float64)
    for i, (gt, hyp, gl, hl) in enumerate(calls):
        out["b%d/asr_tokens" % i] = hyp
        out["b%d/asr_lens" % i] = hl
    assert not model.training
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, {k: [round(x, 5) for x in v] for k, v in losses.items()}, "asr lens", [c[3].tolist() for c in calls], os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    mods = G.import_reference()
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    torch.set_num_threads(8)
    run(mods, "eval_b3_t12_m40_l2_s77", 3, 12, 40, 2, 77, out_dir, text_cap=20, mel_cap=30)
