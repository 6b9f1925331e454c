"""Eager step vs replayed capture at frozen parameters over large and odd shapes (config 5's B=32 / T_text=300 / T_mel=2000 among them),
shape changes every two steps: every loss of every step must agree, parameters must stay finite."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
import numpy as np
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.engine import join_streams
from unast_amd.graphed import GraphedTrainStep
from unast_amd.portable import synth_batch
D = torch.device("cuda:0"); train.DEVICE = D
SHAPES = [(32, 300, 2000), (16, 180, 800), (8, 97, 1203), (32, 180, 800), (5, 301, 333)]
seq = [SHAPES[(i // 3) % len(SHAPES)] for i in range(30)]


def run(graphed):
    utils.set_seed(0); utils.set_deterministic(True)
    args = make_args(num_layers=4, cm_steps=0, ae_steps=1, sp_steps=1, d_steps=1, lr=1e-7)
    _, _, model, opt, sched = train.initialize_model(args)
    stepper = GraphedTrainStep(model, opt, None, args) if graphed else None
    losses = defaultdict(list)
    for i, (B, Tt, Tm) in enumerate(seq):
        mk = lambda s: tuple(torch.from_numpy(x).to(D) for x in synth_batch(B, Tt, Tm, seed=s, ragged=True))
        b = dict(unsup=[mk(3 * i)], sup=[mk(3 * i + 1)], disc=[mk(3 * i + 2)], cm=[])
        if graphed:
            stepper(losses, b, i)
        else:
            train.train_step(losses, model, opt, None, b, i, args, defer_d_phase=True)
    if graphed:
        stepper.flush(losses)
        print("graph cache:", {k: v for k, v in stepper.cache_report().items() if k != "per_capture"}, flush=True)
    join_streams(); torch.cuda.synchronize()
    assert bool(torch.isfinite(model._store().flat).all())
    return {k: [float(x) for x in v] for k, v in losses.items()}


e, g = run(False), run(True)
worst = 0.0
for k in e:
    assert len(e[k]) == len(g[k]), (k, len(e[k]), len(g[k]))
    for i, (x, y) in enumerate(zip(e[k], g[k])):
        assert np.isfinite(x) and np.isfinite(y), (k, i, x, y)
        worst = max(worst, abs(x - y) / max(1.0, abs(x)))
assert worst < 5e-5, worst
print("stress big ok: %d steps over %d shapes, worst eager-vs-replay loss difference %.1e" % (len(seq), len(SHAPES), worst))
