"""Odd shapes and step layouts through the eager step and the replayed capture: finite losses, eager == replay at frozen parameters.
(tiny / non-multiple lengths, batch 1, accumulation over several sub-steps, more input signatures than cached captures)."""
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


def run(graphed, shapes, max_graphs=None, **kw):
    import unast_amd.graphed as G
    G.MAX_GRAPHS = max_graphs or 32
    utils.set_seed(0); utils.set_deterministic(True)
    args = make_args(num_layers=2, cm_steps=0, lr=1e-7, **kw)
    _, _, model, opt, sched = train.initialize_model(args)
    stepper = GraphedTrainStep(model, opt, None, args) if graphed else None
    losses = defaultdict(list)
    for i, (B, Tt, Tm) in enumerate(shapes):
        mk = lambda s: tuple(torch.from_numpy(x).to(D) for x in synth_batch(B, Tt, Tm, seed=s, ragged=True))
        b = dict(unsup=[mk(7 * i + j) for j in range(args.ae_steps)], sup=[mk(7 * i + 3 + j) for j in range(args.sp_steps)],
                 disc=[mk(7 * i + 5 + j) for j in range(args.d_steps)], cm=[])
        if graphed:
            stepper(losses, b, i)
        else:
            train.train_step(losses, model, opt, None, b, i, args, defer_d_phase=True)
    if graphed:
        stepper.flush(losses)
        if max_graphs:
            assert stepper.stats["evictions"] > 0 and len(stepper.graphs) <= max_graphs, stepper.stats
    join_streams(); torch.cuda.synchronize()
    return {k: [float(x) for x in v] for k, v in losses.items()}


def check(name, shapes, **kw):
    e, g = run(False, shapes, **kw), run(True, shapes, **kw)
    worst = 0.0
    for k in e:
        assert len(e[k]) == len(g[k]), (name, k, len(e[k]), len(g[k]))
        for x, y in zip(e[k], g[k]):
            assert np.isfinite(x) and np.isfinite(y), (name, k, x, y)
            worst = max(worst, abs(x - y) / max(1.0, abs(x)))
    assert worst < 5e-5, (name, worst)
    print("%-46s ok  (%d steps, worst eager-vs-replay loss difference %.1e)" % (name, len(shapes), worst), flush=True)


check("tiny lengths (T_text 3, T_mel 7)", [(2, 3, 7)] * 5)
check("batch 1, odd lengths", [(1, 17, 33)] * 5)
check("lengths not multiples of 16", [(3, 21, 131)] * 5)
check("six signatures in turn (six shape pairs)", [(2, 8 + 4 * (i % 6), 32 + 16 * (i % 6)) for i in range(24)])
check("six signatures, three cached captures (eviction)", [(2, 8 + 4 * (i % 6), 32 + 16 * (i % 6)) for i in range(30)], max_graphs=3)
check("accumulation: ae 2, sp 2, d 2", [(2, 12, 40)] * 5, ae_steps=2, sp_steps=2, d_steps=2)
check("generator only", [(2, 12, 40)] * 5, use_discriminator=False)
print("stress ok")
