"""Replayed against eager steps, to the bit, for longer than the test suite does: N steps (default 36) alternating between config 5's shape
(B=32 / T_text=300 / T_mel=2000) and B=16 / 180 / 800 and config 3's (B=32 / 180 / 800), ragged lengths, with every fp32 sum in a fixed
order (utils.set_deterministic(True, fixed_sums=True)); the replayed runs are captured with their streams shifted against each other by
random spins (config.STREAM_JITTER, two seeds).  Prints the first step at which losses or parameters differ, or that none does.
usage (GPU box): python tools/soak_bitexact.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
import torch
from unast_amd import config, train, utils
from unast_amd.configs import make_args
from unast_amd.engine import join_streams
from unast_amd.graphed import GraphedTrainStep
from unast_amd.portable import synth_batch

D = torch.device("cuda:0")
train.DEVICE = D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 36
shapes = ((32, 300, 2000), (16, 180, 800), (32, 180, 800))
data = {s: [tuple(torch.from_numpy(x).to(D) for x in synth_batch(*s, seed=k, ragged=True)) for k in range(2)] for s in shapes}
utils.set_deterministic(True, fixed_sums=True)


def run(graphed, jitter=0, seed=0):
    config.STREAM_JITTER, config.STREAM_JITTER_SEED = jitter, seed
    args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, warmup_steps=200)
    utils.set_seed(0)
    _, _, model, opt, sched = train.initialize_model(args)
    stepper = GraphedTrainStep(model, opt, sched, args) if graphed else None
    losses = defaultdict(list)
    for i in range(N):
        batch = data[shapes[(i // 4) % 3]][i % 2]
        b = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
        if graphed:
            stepper(losses, b, i)
        else:
            train.train_step(losses, model, opt, sched, b, i, args, defer_d_phase=True)
    if graphed:
        stepper.flush(losses)
        print("   ", stepper.stats, flush=True)
    join_streams(); torch.cuda.synchronize()
    config.STREAM_JITTER = 0
    return {k: [float(x) for x in v] for k, v in losses.items()}, model._store().flat.detach().clone()


ref_l, ref_p = run(False)
assert bool(torch.isfinite(ref_p).all())
ok = True
for (jit, seed) in ((0, 0), (300, 21), (300, 22)):
    l, p = run(True, jit, seed)
    first = min([i for k in ref_l for i, (x, y) in enumerate(zip(ref_l[k], l[k])) if x != y] or [None], key=lambda v: (v is None, v))
    same = torch.equal(p, ref_p)
    ok = ok and first is None and same
    print("replayed, jitter %3d us seed %2d: losses of %d steps %s, parameters %s" % (
        jit, seed, N, "equal" if first is None else "DIFFER from step %d" % first, "equal" if same else "DIFFER (max %.3e)" % float((p - ref_p).abs().max())), flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
