"""Per-step host enqueue time vs completion time at config 3 (each step starts with an empty GPU queue, so the host is never
blocked by queue back-pressure): is the step host-bound or GPU-bound?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
dev = torch.device("cuda:0"); train.DEVICE = dev
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(False)
_, _, model, opt, sched = train.initialize_model(args)
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(32, 180, 800, seed=0))
batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]); losses = defaultdict(list)
for i in range(3): train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True)
torch.cuda.synchronize()
hs, ts = [], []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train.train_step(losses, model, opt, sched, batches, 3 + i, args, defer_d_phase=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print("host enqueue ms/step: min %.1f median %.1f | enqueue+drain ms/step: median %.1f" % (min(hs), sorted(hs)[len(hs) // 2], sorted(ts)[len(ts) // 2]))
# back-to-back (as bench.py)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8): train.train_step(losses, model, opt, sched, batches, 20 + i, args, defer_d_phase=True)
torch.cuda.synchronize(); print("back-to-back ms/step: %.1f" % ((time.perf_counter() - t0) / 8 * 1e3))
