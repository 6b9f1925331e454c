"""Soak: N config-3 train steps (deferred discriminator phase, side streams); memory must stay flat and losses finite."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0"); train.DEVICE = dev
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(False)
_, _, model, opt, sched = train.initialize_model(args)
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(32, 180, 800, seed=0))
batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]); losses = defaultdict(list)
marks = []
t0 = time.perf_counter()
for i in range(N):
    train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True)
    if i in (20, N // 2, N - 1):
        torch.cuda.synchronize()
        marks.append((i, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, {k: round(float(v[-1]), 4) for k, v in losses.items()}))
torch.cuda.synchronize()
print("%d steps in %.1f s (%.1f ms/step)" % (N, time.perf_counter() - t0, (time.perf_counter() - t0) / N * 1e3))
for m in marks:
    print("step %d: allocated %.2f GiB, reserved %.2f GiB, losses %s" % m)
assert marks[-1][2] <= marks[1][2] * 1.02 + 0.25, "reserved memory keeps growing"      # the caching allocator settles within the first ~100 steps
assert all(v == v and abs(v) < 1e6 for v in marks[-1][3].values()), "non-finite losses"
print("soak ok")
