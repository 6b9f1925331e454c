import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
from unast_amd.engine import join_streams
dev = torch.device("cuda:0"); train.DEVICE = dev
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(False)
_, _, model, opt, sched = train.initialize_model(args)
for shape in ((64, 400, 3000), (96, 180, 800), (8, 600, 4800), (32, 180, 800)):
    losses = defaultdict(list)
    batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(*shape, seed=1, ragged=True))
    t0 = time.perf_counter()
    for i in range(3):
        train.train_step(losses, model, opt, sched, dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]), i, args, defer_d_phase=True)
    join_streams(); torch.cuda.synchronize()
    last = {k: round(float(v[-1]), 4) for k, v in losses.items()}
    assert all(v == v and abs(v) < 1e6 for v in last.values()), (shape, last)
    print(shape, "3 steps %.2f s, reserved %.1f GiB" % (time.perf_counter() - t0, torch.cuda.memory_reserved() / 2**30), last, flush=True)
print("big ok")
