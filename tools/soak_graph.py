import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
from unast_amd.graphed import GraphedTrainStep
dev = torch.device("cuda:0"); train.DEVICE = dev
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(False)
_, _, model, opt, sched = train.initialize_model(args)
stepper = GraphedTrainStep(model, opt, sched, args)
losses = defaultdict(list)
t0 = time.perf_counter()
for i in range(120):
    # ragged batches of two shapes alternate: the stepper keeps one capture per input shape
    shape = (32, 300, 2000) if (i // 20) % 2 == 0 else (16, 180, 800)
    batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(*shape, seed=i % 3, ragged=True))
    stepper(losses, dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]), i)
stepper.flush(losses)
torch.cuda.synchronize()
last = {k: round(float(v[-1]), 4) for k, v in losses.items()}
assert all(v == v and abs(v) < 1e6 for v in last.values()), last
print("120 replayed steps over two ragged shapes in %.1f s, losses %s" % (time.perf_counter() - t0, last))
