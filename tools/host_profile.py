"""cProfile of the host-side enqueue path of one train step (config 3)."""
import cProfile, pstats, os, sys, time, torch
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
batches = dict(unsup=[batch], sup=[batch], disc=[batch]); losses = defaultdict(list)
for i in range(3): train.train_step(losses, model, opt, sched, batches, i, args)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(3): train.train_step(losses, model, opt, sched, batches, 3 + i, args)
t1 = time.perf_counter()
pr.disable(); torch.cuda.synchronize()
print("host enqueue ms/step (under cProfile): %.1f" % ((t1 - t0) / 3 * 1e3))
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
