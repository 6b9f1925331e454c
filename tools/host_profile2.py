"""Host-side cost of enqueueing one train step: tiny batch (the GPU is never the bound), full depth (L=4: the launch count of
config 3).  Prints ms/step of pure enqueue and a cProfile by internal time."""
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
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(2, 32, 128, seed=0))
batches = dict(unsup=[batch], sup=[batch], disc=[batch]); losses = defaultdict(list)
for i in range(5): train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): train.train_step(losses, model, opt, sched, batches, 5 + i, args, defer_d_phase=True)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue ms/step: %.2f" % ((t1 - t0) / 10 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(3): train.train_step(losses, model, opt, sched, batches, 20 + i, args, defer_d_phase=True)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
