"""Step time on a RAGGED config-3 batch (lengths uniform in [T/2, T]; bench.py's batch is full length): what padding-aware options are worth.
usage: [UNAST_ENC_SKIP_PAD_GRADS=1] python tools/bench_ragged.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
from unast_amd import train, utils, config
from unast_amd.configs import make_args
from unast_amd.graphed import GraphedTrainStep
from unast_amd.portable import synth_batch
D = torch.device("cuda:0"); train.DEVICE = D
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0)
_, _, model, opt, sched = train.initialize_model(args)
stepper = GraphedTrainStep(model, opt, sched, args)
batch = tuple(torch.from_numpy(x).to(D) for x in synth_batch(32, 180, 800, seed=1, ragged=True))
b = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
losses = defaultdict(list)
for i in range(8): stepper(losses, b, i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(30): stepper(losses, b, 8 + i)
torch.cuda.synchronize()
print("ragged c3 batch (mean mel length %.0f of 800): %.2f ms/step, ENC_SKIP_PAD_GRADS=%s" % (float(batch[3].float().mean()), (time.perf_counter() - t0) / 30 * 1e3, config.ENC_SKIP_PAD_GRADS))
