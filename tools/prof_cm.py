"""Host-side profile (cProfile, wall clock incl. a final device synchronise) of one cross-model sub-step at config-3 size."""
import cProfile
import os
import pstats
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import train, utils                       # noqa: E402
from unast_amd.configs import make_args                  # noqa: E402
from unast_amd.portable import synth_batch, portable_tensor   # noqa: E402
from unast_amd.spec import state_dict_spec               # noqa: E402

B, Tt, Tm, L = 32, 180, 800, 4
dev = torch.device("cuda:0")
train.DEVICE = dev
utils.set_seed(0)
args = make_args(num_layers=L, cm_steps=1)
_, _, model, opt, _ = train.initialize_model(args)
model.load_state_dict({k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()})
model.train()
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(B, Tt, Tm, 0))
losses = defaultdict(list)


def one():
    train.train_cm_step(losses, model, batch, 0, 1, args)
    torch.cuda.synchronize()


one()
t0 = time.perf_counter()
one()
print("cm sub-step: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
one()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
