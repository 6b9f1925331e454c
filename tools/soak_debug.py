"""soak_graph.py with a finiteness check after every step (graph and eager), to locate a divergence."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
from unast_amd.graphed import GraphedTrainStep
dev = torch.device("cuda:0"); train.DEVICE = dev
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(False)
_, _, model, opt, sched = train.initialize_model(args)
stepper = GraphedTrainStep(model, opt, sched, args) if mode == "graph" else None
losses = defaultdict(list)
for i in range(int(os.environ.get("STEPS", "120"))):
    shape = (32, 300, 2000) if (i // 20) % 2 == 0 else (16, 180, 800)
    batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(*shape, seed=i % 3, ragged=True))
    b = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
    if stepper is not None:
        stepper(losses, b, i)
    else:
        train.train_step(losses, model, opt, sched, b, i, args, defer_d_phase=True)
    torch.cuda.synchronize()
    last = {k: float(v[-1]) for k, v in losses.items()}
    flat = model._store().flat
    ok = all(v == v for v in last.values()) and bool(torch.isfinite(flat).all())
    if i % 10 == 0 or not ok:
        print(i, shape, "lr %.2e" % opt.param_groups[0]["lr"], {k: round(v, 4) for k, v in last.items()}, "params finite" if bool(torch.isfinite(flat).all()) else "PARAMS NOT FINITE",
              (stepper.stats if stepper else ""), flush=True)
    if not ok:
        g = model._store().grad
        print("first bad step", i, "grad finite:", bool(torch.isfinite(g).all()))
        st = model._store()
        bad = [(n, int((~torch.isfinite(p)).sum()), p.numel()) for n, p in st.params.items() if not bool(torch.isfinite(p).all())]
        print("non-finite parameters:", bad[:12], "... %d tensors" % len(bad))
        print("adam m finite", bool(torch.isfinite(opt._m).all()), "v finite", bool(torch.isfinite(opt._v).all()), "norm^2", [float(x) for x in opt._ss_by_phase.values()])
        break
