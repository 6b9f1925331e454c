import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from collections import defaultdict
from test_gpu_parity import build, load_case
from unast_amd import train
name = sys.argv[1]; use_d = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g, batch = load_case("tests/golden", name)
B, Tt, Tm, L, _ = [int(v) for v in g["meta"]]
args, model, opt, sd = build(L, float(g["lr"]))
names = [str(n) for n in g["param_names"]]; params = dict(model.named_parameters())
losses = defaultdict(list); model.train()
train.freeze_model_parameters(model.discriminator)
train.train_ae_step(losses, model, batch, 0, 2, args)
train.train_sp_step(losses, model, batch, 0, 2, args)
model.expose_grads()
print({k: float(v[0]) for k, v in losses.items()})
for n, r in zip(names, g["gen_grad_norms"]):
    a = params[n].grad.double().norm().item() if params[n].grad is not None else -1
    if abs(a - r) > 1e-3 * r + 1e-6: print("%-80s %.6f %.6f ratio %.4f" % (n, a, r, a / max(r, 1e-30)))
