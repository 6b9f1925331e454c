import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from collections import defaultdict
from test_gpu_parity import build
from oracle import unast_ref as R
from unast_amd import train
from unast_amd.portable import synth_batch
L = 2
args, model, opt, sd = build(L, 0.0)
batch = tuple(torch.from_numpy(x) for x in synth_batch(8, 70, 300, seed=3, ragged=True))
m = R.Model({k: v.clone() for k, v in sd.items()}, L)
for n, p in m.P.items():
    if n.startswith("discriminator."): p.requires_grad_(False)
ae = R.generator_losses(m, batch); ae.pop("_ae_out"); (sum(ae.values()) / 2).backward()
sp = R.supervised_losses(m, batch); (sum(sp.values()) / 2).backward()
losses = defaultdict(list); model.train()
train.freeze_model_parameters(model.discriminator)
train.train_ae_step(losses, model, batch, 0, 2, args); train.train_sp_step(losses, model, batch, 0, 2, args)
model.expose_grads()
rows = []
for n, p in model.named_parameters():
    r = m.P[n].grad
    if r is None: continue
    d = p.grad.cpu().double() - r.double()
    rows.append((d.abs().max().item() / max(r.abs().max().item(), 1e-30), d.norm().item() / max(r.double().norm().item(), 1e-30), n))
rows.sort(reverse=True)
for a, b, n in rows[:25]: print("%.2e %.2e %s" % (a, b, n))
print("median maxrel %.2e normrel %.2e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
