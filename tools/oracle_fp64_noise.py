"""How much rounding noise is in the reference-precision (fp32) gradients themselves?  The pinned oracle is evaluated in fp32
and -- the same code, `.float()` casts redirected -- in fp64 on the ragged B=8 case of tests/test_gpu_parity.py; the
norm-relative difference per parameter is the floor below which a comparison with the fp32 reference says nothing.
CPU only (test infrastructure: imports oracle/)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unast_ref as R                      # noqa: E402
from unast_amd.portable import synth_batch, portable_tensor   # noqa: E402
from unast_amd.spec import state_dict_spec             # noqa: E402

L = 2
sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()}
batch = tuple(torch.from_numpy(x) for x in synth_batch(8, 70, 300, seed=3, ragged=True))


def grads(dtype):
    orig = torch.Tensor.float
    if dtype == torch.float64:
        torch.Tensor.float = lambda self: self.double()
        torch.set_default_dtype(torch.float64)
    try:
        m = R.Model({k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, L)
        for n, p in m.P.items():
            if n.startswith("discriminator."):
                p.requires_grad_(False)
        b = (batch[0], batch[1].to(dtype), batch[2], batch[3])
        ae = R.generator_losses(m, b)
        ae.pop("_ae_out")
        (sum(ae.values()) / 2).backward()
        sp = R.supervised_losses(m, b)
        (sum(sp.values()) / 2).backward()
        return {n: p.grad.double() for n, p in m.P.items() if p.grad is not None}
    finally:
        torch.Tensor.float = orig
        torch.set_default_dtype(torch.float32)


g32, g64 = grads(torch.float32), grads(torch.float64)
tot = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
rows = sorted(((g32[n] - g64[n]).norm().item() / g64[n].norm().item(), n) for n in g64 if g64[n].norm().item() >= 1e-5 * tot)
rows.reverse()
print("fp32 oracle vs fp64 oracle, norm-relative gradient difference: median %.2e, max %.2e" % (np.median([r[0] for r in rows]), rows[0][0]))
for r in rows[:12]:
    print("  %.2e  %s" % r)

if torch.cuda.is_available():
    # the HIP path on the same case, against the same fp64 evaluation
    from collections import defaultdict
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    D = torch.device("cuda:0")
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
    train.DEVICE = D
    utils.set_seed(0)
    utils.set_deterministic(True)
    _, _, model, opt, _ = train.initialize_model(args)
    model.load_state_dict(sd)
    model.train()
    train.freeze_model_parameters(model.discriminator)
    losses = defaultdict(list)
    train.train_ae_step(losses, model, batch, 0, 2, args)
    train.train_sp_step(losses, model, batch, 0, 2, args)
    model.expose_grads()
    hip = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    rows = sorted(((hip[n] - g64[n]).norm().item() / g64[n].norm().item(), (g32[n] - g64[n]).norm().item() / g64[n].norm().item(), n)
                  for n in g64 if n in hip and g64[n].norm().item() >= 1e-5 * tot)
    rows.reverse()
    print("HIP (split-bf16) vs fp64: median %.2e, max %.2e   [fp32 reference vs fp64: median %.2e]" % (
        np.median([r[0] for r in rows]), rows[0][0], np.median([r[1] for r in rows])))
    print("  HIP-vs-fp64  fp32-vs-fp64  parameter")
    for r in rows[:15]:
        print("  %.2e     %.2e      %s" % r)
    worse = sum(1 for r in rows if r[0] > 10 * max(r[1], 1e-6))
    print("parameters whose HIP error exceeds 10x the fp32 reference's own error: %d of %d" % (worse, len(rows)))
