"""Which gradients are non-finite after the generator sub-steps of one step at the soak's first shape (B=32, T_text=300, T_mel=2000)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
from unast_amd.engine import join_streams
dev = torch.device("cuda:0"); train.DEVICE = dev
B, Tt, Tm = [int(x) for x in os.environ.get("SHAPE", "32,300,2000").split(",")]
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(os.environ.get("DET", "0") == "1")
_, _, model, opt, sched = train.initialize_model(args)
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(B, Tt, Tm, seed=0, ragged=os.environ.get("RAGGED", "1") == "1"))
print("lens text", batch[2].tolist()[:8], "... mel", batch[3].tolist()[:8], flush=True)
losses = defaultdict(list)
train.freeze_model_parameters(model.discriminator)
st = model._store()
def report(tag):
    join_streams(); torch.cuda.synchronize()
    bad = []
    for n, p in st.params.items():
        g = st.grad[st.offsets[n]:st.offsets[n] + p.numel()]
        if not bool(torch.isfinite(g).all()):
            bad.append((n, int((~torch.isfinite(g)).sum()), p.numel()))
    print(tag, {k: round(float(v[-1]), 4) for k, v in losses.items()}, "non-finite grads:", bad[:10], "(%d tensors)" % len(bad), flush=True)
for which in os.environ.get("SUBS", "ae,sp").split(","):
    if which == "ae":
        train.train_ae_step(losses, model, batch, 0, 2, args); report("after AE")
    else:
        train.train_sp_step(losses, model, batch, 0, 2, args); report("after SP")
