import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from collections import defaultdict
import test_gpu_fullsize as T
from unast_amd import train
from unast_amd.engine import join_streams
from unast_amd.portable import synth_batch
def run(defer, lr):
    args, model, opt, sd = T.build(2, lr)
    batch = tuple(torch.from_numpy(x) for x in synth_batch(8, 60, 256, seed=2, ragged=True))
    batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[])
    losses = defaultdict(list)
    for i in range(4):
        train.train_step(losses, model, opt, None, batches, i + 1, args, defer_d_phase=defer)
    join_streams(); torch.cuda.synchronize()
    return {k: [float(x) for x in v] for k, v in losses.items()}, {n: p.detach().cpu().double() for n, p in model.named_parameters()}
for lr in (2e-3, 0.0):
    a, b, c = run(False, lr), run(False, lr), run(True, lr)
    def dl(x, y): return max(abs(p - q) / max(1.0, abs(p)) for k in x[0] for p, q in zip(x[0][k], y[0][k]))
    def dp(x, y): return max(float((x[1][n] - y[1][n]).abs().max() / x[1][n].abs().max().clamp_min(1e-12)) for n in x[1])
    print("lr", lr, "joined vs joined: loss %.2e param %.2e | joined vs deferred: loss %.2e param %.2e" % (dl(a, b), dp(a, b), dl(a, c), dp(a, c)))
    for k in a[0]:
        print("   ", k, ["%.6f" % v for v in a[0][k]], ["%.6f" % v for v in c[0][k]])
