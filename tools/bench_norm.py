"""Micro-benchmark of the LayerNorm kernels at the train step's shapes (rows x 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
D = torch.device("cuda:0")

def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for rows in (25600, 5760, 64000):
    C = 256
    z = torch.randn(rows, C, device=D); dy = torch.randn(rows, C, device=D); g = torch.randn(C, device=D); b = torch.randn(C, device=D)
    y = torch.empty_like(z); mean = torch.empty(rows, device=D); rstd = torch.empty(rows, device=D)
    dz = torch.empty_like(z); dzd = torch.empty_like(z); dg = torch.zeros(C, device=D); db = torch.zeros(C, device=D)
    f = timeit(lambda: ops.layernorm_fwd(z, g, b, y, mean, rstd))
    bw = timeit(lambda: ops.layernorm_bwd(dy, z, g, mean, rstd, dz, dzd, dg, db, drop_p=0.1, seed=1, stream_id=1))
    bw0 = timeit(lambda: ops.layernorm_bwd(dy, z, g, mean, rstd, dz, None, dg, db))
    mb = rows * C * 4 / 1e6
    print("rows %6d  fwd %5.1f us (%.2f TB/s)   bwd+drop %5.1f us (%.2f TB/s)   bwd %5.1f us (%.2f TB/s)" % (rows, f, 2 * mb / f, bw, 4 * mb / bw, bw0, 3 * mb / bw0), flush=True)
