"""The input-gradient GEMM + LayerNorm backward as one launch (ops.linear_dgrad_lnbwd) against the two launches it replaces, at the shapes
of the speech side (rows = 2 x 32 x 800 paired, contraction 1024 = FFN linear1, 768 = self-attention in-projection).  HIP events, interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import config, ops
from unast_amd.planes import Planes
D = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (51200, 25600):
    for K in (1024, 768):
        E = 256
        dy = torch.randn(M, K, device=D); W = torch.randn(K, E, device=D) * 0.05; R = torch.randn(M, E, device=D)
        z = torch.randn(M, E, device=D); gamma = torch.rand(E, device=D) + 0.5
        mean = z.mean(1).contiguous(); rstd = (z.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
        pl = Planes([W], transposed=True)
        ops._weight_planes = lambda w, transposed=False: pl.ref(0) if transposed else None
        dx = torch.empty(M, E, device=D); dz = torch.empty(M, E, device=D); dzd = torch.empty(M, E, device=D)
        dg = torch.zeros(E, device=D); db = torch.zeros(E, device=D)
        config.LN_FINALIZE_OFFLOAD = False
        a = lambda: ops.linear_dgrad(dy, W, dx, R=R)
        b = lambda: ops.layernorm_bwd(dx, z, gamma, mean, rstd, dz, dzd, dg, db, drop_p=0.1, seed=3, stream_id=2)
        c = lambda: ops.linear_dgrad_lnbwd(dy, W, R, z, mean, rstd, gamma, dz, dzd, dg, db, drop_p=0.1, seed=3, stream_id=2)
        ta, tb, tc = timeit(a), timeit(b), timeit(c)
        ta2, tb2, tc2 = timeit(a), timeit(b), timeit(c)
        print("M=%6d K=%5d  GEMM %.1f/%.1f us + LayerNorm backward (with its finalize) %.1f/%.1f us = %.1f   fused (with its finalize) %.1f/%.1f us" % (
            M, K, ta, ta2, tb, tb2, min(ta, ta2) + min(tb, tb2), tc, tc2), flush=True)
