"""Tile GEMM at the text side's shapes (5 760 rows into 256 columns): 128x128 tiles (90 workgroups) against 128x64 tiles (180)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_panel import timeit
D = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, rc) in [(5760, 256, 256, False), (5760, 256, 1024, False), (5760, 256, 1024, True), (5760, 256, 768, True), (5760, 512, 256, False), (5760, 1024, 256, False), (9600, 256, 256, False), (96, 768, 256, False)]:
    x = torch.randn(M, K, device=D); b = torch.randn(N, device=D)
    W = (torch.randn(K, N, device=D) if rc else torch.randn(N, K, device=D)) * 0.05
    y = {}
    t = {}
    for wn in (8, 1):
        y[wn] = torch.empty(M, N, device=D)
        if rc:
            f = lambda wn=wn: ops.gemm(ops.OP_KC, ops.OP_RC, x, K, W, N, y[wn], N, M, N, K, tile_wn=wn)
        else:
            f = lambda wn=wn: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y[wn], N, M, N, K, bias=b, tile_wn=wn)
        f(); torch.cuda.synchronize()
        t[wn] = (timeit(f), timeit(f))
    print("M=%5d N=%4d K=%4d %s  128x128: %.1f/%.1f us   128x64: %.1f/%.1f us   identical: %s" % (M, N, K, "W row-major (dgrad)" if rc else "W k-contiguous     ", t[8][0], t[8][1], t[1][0], t[1][1], torch.equal(y[8], y[1])), flush=True)
