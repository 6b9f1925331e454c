"""Does the row stride of A (power of two vs padded) change the forward GEMM time?  (L2-channel / TCP-set camping test)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
for (M, N, K) in [(1280, 256, 1024), (25600, 256, 1024), (25600, 256, 256), (25600, 1024, 256), (25600, 768, 256)]:
    W = torch.randn(N, K, device=D)
    res = []
    for pad in (0, 16, 32, 64, 96):
        xb = torch.randn(M, K + pad, device=D); x = xb[:, :K]
        for padc in ((0, 32) if pad in (0, 32) else (0,)):
            yb = torch.empty(M, N + padc, device=D); y = yb[:, :N]
            us = timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K + pad, W, K, y, N + padc, M, N, K), n=30)
            res.append("lda=K+%d ldc=N+%d: %.1f us" % (pad, padc, us))
    print((M, N, K), " | ".join(res), flush=True)
