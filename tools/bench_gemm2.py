import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
print("fwd linear: time per k-step when all blocks are co-resident (<=512 blocks)")
for (M, N, K) in [(1280, 256, 1024), (2560, 256, 1024), (6400, 256, 1024), (12800, 256, 1024), (25600, 256, 1024), (25600, 256, 256), (25600, 256, 2048), (12800, 512, 1024), (25600, 128, 1024)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D)
    for wn in (2, 4):
        us = timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K, tile_wn=wn), n=30)
        print((M, N, K), "wn", wn, "%.1f us  %.2f us/kstep  %.0f TF  A+B+C=%.0f MB -> %.2f TB/s" % (us, us / (K / 32), 2.0 * M * N * K / us / 1e6, (M * K + N * K + M * N) * 4 / 1e6, (M * K + N * K + M * N) * 4 / us / 1e6), flush=True)
