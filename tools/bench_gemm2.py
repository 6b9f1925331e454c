import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
print("fwd linear: time per k-step when all blocks are co-resident (<=512 blocks)")
for (M, N, K) in [(1280, 256, 1024), (2560, 256, 1024), (6400, 256, 1024), (12800, 256, 1024), (25600, 256, 1024), (25600, 256, 256), (25600, 256, 2048), (12800, 512, 1024), (25600, 128, 1024)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D)
    for wn in (2, 8):
        us = timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K, tile_wn=wn), n=30)
        print((M, N, K), "wn", wn, "%.1f us  %.2f us/kstep  %.0f TF  A+B+C=%.0f MB -> %.2f TB/s" % (us, us / (K / 32), 2.0 * M * N * K / us / 1e6, (M * K + N * K + M * N) * 4 / 1e6, (M * K + N * K + M * N) * 4 / us / 1e6), flush=True)
print("dgrad / wgrad with both tilings")
for (M, N, K) in [(25600, 256, 1024), (25600, 1024, 256), (5760, 256, 1024)]:
    dy = torch.randn(M, N, device=D); W = torch.randn(N, K, device=D); dx = torch.empty(M, K, device=D)
    for wn in (2, 8):
        us = timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_RC, dy, N, W, K, dx, K, M, K, N, tile_wn=wn), n=30)
        print("dgrad", (M, N, K), "wn", wn, "%.1f us %.0f TF" % (us, 2.0 * M * N * K / us / 1e6), flush=True)
for (M, N, K, sk) in [(256, 256, 25600, 64), (1024, 256, 25600, 24), (256, 1024, 25600, 40)][:1]:
    dy = torch.randn(K, M, device=D); x = torch.randn(K, N, device=D); dW = torch.zeros(M, N, device=D)
    for wn in (2, 8):
        us = timeit(lambda: ops.gemm(ops.OP_RC, ops.OP_RC, dy, M, x, N, dW, N, M, N, K, beta=1, splitk=sk, tile_wn=wn), n=30)
        print("wgrad", (M, N, K), "wn", wn, "%.1f us %.0f TF" % (us, 2.0 * M * N * K / us / 1e6), flush=True)
