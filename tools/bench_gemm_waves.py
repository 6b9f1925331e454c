"""Forward GEMM time vs the number of tile waves and vs the tile shape for the wide-output shapes (FFN1, QKV)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
for (M, N, K) in [(12800, 256, 256), (25600, 256, 256), (51200, 256, 256), (102400, 256, 256), (25600, 1024, 256), (25600, 768, 256), (25600, 512, 256), (25600, 256, 1024), (5760, 1024, 256)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D)
    mb = (M * K + N * K + M * N) * 4 / 1e6
    res = []
    for wn in (8, 4, 2):
        us = timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K, tile_wn=wn), n=30)
        res.append("wn%d %.1f us (%.2f TB/s)" % (wn, us, mb / us))
    print((M, N, K), "tiles128", (M // 128) * (N // 128), "  ".join(res), flush=True)
