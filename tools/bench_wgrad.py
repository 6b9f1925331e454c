"""Weight-gradient contraction dW[N,K] += dY[M,N]^T X[M,K] (M = tokens) by tile variant and split count, split-K reduction
included: which (tile, split) is best per shape, and how far from streaming dY and X once?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
for (N, K, M) in [(256, 256, 25600), (768, 256, 25600), (1024, 256, 25600), (256, 1024, 25600), (256, 256, 5760), (1024, 256, 5760)]:
    dy = torch.randn(M, N, device=D); x = torch.randn(M, K, device=D); dW = torch.zeros(N, K, device=D)
    mb = (M * N + M * K) * 4 / 1e6
    auto = ops._splitk_for(N, K, M)
    best = None
    for wn in (2, 8, 4):
        for sk in sorted(set([auto, max(1, auto // 2), auto * 2, max(1, auto // 4)])):
            try:
                us = timeit(lambda: ops.gemm(ops.OP_RC, ops.OP_RC, dy, N, x, K, dW, K, N, K, M, beta=1, splitk=sk, tile_wn=wn), n=20)
            except Exception as e:
                print((N, K, M), "wn", wn, "sk", sk, "failed:", str(e)[:80]); continue
            tag = " (auto)" if (sk == auto and wn == 2) else ""
            print("wgrad %s wn %d splitk %3d: %6.1f us  dY+X = %.0f MB -> %.2f TB/s%s" % ((N, K, M), wn, sk, us, mb, mb / us, tag), flush=True)
            if best is None or us < best[0]:
                best = (us, wn, sk)
    print("   best:", best, flush=True)
