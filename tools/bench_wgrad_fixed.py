"""Fixed cost vs per-k-step cost of the weight-gradient launch (256x256 output, 64 splits = 256 workgroups) and of its split-K
reduction: tokens = 64 splits x 32 x ksteps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
N = K = 256
for ks in (1, 2, 4, 8, 12, 16, 32):
    M = 64 * 32 * ks
    dy = torch.randn(M, N, device=D); x = torch.randn(M, K, device=D); dW = torch.zeros(N, K, device=D)
    us = timeit(lambda: ops.gemm(ops.OP_RC, ops.OP_RC, dy, N, x, K, dW, K, N, K, M, beta=1, splitk=64), n=30)
    us1 = timeit(lambda: ops.gemm(ops.OP_RC, ops.OP_RC, dy, N, x, K, dW, K, N, K, M, beta=1, splitk=1), n=5) if ks <= 2 else float("nan")
    print("ksteps/WG %2d (tokens %5d): %6.1f us with 64 splits (gemm + reduce)   [1 split, 4 workgroups: %6.1f us]" % (ks, M, us, us1), flush=True)
# the reduce alone: time a launch with the same slab size through the public path is not separable; estimate from an empty-ish GEMM
