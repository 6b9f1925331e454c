"""Micro-benchmark of the GEMM kernel over the train step's shapes (HIP events, interleaved rounds in one process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
D = torch.device("cuda:0")

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def wgrad(M, N, K, sk, atomic=False):
    dy = torch.randn(K, M, device=D); x = torch.randn(K, N, device=D); dW = torch.zeros(M, N, device=D)
    return timeit(lambda: ops.gemm(ops.OP_RC, ops.OP_RC, dy, M, x, N, dW, N, M, N, K, beta=1, splitk=sk, atomic=atomic))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
    if which == "wgrad":
        for (M, N, K) in [(256, 256, 25600), (256, 1024, 25600), (1024, 256, 25600), (768, 256, 25600), (256, 256, 5760), (1024, 256, 5760), (256, 1280, 25600), (512, 256, 51200)]:
            row = []
            for sk in (8, 16, 32, 64, 100, 128, 200):
                us = wgrad(M, N, K, sk)
                row.append("sk%d:%.0fus(%.0fTF)" % (sk, us, 2.0 * M * N * K / us / 1e6))
            print((M, N, K), " ".join(row), flush=True)
    else:
        for (M, N, K) in [(25600, 256, 256), (25600, 1024, 256), (25600, 256, 1024), (25600, 768, 256), (5760, 256, 256), (5760, 1024, 256), (51200, 512, 256)]:
            x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D); b = torch.randn(N, device=D)
            us = timeit(lambda: ops.linear_fwd(x, W, b, y))
            dx = torch.empty(M, K, device=D)
            us2 = timeit(lambda: ops.linear_dgrad(y, W, dx))
            print((M, N, K), "fwd %.0f us (%.0f TF)  dgrad %.0f us (%.0f TF)" % (us, 2.0*M*N*K/us/1e6, us2, 2.0*M*N*K/us2/1e6), flush=True)
