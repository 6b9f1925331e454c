"""Achievable HBM rates on this chip with plain streaming kernels (torch fill / copy / sum): the ceilings the GEMM epilogues' writes meet."""
import torch
D = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (26, 105, 420, 1680):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device=D); y = torch.empty(n, device=D)
    tf = timeit(lambda: x.zero_()); tc = timeit(lambda: y.copy_(x)); ts = timeit(lambda: x.sum())
    print("%5d MB: fill %.2f TB/s (%.1f us)   copy %.2f TB/s read+write (%.1f us)   sum %.2f TB/s (%.1f us)" % (mb, mb * 1e6 / tf / 1e12, tf * 1e6, 2 * mb * 1e6 / tc / 1e12, tc * 1e6, mb * 1e6 / ts / 1e12, ts * 1e6), flush=True)
