import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unast_amd import ops
D = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(25600, 256, 256), (25600, 256, 1024), (25600, 768, 256), (25600, 1024, 256), (25600, 512, 256)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D); b = torch.randn(N, device=D)
    r = []
    for wn in (0, 4, 2):
        r.append(timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K, bias=b, tile_wn=wn)))
    print((M, N, K), "default(8 waves,128x128) %.1f us   128x256 %.1f us   4 waves 128x128 %.1f us" % tuple(r), flush=True)
