"""Forward GEMM timing for one library build (UNAST_HIP_LIB); prints one line per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from bench_gemm import timeit
D = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("UNAST_HIP_LIB", "default"))
res = []
for (M, N, K) in [(25600, 256, 1024), (25600, 256, 256), (25600, 1024, 256), (25600, 768, 256)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D)
    Ws = torch.empty_like(W); ops.split_f32(W, Ws)
    for pre in (0, 1):
        if pre: ops.register_weight_span(W.data_ptr(), W.numel() * 4, Ws.data_ptr())
        us = timeit(lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K), n=30)
        ops.unregister_weight_span(W.data_ptr())
        res.append("%s pre%d %.1f us (%.2f/kstep)" % ((M, N, K), pre, us, us / (K / 32)))
print("%-34s" % tag, " | ".join(res), flush=True)
