"""K-streamed form of the panel GEMM (K > 256 into 256 columns) against the tile GEMM (+ stand-alone LayerNorm) on the train step's
shapes: results and time (HIP events, interleaved rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes
from bench_panel import timeit
D = torch.device("cuda:0")
torch.manual_seed(0)
for (M, K, what) in [(25600, 1024, "linear2 + LN"), (25600, 1024, "linear1 dgrad + R"), (25600, 768, "in-proj dgrad + R"), (25600, 512, "kv dgrad"),
                     (5760, 1024, "text linear2 + LN"), (5760, 1024, "text linear1 dgrad + R"), (64000, 1024, "c5 linear2 + LN")]:
    x = torch.randn(M, K, device=D); R = torch.randn(M, 256, device=D)
    z0 = torch.empty(M, 256, device=D); z1 = torch.empty_like(z0)
    if "LN" in what:
        W = torch.randn(256, K, device=D) * 0.03; b = torch.randn(256, device=D); gm = torch.rand(256, device=D) + 0.5; bt = torch.randn(256, device=D)
        pl = Planes([W])
        y0 = torch.empty_like(z0); m0 = torch.empty(M, device=D); r0 = torch.empty(M, device=D); y1 = torch.empty_like(z0); m1 = torch.empty(M, device=D); r1 = torch.empty(M, device=D)

        def f_old():
            ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, z0, 256, M, 256, K, bias=b, drop_p=0.1, seed=9, stream_id=2, R=R, ldr=256)
            ops.layernorm_fwd(z0, gm, bt, y0, m0, r0, 1e-5)
        f_gemm = lambda: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, z0, 256, M, 256, K, bias=b, drop_p=0.1, seed=9, stream_id=2, R=R, ldr=256)
        f_new = lambda: ops.panel_gemm(x, pl.ref(0), z1, 256, bias=b, R=R, drop_p=0.1, seed=9, stream_id=2, ln=(gm, bt, y1, m1, r1, 1e-5))
        f_old(); f_new(); torch.cuda.synchronize()
        msg = "z %.1e y %.1e" % (float((z0 - z1).abs().max()), float((y0 - y1).abs().max()))
        a, g1, n1, a2, n2 = timeit(f_old), timeit(f_gemm), timeit(f_new), timeit(f_old), timeit(f_new)
        print("%-24s M=%6d K=%4d  gemm+ln %.1f/%.1f us (gemm alone %.1f)  k-streamed panel %.1f/%.1f us   %s" % (what, M, K, a, a2, g1, n1, n2, msg), flush=True)
    else:
        W2 = torch.randn(K, 256, device=D) * 0.05
        pt = Planes([W2], transposed=True)
        Rr = R if "+ R" in what else None
        from unast_amd import config
        f_old = lambda: ops.gemm(ops.OP_KC, ops.OP_RC, x, K, W2, 256, z0, 256, M, 256, K, R=Rr, ldr=256 if Rr is not None else 0)
        f_new = lambda: ops.panel_gemm(x, pt.ref(0), z1, 256, R=Rr)
        f_old(); f_new(); torch.cuda.synchronize()
        msg = "max|diff| %.1e" % float((z0 - z1).abs().max())
        a, n1, a2, n2 = timeit(f_old), timeit(f_new), timeit(f_old), timeit(f_new)
        print("%-24s M=%6d K=%4d  tile (row-major W) %.1f/%.1f us  k-streamed panel %.1f/%.1f us   %s" % (what, M, K, a, a2, n1, n2, msg), flush=True)
