"""Knock-out timing of the row-panel GEMM (wrong results, timing only): which part of the loop the time is in."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes
from bench_panel import timeit
D = torch.device("cuda:0")
for (M, N, K) in [(25600, 1024, 256), (25600, 256, 256)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.05; b = torch.randn(N, device=D)
    pl = Planes([W]); y = torch.zeros(M, N, device=D)
    for name, dbg in [("full", 0), ("no store", 1), ("no mfma", 2), ("no mfma, no store", 3), ("no lds read+mfma", 6), ("no lds+mfma+store", 7), ("no dma", 8), ("no dma, no store", 9),
                      ("no barrier", 16), ("only dma+barrier", 7), ("nothing (prologue only)", 31), ("no dma no barrier", 24)]:
        f = lambda: ops.panel_gemm(x, pl.ref(0), y, N, bias=b, rows_per_wg=128 | (dbg << 8))
        print("M=%d N=%d  %-28s %.1f us" % (M, N, name, timeit(f)), flush=True)
