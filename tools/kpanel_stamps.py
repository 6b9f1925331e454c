"""Where a wave of the K-streamed panel GEMM spends its cycles (diagnostic build with s_memtime stamps: make -C unast_amd/csrc stamps).
Shares, not absolute run time: the stamps themselves cost cycles."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["UNAST_HIP_LIB"] = os.path.join(ROOT, "unast_amd", "libunast_hip_stamps.so")
sys.path.insert(0, ROOT)
import torch
from unast_amd import ops
from unast_amd._lib import lib
from unast_amd.planes import Planes
D = torch.device("cuda:0")
for M, K in ((25600, 1024), (25600, 512)):
    x = torch.randn(M, K, device=D); W = torch.randn(256, K, device=D) * 0.05; b = torch.randn(256, device=D); y = torch.empty(M, 256, device=D)
    R = torch.randn(M, 256, device=D)
    pl = Planes([W])
    nwg = (M + 127) // 128
    buf = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device=D)
    fn = lib().unast_panel_debug_stamps
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(buf.data_ptr())
    for _ in range(5):
        ops.panel_gemm(x, pl.ref(0), y, 256, bias=b, R=R)
    torch.cuda.synchronize()
    s = buf.view(nwg, 16, 8).double().cpu()
    names = ["wait A frags", "split -> W half 0", "wait vmcnt", "barrier", "issue reads + dma", "mfma h0 -> reads back", "total", "epilogue"]
    ng = K // 32
    print("M=%d K=%d: %d groups; mean cycles per wave (s_memtime ticks), per group in brackets; min / max over waves of the total: %.0f / %.0f" % (M, K, ng, float(s[:, :, 6].min()), float(s[:, :, 6].max())))
    for k in range(8):
        a = float(s[:, :, k].mean())
        print("  %-18s %9.0f   [%6.0f]" % (names[k], a, a / ng))
    fn(None)
