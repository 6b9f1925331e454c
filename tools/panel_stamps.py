"""Where a wave of the row-panel GEMM spends its cycles (diagnostic build with s_memtime stamps, -DPANEL_STAMPS;
build: see tools/build_panel_stamps.sh).  Shares, not absolute run time: the stamps themselves cost cycles."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["UNAST_HIP_LIB"] = os.path.join(ROOT, "unast_amd", "libunast_hip_stamps.so")
sys.path.insert(0, ROOT)
import torch
from unast_amd import ops
from unast_amd._lib import lib
from unast_amd.planes import Planes
D = torch.device("cuda:0")
shapes = [(25600, 1024, 256, dict(act=1, drop_p=0.1, seed=5, stream_id=3)), (25600, 768, 256, dict(out_split=True)), (25600, 256, 256, {})]
for M, N, K, kw in shapes:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.05; b = torch.randn(N, device=D); y = torch.empty(M, N, device=D)
    pl = Planes([W])
    nwg = (M + 127) // 128
    buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=D)
    fn = lib().unast_panel_debug_stamps
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(buf.data_ptr())
    for _ in range(5):
        ops.panel_gemm(x, pl.ref(0), y, N, bias=b, rows_per_wg=128, **kw)
    torch.cuda.synchronize()
    s = buf.view(nwg, 8, 8).double().cpu()
    names = ["wait vmcnt", "barrier", "dma issue", "epilogue", "mfma phase", "prologue", "total", "-"]
    npieces = (N + 63) // 64
    print("M=%d N=%d K=%d: %d groups; mean cycles per wave (waves 0-3 | waves 4-7), per group in brackets" % (M, N, K, npieces))
    for k in range(7):
        a, c = float(s[:, :4, k].mean()), float(s[:, 4:, k].mean())
        print("  %-12s %9.0f | %9.0f   [%6.0f | %6.0f]" % (names[k], a, c, a / npieces, c / npieces))
    fn(None)
