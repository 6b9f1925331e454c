"""One timing line for tools/panel_knockout.sh: the row-panel GEMM on three train-step shapes through whatever library UNAST_HIP_LIB names."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes
D = torch.device("cuda:0")
out = []
for (M, N, K, what) in [(25600, 1024, 256, "ffn1 relu+drop+bits"), (25600, 768, 256, "qkv split"), (25600, 256, 256, "out-proj")]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.05; b = torch.randn(N, device=D)
    pl = Planes([W]); y = torch.zeros(M, N, device=D)
    bits = torch.zeros(ops.gate_bits_bytes(M, N), dtype=torch.uint8, device=D) if "bits" in what else None
    kw = dict(act=1, drop_p=0.1, seed=5, stream_id=3, gate_bits=bits) if "relu" in what else {}
    fn = lambda: ops.panel_gemm(x, pl.ref(0), y, N, bias=b, out_split=("split" in what), rows_per_wg=1128, **kw)
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): fn()
    e1.record(); torch.cuda.synchronize()
    out.append("%s %.1f us" % (what, e0.elapsed_time(e1) / 40 * 1e3))
print("   ".join(out), flush=True)
