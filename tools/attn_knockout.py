"""One timing line for tools/attn_knockout.sh: forward attention at config-3 size through whatever library UNAST_HIP_LIB names."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
D = torch.device("cuda:0")
B, H, E, Tq, Tk = 32, 4, 256, 800, 800
qkv = torch.randn(B * Tq, 3 * E, device=D); qs = torch.empty_like(qkv); ops.split_f32(qkv.view(-1), qs.view(-1))
O = torch.empty(B * Tq, E, device=D); LSE = torch.empty(B, H, Tq, device=D); lens = torch.full((B,), Tk, dtype=torch.int32, device=D)
res = []
for p in (0.1, 0.0):
    fn = lambda: ops.attn_fwd(qs[:, :E], qs[:, E:2*E], qs[:, 2*E:], O, LSE, lens, B, H, Tq, Tk, 0, drop_p=p, seed=1, stream_id=1, qkv_split=True)
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50 * 1e3)
print("fwd 800x800  p=0.1 %.1f us   p=0 %.1f us" % tuple(res), flush=True)
