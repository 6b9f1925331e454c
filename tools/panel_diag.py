import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes
D = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, rows) in [(25600, 256, 256, 128), (25600, 256, 256, 64), (25600, 512, 256, 128), (256, 256, 256, 128), (25600, 1024, 256, 128)]:
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.05; b = torch.randn(N, device=D)
    pl = Planes([W]); y0 = torch.zeros(M, N, device=D); y1 = torch.full((M, N), -77.0, device=D)
    ops.linear_fwd(x, W, b, y0)
    for rep in range(3):
        y1.fill_(-77.0)
        ops.panel_gemm(x, pl.ref(0), y1, N, bias=b, rows_per_wg=rows)
        torch.cuda.synchronize()
        bad = (y0 != y1)
        print("M=%d N=%d rows=%d rep %d: bad %.4f  untouched %.4f" % (M, N, rows, rep, float(bad.float().mean()), float((y1 == -77.0).float().mean())))
        if bad.any():
            cols = bad.float().mean(0).view(-1, 16).mean(1)       # per 16-column tile
            rws = bad.float().mean(1)
            print("   by column tile:", [round(float(c), 3) for c in cols])
            print("   by row tile within 128-panel:", [round(float(v), 3) for v in rws.view(-1, 128 if M >= 128 else M)[:, :].mean(0).view(-1, 16).mean(1)])
            i = bad.nonzero()[0]
            print("   first bad", i.tolist(), float(y0[i[0], i[1]]), float(y1[i[0], i[1]]), " y1-bias:", float(y1[i[0], i[1]] - b[i[1]]), " y0-bias", float(y0[i[0], i[1]] - b[i[1]]))
