"""Random shapes and epilogues through both forms of unast_panel_gemm against the tile GEMM (bit for bit; LayerNorm epilogue to fp32 rounding).
usage: python tools/fuzz_panel.py [cases] [seed]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes, eligible
D = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
done = 0
while done < cases:
    deep = rnd.random() < 0.4
    M = rnd.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 300, 1000, 1127, 2049, 5760])
    if deep:
        N, K = rnd.choice([256, 512]), rnd.choice([320, 384, 512, 768, 1024, 1088])
    else:
        N, K = rnd.choice([1, 4, 46, 64, 81, 84, 128, 200, 256, 512, 768, 1000, 1024]), rnd.choice([4, 8, 32, 36, 64, 80, 96, 128, 192, 224, 256])
    if not eligible(N, K):
        continue
    torch.manual_seed(done)
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.05; b = torch.randn(N, device=D)
    ld = (N + 3) // 4 * 4
    pl = Planes([W])
    kinds = ["plain", "bias", "resid"] if deep else ["plain", "bias", "relu", "relu+drop", "resid"] + (["split"] if N % 4 == 0 else [])
    kind = rnd.choice(kinds)
    rows = 0 if deep else rnd.choice([64, 128, 1128])
    kw = dict(bias=b) if kind != "plain" else {}
    if kind == "relu": kw.update(act=1)
    if kind == "relu+drop": kw.update(act=1, drop_p=0.2, seed=done, stream_id=3)
    if kind == "split": kw.update(out_split=True)
    R = torch.randn(M, ld, device=D) if kind == "resid" else None
    y0 = torch.zeros(M, ld, device=D); y1 = torch.full((M + 3, ld), 7.0, device=D)
    ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y0, ld, M, N, K, R=R, ldr=ld if R is not None else 0, **kw)
    ops.panel_gemm(x, pl.ref(0), y1[:M], N, R=R, rows_per_wg=rows, **kw)
    torch.cuda.synchronize()
    ok = torch.equal(y0[:, :N].view(torch.int32), y1[:M, :N].view(torch.int32)) and bool((y1[M:] == 7.0).all()) and bool((y1[:M, N:] == 7.0).all())
    assert ok, (M, N, K, kind, rows, float((y0[:, :N] - y1[:M, :N]).abs().max()))
    if N == 256 and kind in ("bias", "resid", "plain"):            # LayerNorm epilogue
        gm = torch.rand(256, device=D) + 0.5; bt = torch.randn(256, device=D); Rr = torch.randn(M, 256, device=D)
        z0 = torch.empty(M, 256, device=D); yy0 = torch.empty_like(z0); m0 = torch.empty(M, device=D); r0 = torch.empty(M, device=D)
        z1 = torch.empty_like(z0); yy1 = torch.empty_like(z0); m1 = torch.empty(M, device=D); r1 = torch.empty(M, device=D)
        ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, z0, 256, M, 256, K, bias=b, drop_p=0.1, seed=9, stream_id=2, R=Rr, ldr=256)
        ops.layernorm_fwd(z0, gm, bt, yy0, m0, r0, 1e-5)
        ops.panel_gemm(x, pl.ref(0), z1, 256, bias=b, R=Rr, drop_p=0.1, seed=9, stream_id=2, ln=(gm, bt, yy1, m1, r1, 1e-5), rows_per_wg=rows)
        torch.cuda.synchronize()
        assert torch.equal(z0, z1) and float((yy0 - yy1).abs().max()) < 1e-5 and float((m0 - m1).abs().max()) < 1e-5, (M, K, "ln", rows)
    done += 1
print("fuzz panel ok: %d cases" % done)
