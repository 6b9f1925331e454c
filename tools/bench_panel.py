"""Row-panel GEMM (csrc/panel.hip) against the general tile GEMM on the train step's K <= 256 shapes: results and time
(HIP events, interleaved rounds in one process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes
D = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.manual_seed(0)
    shapes = [(25600, 1024, 256, "ffn1 relu+drop"), (25600, 768, 256, "qkv split"), (25600, 256, 256, "out-proj"), (25600, 512, 256, "kv"),
              (51200, 512, 256, "lstm xproj"), (5760, 1024, 256, "text ffn1"), (5760, 768, 256, "text qkv"), (5760, 256, 256, "text out"),
              (25600, 256, 80, "prenet fc1"), (25600, 84, 256, "heads 81"), (64000, 1024, 256, "c5 ffn1")]
    for (M, N, K, what) in shapes:
        x = torch.randn(M, K, device=D)
        W = torch.randn(N, K, device=D) * 0.05
        b = torch.randn(N, device=D)
        pl = Planes([W])
        y0 = torch.zeros(M, (N + 3) // 4 * 4, device=D)
        y1 = torch.zeros_like(y0)
        kw = dict(act=1, drop_p=0.1, seed=5, stream_id=3) if "relu" in what else {}
        sp = "split" in what
        f_old = lambda: ops.linear_fwd(x, W, b, y0[:, :N], out_split=sp, **kw)
        f_new = lambda rows=0: ops.panel_gemm(x, pl.ref(0), y1[:, :N] if N % 4 else y1, N, bias=b, out_split=sp, rows_per_wg=rows, **kw)
        f_old(); f_new(); torch.cuda.synchronize()
        if sp:
            err = float((y0.view(torch.int32) != y1.view(torch.int32)).float().mean())
            msg = "mismatching words %.2e" % err
        else:
            err = float((y0 - y1).abs().max()); ref = float(y0.abs().max())
            msg = "max|diff| %.2e of %.2e" % (err, ref)
        t_old = [timeit(f_old), 0, 0]
        t128 = timeit(lambda: f_new(128)); t64 = timeit(lambda: f_new(64)); tw = timeit(lambda: f_new(1128))
        t_old[1] = timeit(f_old)
        t128b = timeit(lambda: f_new(128)); t64b = timeit(lambda: f_new(64)); twb = timeit(lambda: f_new(1128))
        y1.zero_(); f_new(1128); torch.cuda.synchronize()
        errw = float((y0.view(torch.int32) != y1.view(torch.int32)).float().mean())
        m32 = ""
        if K == 256 and N % 64 == 0:                     # the 32x32x16 form (rows_per_wg = 2128): sums over k formed 16 at a time, so not bit-equal
            y1.zero_(); f_new(2128); torch.cuda.synchronize()
            if sp:
                e32 = "words off %.1e" % float((y0.view(torch.int32) != y1.view(torch.int32)).float().mean())
            else:
                e32 = "max|diff| %.1e" % float((y0 - y1).abs().max())
            t32 = timeit(lambda: f_new(2128)); tw2 = timeit(lambda: f_new(1128)); t32b = timeit(lambda: f_new(2128))
            m32 = "  | 32x32x16 %.1f/%.1f against %.1f (%s)" % (t32, t32b, tw2, e32)
        print("%-16s M=%6d N=%5d K=%4d  tile %.1f/%.1f us  panel128 %.1f/%.1f  panel64 %.1f/%.1f  panel128x16w %.1f/%.1f (mismatch %.1e)  %s%s" %
              (what, M, N, K, t_old[0], t_old[1], t128, t128b, t64, t64b, tw, twb, errw, msg, m32), flush=True)
    # LayerNorm epilogue against GEMM + unast_layernorm_fwd
    for M in (25600, 5760):
        x = torch.randn(M, 256, device=D); W = torch.randn(256, 256, device=D) * 0.05; b = torch.randn(256, device=D)
        R = torch.randn(M, 256, device=D); gm = torch.rand(256, device=D) + 0.5; bt = torch.randn(256, device=D)
        pl = Planes([W])
        z0 = torch.empty(M, 256, device=D); y0 = torch.empty_like(z0); m0 = torch.empty(M, device=D); r0 = torch.empty(M, device=D)
        z1 = torch.empty_like(z0); y1 = torch.empty_like(z0); m1 = torch.empty(M, device=D); r1 = torch.empty(M, device=D)

        def f_old():
            ops.linear_fwd(x, W, b, z0, drop_p=0.1, seed=9, stream_id=2, R=R)
            ops.layernorm_fwd(z0, gm, bt, y0, m0, r0, 1e-5)
        f_new = lambda rows=0: ops.panel_gemm(x, pl.ref(0), z1, 256, bias=b, R=R, drop_p=0.1, seed=9, stream_id=2, ln=(gm, bt, y1, m1, r1, 1e-5), rows_per_wg=rows)
        f_old(); f_new(); torch.cuda.synchronize()
        msg = "z %.2e y %.2e mean %.2e rstd %.2e" % (float((z0 - z1).abs().max()), float((y0 - y1).abs().max()), float((m0 - m1).abs().max()), float((r0 - r1).abs().max()))
        a = timeit(f_old); b128 = timeit(lambda: f_new(128)); b64 = timeit(lambda: f_new(64)); bw = timeit(lambda: f_new(1128)); a2 = timeit(f_old)
        y1.zero_(); f_new(1128); torch.cuda.synchronize()
        print("out-proj + LN    M=%6d  gemm+ln %.1f/%.1f us  panel128 %.1f  panel64 %.1f  panel128x16w %.1f (y %.2e)  %s" % (M, a, a2, b128, b64, bw, float((y0 - y1).abs().max()), msg), flush=True)


if __name__ == "__main__":
    main()
