"""Runs ONE kernel configuration a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
D = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "gemm_fwd"
if which == "gemm_fwd":
    M, N, K = 25600, 256, 1024
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D)
    for _ in range(10): ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K)
elif which == "gemm_small":
    M, N, K = 1280, 256, 1024
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D); y = torch.empty(M, N, device=D)
    for _ in range(10): ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K)
elif which == "attn_fwd":
    B, H, T, E = 32, 4, 800, 256
    qkv = torch.randn(B * T, 3 * E, device=D); O = torch.empty(B * T, E, device=D); LSE = torch.empty(B, H, T, device=D)
    lens = torch.full((B,), T, dtype=torch.int32, device=D)
    for _ in range(10): ops.attn_fwd(qkv[:, :E], qkv[:, E:2*E], qkv[:, 2*E:], O, LSE, lens, B, H, T, T, False, drop_p=0.1, seed=1, stream_id=1)
elif which == "attn_bwd":
    B, H, T, E = 32, 4, 800, 256
    qkv = torch.randn(B * T, 3 * E, device=D); O = torch.empty(B * T, E, device=D); LSE = torch.empty(B, H, T, device=D)
    dO = torch.randn(B * T, E, device=D); dqkv = torch.empty(B * T, 3 * E, device=D); delta = torch.empty(B, H, T, device=D)
    lens = torch.full((B,), T, dtype=torch.int32, device=D)
    ops.attn_fwd(qkv[:, :E], qkv[:, E:2*E], qkv[:, 2*E:], O, LSE, lens, B, H, T, T, False, drop_p=0.1, seed=1, stream_id=1)
    for _ in range(10):
        ops.attn_bwd(qkv[:, :E], qkv[:, E:2*E], qkv[:, 2*E:], O, dO, LSE, delta, dqkv[:, :E], dqkv[:, E:2*E], dqkv[:, 2*E:], lens, B, H, T, T, False,
                     drop_p=0.1, seed=1, stream_id=1)
elif which in ("panel_ffn1", "tile_ffn1"):
    from unast_amd.planes import Planes
    M, N, K = 25600, 1024, 256
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.05; b = torch.randn(N, device=D); y = torch.empty(M, N, device=D)
    pl = Planes([W])
    for _ in range(10):
        if which == "panel_ffn1": ops.panel_gemm(x, pl.ref(0), y, N, bias=b, act=1, drop_p=0.1, seed=5, stream_id=3, rows_per_wg=int(os.environ.get("PANEL_ROWS", "1128")))
        else: ops.gemm(ops.OP_KC, ops.OP_KC, x, K, W, K, y, N, M, N, K, bias=b, act=1, drop_p=0.1, seed=5, stream_id=3)
elif which == "kpanel_ffn2":
    from unast_amd.planes import Planes
    M, N, K = 25600, 256, 1024
    x = torch.randn(M, K, device=D); W = torch.randn(N, K, device=D) * 0.03; b = torch.randn(N, device=D); z = torch.empty(M, N, device=D); y = torch.empty(M, N, device=D)
    R = torch.randn(M, N, device=D); gm = torch.rand(N, device=D) + 0.5; bt = torch.randn(N, device=D); mean = torch.empty(M, device=D); rstd = torch.empty(M, device=D)
    pl = Planes([W])
    for _ in range(10):
        ops.panel_gemm(x, pl.ref(0), z, N, bias=b, R=R, drop_p=0.1, seed=9, stream_id=2, ln=(gm, bt, y, mean, rstd, 1e-5))
torch.cuda.synchronize()
