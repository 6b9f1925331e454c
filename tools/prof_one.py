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
torch.cuda.synchronize()
