"""Micro-benchmark of the attention kernels at the train step's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
D = torch.device("cuda:0")

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, Tq, Tk, causal) in [(32, 800, 800, 0), (32, 800, 800, 1), (32, 180, 180, 0), (32, 800, 180, 0), (32, 180, 800, 0), (32, 2000, 2000, 1)]:
    H, E = 4, 256
    qkv = torch.randn(B * Tq, 3 * E, device=D); kv = torch.randn(B * Tk, 3 * E, device=D) if Tk != Tq else qkv
    O = torch.empty(B * Tq, E, device=D); LSE = torch.empty(B, H, Tq, device=D); lens = torch.full((B,), Tk, dtype=torch.int32, device=D)
    dO = torch.randn(B * Tq, E, device=D); ws = torch.empty(B, H, Tq, device=D); dQ = torch.empty(B * Tq, E, device=D); dKV = torch.empty(B * Tk, 2 * E, device=D)
    from unast_amd import config
    qkv_s, kv_s, dO_s = torch.empty_like(qkv), torch.empty_like(kv), torch.empty_like(dO)
    ops.split_f32(qkv.view(-1), qkv_s.view(-1)); ops.split_f32(kv.view(-1), kv_s.view(-1)); ops.split_f32(dO.view(-1), dO_s.view(-1))
    if Tk == Tq:
        kv_s = qkv_s
    for p in (0.0, 0.1):
        fs = timeit(lambda: ops.attn_fwd(qkv_s[:, :E], kv_s[:, E:2*E], kv_s[:, 2*E:], O, LSE, lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1, qkv_split=True))
        bs = timeit(lambda: ops.attn_bwd(qkv_s[:, :E], kv_s[:, E:2*E], kv_s[:, 2*E:], O, dO_s, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1, qkv_split=True))
        config.ATTN_FUSED_BWD = False
        b0 = timeit(lambda: ops.attn_bwd(qkv[:, :E], kv[:, E:2*E], kv[:, 2*E:], O, dO, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1))
        config.ATTN_FUSED_BWD = True
        f = timeit(lambda: ops.attn_fwd(qkv[:, :E], kv[:, E:2*E], kv[:, 2*E:], O, LSE, lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1))
        b = timeit(lambda: ops.attn_bwd(qkv[:, :E], kv[:, E:2*E], kv[:, 2*E:], O, dO, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1))
        pairs = Tq * (Tq + 1) / 2 if causal else Tq * Tk
        fl = 4.0 * B * H * pairs * 64
        print((B, Tq, Tk, causal), "p=%.1f fwd %.0f us (%.0f TF; pre-split operands %.0f us)  bwd fused %.0f us (%.0f TF; pre-split %.0f us)  bwd two-kernel %.0f us" % (p, f, fl / f / 1e6, fs, b, 2.5 * fl / b / 1e6, bs, b0), flush=True)
