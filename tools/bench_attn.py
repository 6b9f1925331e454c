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
    from unast_amd._lib import lib
    for p in (0.0, 0.1):
        pairs = Tq * (Tq + 1) / 2 if causal else Tq * Tk
        fl = 4.0 * B * H * pairs * 64
        f = {}
        for m32 in (0, 1):          # pre-split operands, as the train step runs it
            old = lib().unast_attn_fwd_variant(m32)
            f[m32] = timeit(lambda: ops.attn_fwd(qkv_s[:, :E], kv_s[:, E:2*E], kv_s[:, 2*E:], O, LSE, lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1, qkv_split=True))
            lib().unast_attn_fwd_variant(old)
        b = {}
        for terms in (3, 2):
            config.ATTN_BWD_TERMS = terms
            b[terms] = timeit(lambda: ops.attn_bwd(qkv_s[:, :E], kv_s[:, E:2*E], kv_s[:, 2*E:], O, dO_s, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1, qkv_split=True))
        config.ATTN_BWD_TERMS = 3
        config.ATTN_FUSED_BWD = False
        b0 = timeit(lambda: ops.attn_bwd(qkv[:, :E], kv[:, E:2*E], kv[:, 2*E:], O, dO, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, Tq, Tk, causal, drop_p=p, seed=1, stream_id=1))
        config.ATTN_FUSED_BWD = True
        print((B, Tq, Tk, causal), "p=%.1f fwd 32x32x16 %.0f us (%.0f TF), 16x16x32 %.0f us | bwd one-pass %.0f us (%.0f TF), two-term %.0f us, two-kernel %.0f us" % (
            p, f[1], fl / f[1] / 1e6, f[0], b[3], 2.5 * fl / b[3] / 1e6, b[2], b0), flush=True)
