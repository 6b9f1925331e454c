"""Block dispatch order of the attention kernels (UNAST_ATTN_ORDER, csrc/attention.hip xcd_remap): 0 = the blocks of a (batch, head) back to
back, 1 = position by position over all (batch, head) of an XCD (the short last block of every head at the end of the launch).  Runs itself
once per order and prints forward / one-pass-backward times at the train step's shapes (pre-split operands, dropout 0.1).  GPU box."""
import os, subprocess, sys
if os.environ.get("UNAST_ATTN_ORDER") is None:
    for o in ("0", "1", "0", "1"):
        print("order %s:" % o, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, UNAST_ATTN_ORDER=o))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unast_amd import ops
D = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for (B, Tq, Tk, causal, ragged) in [(64, 800, 800, 0, 0), (64, 800, 800, 0, 1), (64, 800, 800, 1, 1), (32, 800, 800, 0, 1), (32, 800, 180, 0, 1), (32, 180, 800, 0, 1), (64, 2000, 2000, 1, 1)]:
    H, E = 4, 256
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * Tq, 3 * E, generator=g).to(D); kv = torch.randn(B * Tk, 3 * E, generator=g).to(D) if Tk != Tq else qkv
    O = torch.empty(B * Tq, E, device=D); LSE = torch.empty(B, H, Tq, device=D)
    lens = (torch.randint(Tk // 2, Tk + 1, (B,), generator=g) if ragged else torch.full((B,), Tk)).to(torch.int32).to(D)
    dO = torch.randn(B * Tq, E, generator=g).to(D); ws = torch.empty(B, H, Tq, device=D); dQ = torch.empty(B * Tq, E, device=D); dKV = torch.empty(B * Tk, 2 * E, device=D)
    qkv_s, kv_s, dO_s = torch.empty_like(qkv), torch.empty_like(kv), torch.empty_like(dO)
    ops.split_f32(qkv.view(-1), qkv_s.view(-1)); ops.split_f32(kv.view(-1), kv_s.view(-1)); ops.split_f32(dO.view(-1), dO_s.view(-1))
    if Tk == Tq:
        kv_s = qkv_s
    f = timeit(lambda: ops.attn_fwd(qkv_s[:, :E], kv_s[:, E:2*E], kv_s[:, 2*E:], O, LSE, lens, B, H, Tq, Tk, causal, drop_p=0.1, seed=1, stream_id=1, qkv_split=True))
    b = timeit(lambda: ops.attn_bwd(qkv_s[:, :E], kv_s[:, E:2*E], kv_s[:, 2*E:], O, dO_s, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens, B, H, Tq, Tk, causal, drop_p=0.1, seed=1, stream_id=1, qkv_split=True))
    out.append("   B=%d %dx%d%s%s: fwd %.0f us, bwd %.0f us" % (B, Tq, Tk, " causal" if causal else "", " ragged" if ragged else "", f, b))
print("\n".join(out), flush=True)
