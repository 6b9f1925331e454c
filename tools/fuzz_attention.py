"""Random-shape fuzz of the attention kernels (forward + one-pass backward, fp32 and pre-split operands) against fp64 math:
short and length-1 key sets, single queries, lengths that straddle tile edges, causal and not.  usage: fuzz_attention.py [cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_kernels import naive_attention


def relerr(a, b, floor=1.0):
    """max |a - b| over max |b|, with a floor of 1 on the denominator (operands are N(0,1)): when every query sees a single key the exact gradients of Q and K are zero and what is left is the split products' 1e-5 absolute error."""
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(floor)).item()

from unast_amd import ops
D = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
H, E = 4, 256
edges = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 300]
worst = 0.0
for case in range(N):
    B = int(torch.randint(1, 4, (1,), generator=g))
    Tq = edges[int(torch.randint(0, len(edges), (1,), generator=g))]
    causal = bool(torch.randint(0, 2, (1,), generator=g))
    Tk = Tq if causal else edges[int(torch.randint(0, len(edges), (1,), generator=g))]
    lens = torch.randint(1, Tk + 1, (B,), generator=g)
    if case % 3 == 0:
        lens[0] = 1
    split = bool(case % 2)
    q = torch.randn(B, Tq, E, generator=g, dtype=torch.float64).requires_grad_(True)
    k = torch.randn(B, Tk, E, generator=g, dtype=torch.float64).requires_grad_(True)
    v = torch.randn(B, Tk, E, generator=g, dtype=torch.float64).requires_grad_(True)
    o_ref, lse_ref = naive_attention(q, k, v, lens, causal, H)
    do = torch.randn(B, Tq, E, generator=g, dtype=torch.float64)
    o_ref.backward(do)
    dev = lambda t: t.detach().float().to(D).contiguous()
    qd, kd, vd, dod = dev(q).view(B * Tq, E), dev(k).view(B * Tk, E), dev(v).view(B * Tk, E), dev(do).view(B * Tq, E)
    if split:
        def sp(x):
            y = torch.empty_like(x); ops.split_f32(x.view(-1), y.view(-1)); return y
        qd, kd, vd, dod_in = sp(qd), sp(kd), sp(vd), sp(dod)
    else:
        dod_in = dod
    O = torch.full((B * Tq, E), float("nan"), device=D); LSE = torch.full((B, H, Tq), float("nan"), device=D)
    lens_d = lens.to(torch.int32).to(D)
    ops.attn_fwd(qd, kd, vd, O, LSE, lens_d, B, H, Tq, Tk, causal, qkv_split=split)
    dQ = torch.full((B * Tq, E), float("nan"), device=D); dKV = torch.full((B * Tk, 2 * E), float("nan"), device=D)
    ws = torch.empty(B, H, Tq, device=D)
    ops.attn_bwd(qd, kd, vd, O, dod_in, LSE, ws, dQ, dKV[:, :E], dKV[:, E:], lens_d, B, H, Tq, Tk, causal, qkv_split=split)
    tol = 2e-4 if split else 6e-5          # pre-split operands carry 16 mantissa bits
    fl = max(1.0, Tq ** 0.5)          # dK / dV sum over the queries: their natural scale with N(0,1) operands
    errs = [relerr(O.view(B, Tq, E), o_ref.detach()), relerr(LSE, lse_ref.detach()), relerr(dQ.view(B, Tq, E), q.grad),
            relerr(dKV[:, :E].reshape(B, Tk, E), k.grad, fl), relerr(dKV[:, E:].reshape(B, Tk, E), v.grad, fl)]
    bad = [e for e in errs if not (e < tol)]
    worst = max(worst, max(errs))
    print("case %2d B=%d Tq=%3d Tk=%3d causal=%d lens=%s split=%d  max err %.1e%s" % (case, B, Tq, Tk, causal, lens.tolist(), split, max(errs), "   <-- FAIL" if bad else ""), flush=True)
    assert not bad, errs
print("fuzz ok, worst %.1e" % worst)
