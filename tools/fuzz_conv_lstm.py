"""Random-shape fuzz of the implicit-GEMM Conv1d (forward with the BatchNorm statistics epilogue, input and weight gradients) against
torch's fp64 conv1d, and of the LSTM recurrences (ragged lengths down to 1) against torch.nn.LSTM on packed sequences.
usage: fuzz_conv_lstm.py [cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
D = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))


def relerr(a, b, floor=1e-6):
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(floor)).item()


worst = 0.0
for case in range(N):
    B, T = ri(1, 5), [1, 2, 3, 4, 5, 7, 16, 33, 127, 128, 129, 300][ri(0, 11)]
    Cin, Cout = [80, 256, 512, 84, 4][ri(0, 4)], [256, 512, 80, 84][ri(0, 3)]
    pad = [2, 4][ri(0, 1)]
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64); W = torch.randn(Cout, Cin, 5, generator=g, dtype=torch.float64) / (5 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    yr = torch.nn.functional.conv1d(torch.nn.functional.pad(xr.transpose(1, 2), (pad, 4 - pad)), Wr, b).transpose(1, 2)
    dy = torch.randn(B, T, Cout, generator=g, dtype=torch.float64)
    yr.backward(dy)
    Wp = W.permute(0, 2, 1).contiguous().float().to(D)
    y = torch.full((B, T, Cout), float("nan"), device=D); ws = torch.zeros(2 * Cout, dtype=torch.float64, device=D)
    ops.conv_fwd(x.float().to(D), Wp, b.float().to(D), y, pad, colstats=ws)
    dx = torch.full((B, T, Cin), float("nan"), device=D)
    ops.conv_dgrad(dy.float().to(D), Wp, dx, pad)
    dWp = torch.zeros(Cout, 5, Cin, device=D); db = torch.zeros(Cout, device=D)
    ops.conv_wgrad(dy.float().to(D), x.float().to(D), dWp, pad, db=db)
    y2 = yr.detach().reshape(-1, Cout)
    errs = [relerr(y, yr.detach()), relerr(dx, xr.grad), relerr(dWp, Wr.grad.permute(0, 2, 1), 1e-3), relerr(db, dy.sum((0, 1)), 1e-3),
            relerr(ws[:Cout], y2.sum(0), 1.0), relerr(ws[Cout:], (y2 * y2).sum(0), 1.0)]
    worst = max(worst, max(errs))
    print("conv case %2d B=%d T=%3d Cin=%3d Cout=%3d pad=%d  max err %.1e" % (case, B, T, Cin, Cout, pad, max(errs)), flush=True)
    assert max(errs) < 1e-4, errs

Hh = 64
for case in range(N // 2):
    Bd, T, Din = ri(1, 6), [1, 2, 5, 17, 40, 97][ri(0, 5)], 256
    lens = torch.randint(1, T + 1, (Bd,), generator=g); lens[ri(0, Bd - 1)] = T
    lstm = torch.nn.LSTM(Din, Hh, num_layers=1, bidirectional=True, batch_first=True).double()
    x = torch.randn(Bd, T, Din, generator=g, dtype=torch.float64, requires_grad=True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    out, (hn, cn) = lstm(packed)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    dy = torch.randn(Bd, T, 2 * Hh, generator=g, dtype=torch.float64)
    for bi in range(Bd):
        dy[bi, lens[bi]:] = 0
    dh = torch.randn(2, Bd, Hh, generator=g, dtype=torch.float64)
    ((yr * dy).sum() + (hn * dh).sum()).backward()
    wih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]).detach(); whh = torch.cat([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]).detach().float().to(D).contiguous()
    bih = torch.cat([lstm.bias_ih_l0, lstm.bias_ih_l0_reverse]).detach().float().to(D); bhh = torch.cat([lstm.bias_hh_l0, lstm.bias_hh_l0_reverse]).detach().float().to(D)
    xproj = (x.detach() @ wih.t()).float().to(D).contiguous()                     # [Bd,T,2*4*Hh]
    lens_d = lens.to(torch.int32).to(D)
    y = torch.zeros(Bd, T, 2 * Hh, device=D); gates = torch.empty(Bd, T, 2, 4 * Hh, device=D); cs = torch.empty(Bd, T, 2, Hh, device=D)
    hprev = torch.zeros(Bd, T, 2, Hh, device=D); hfin = torch.empty(Bd, 2 * Hh, device=D)
    ops.lstm_fwd(xproj, whh, bih, bhh, lens_d, y, gates, cs, hprev, hfin, 2, 4 * Hh * Hh, 4 * Hh)
    dg = torch.zeros(Bd, T, 2, 4 * Hh, device=D)
    dhf = torch.cat([dh[0], dh[1]], 1).float().to(D).contiguous()
    ops.lstm_bwd(dy.float().to(D), dhf, whh, gates, cs, lens_d, dg, 2, 4 * Hh * Hh)
    dx = (dg.view(Bd, T, 2 * 4 * Hh).double().cpu() @ wih)                        # dX through the input projection
    errs = [relerr(y, yr.detach(), 1e-2), relerr(hfin, torch.cat([hn[0], hn[1]], 1).detach(), 1e-2), relerr(dx, x.grad, 1e-2)]
    worst = max(worst, max(errs))
    print("lstm case %2d Bd=%d T=%2d lens=%s  max err %.1e" % (case, Bd, T, lens.tolist(), max(errs)), flush=True)
    assert max(errs) < 1e-4, errs
print("fuzz ok, worst %.1e" % worst)
