import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from unast_amd import ops
D = torch.device("cuda:0")
Bd, T, Hh = 64, 800, 64
g = torch.Generator().manual_seed(0)
lens = torch.full((Bd,), T, dtype=torch.int32, device=D)
whh = (torch.randn(2 * 4 * Hh, Hh, generator=g) * 0.1).to(D); bih = torch.zeros(2 * 4 * Hh, device=D); bhh = torch.zeros(2 * 4 * Hh, device=D)
xproj = torch.randn(Bd, T, 2 * 4 * Hh, generator=g).to(D)
y = torch.zeros(Bd, T, 2 * Hh, device=D); gates = torch.empty(Bd, T, 2, 4 * Hh, device=D); cs = torch.empty(Bd, T, 2, Hh, device=D)
hprev = torch.zeros(Bd, T, 2, Hh, device=D); hfin = torch.empty(Bd, 2 * Hh, device=D)
dy = torch.randn(Bd, T, 2 * Hh, generator=g).to(D); dhf = torch.randn(Bd, 2 * Hh, generator=g).to(D); dg = torch.zeros(Bd, T, 2, 4 * Hh, device=D)
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f = timeit(lambda: ops.lstm_fwd(xproj, whh, bih, bhh, lens, y, gates, cs, hprev, hfin, 2, 4 * Hh * Hh, 4 * Hh))
b = timeit(lambda: ops.lstm_bwd(dy, dhf, whh, gates, cs, lens, dg, 2, 4 * Hh * Hh))
print("lstm fwd %.0f us (%.3f us/step)  bwd %.0f us (%.3f us/step)" % (f, f / T, b, b / T))
