"""Out-of-bounds WRITE detector for the row-panel GEMM: every operand is carved out of one arena with sentinel-filled guard zones around it,
so a stray store corrupts a guard instead of faulting."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops
from unast_amd.planes import Planes
D = torch.device("cuda:0")
torch.manual_seed(0)
arena = torch.empty(96 << 20, dtype=torch.uint8, device=D)
GUARD = 1 << 20


def carve(off, nbytes, dtype, shape):
    t = arena[off:off + nbytes].view(dtype).view(shape)
    return t, off + (nbytes + 255) // 256 * 256


for (M, N, K, kw, rows) in [(144, 1024, 256, dict(act=1, bits=True), 1128), (144, 1024, 256, dict(act=1, bits=True), 128), (60, 1024, 256, dict(act=1, bits=True), 1128),
                            (144, 1024, 256, dict(act=1, drop_p=0.2, seed=3, stream_id=2, bits=True), 1128), (144, 256, 256, dict(ln=True), 1128), (144, 768, 256, dict(out_split=True), 1128),
                            (144, 256, 80, dict(act=1), 1128), (144, 81, 256, dict(), 1128), (144, 1024, 256, dict(rgate=True), 1128)]:
    W = torch.randn(N, K, device=D) * 0.05
    pl = Planes([W])
    for trial in range(6):
        arena.fill_(0x5A)
        start = (8 << 20) + trial * 4096 * 3
        off = start
        x, off = carve(off, M * K * 4, torch.float32, (M, K)); g0 = off; off += GUARD
        ldc = (N + 3) // 4 * 4
        y, off = carve(off, M * ldc * 4, torch.float32, (M, ldc)); g1 = off; off += GUARD
        nb = ops.gate_bits_bytes(M, N)
        bits, off = carve(off, nb, torch.uint8, (nb,)); g2 = off; off += GUARD
        y2, off = carve(off, M * 256 * 4, torch.float32, (M, 256)); g3 = off; off += GUARD
        st, off = carve(off, 2 * M * 4, torch.float32, (2, M)); g4 = off; off += GUARD
        x.copy_(torch.randn(M, K, device=D)); b = torch.randn(N, device=D)
        k2 = dict(kw)
        if k2.pop("bits", False): k2["gate_bits"] = bits
        if k2.pop("rgate", False): k2["gate_bits"] = bits; k2["gate_scale"] = 1.1; b = None; bits.random_(0, 255)
        if k2.pop("ln", False):
            R = torch.randn(M, 256, device=D); gm = torch.ones(256, device=D); bt = torch.zeros(256, device=D)
            k2["ln"] = (gm, bt, y2, st[0], st[1], 1e-5); k2["R"] = R
        ops.panel_gemm(x, pl.ref(0), y[:, :N] if N != ldc else y, N, bias=b, rows_per_wg=rows, **k2)
        torch.cuda.synchronize()
        bad = []
        for name, gs in (("after A", g0), ("after C", g1), ("after bits", g2), ("after Y", g3), ("after stats", g4)):
            z = arena[gs:gs + GUARD]
            if not bool((z == 0x5A).all()):
                idx = (z != 0x5A).nonzero()
                bad.append("%s: %d bytes, first at +%d last at +%d" % (name, idx.numel(), int(idx[0]), int(idx[-1])))
        head = arena[start - GUARD:start]
        if not bool((head == 0x5A).all()): bad.append("before A corrupted")
        print("M=%d N=%d K=%d %s rows=%d trial %d: %s" % (M, N, K, sorted(kw), rows, trial, "; ".join(bad) if bad else "guards intact"), flush=True)
