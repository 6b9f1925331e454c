"""Tiled bf16 planes of weight matrices: the B-operand format of the row-panel GEMM (csrc/panel.hip, unast_panel_gemm).

A destination matrix Wd[n][k] (n < N output columns of the GEMM, k < K contraction index; K <= 256 for the activation-stationary form
of the kernel, a multiple of 64 beyond that for its K-streamed form) is stored as a hi plane and a
lo plane of bf16 (hi = RNE(x), lo = RNE(x - hi)), each cut into 1-KB sub-tiles of 16 n x 32 k, padded with zeros to multiples of
64 n and 32 k.  `plan` lays several matrices out in one buffer and returns the block descriptors unast_retile_weights consumes
(include/unast_hip.h); no arithmetic happens here.
"""
import numpy as np
import torch

DESC_DTYPE = np.dtype([("src_off", "<i4"), ("ld", "<i4"), ("transposed", "<i4"), ("N", "<i4"), ("K", "<i4"), ("n0", "<i4"), ("k0", "<i4"),
                       ("ksteps", "<i4"), ("dst_off", "<i8"), ("plane_bytes", "<i8")])
assert DESC_DTYPE.itemsize == 48
MAX_K = 256
KSTEPS_BUILT = (1, 2, 3, 4, 6, 8)          # template instantiations of panel_kernel


def geometry(N, K):
    """(ksteps, plane_bytes) of a destination matrix Wd[N][K]."""
    ksteps = (K + 31) // 32
    n64 = (N + 63) // 64 * 64
    return ksteps, n64 * ksteps * 64


def eligible(N, K):
    if K > MAX_K:                              # the K-streamed form of the kernel: 64-deep groups of K into 256-column output tiles
        return K % 64 == 0 and K <= 8192 and N % 256 == 0
    if 224 < K < 256:                          # the 8-k-step form takes its activation rows by LDS-DMA as whole 1-KB rows: K = 256 exactly
        return False
    return 4 <= K <= MAX_K and K % 4 == 0 and (K + 31) // 32 in KSTEPS_BUILT and N >= 1


def plan(mats):
    """mats: [(src_off_floats, src_row_stride, transposed, N, K)] -> (descs ndarray, [(dst_off, plane_bytes, ksteps)], total bytes)."""
    descs, placed, off = [], [], 0
    for src_off, ld, tr, N, K in mats:
        ksteps, pb = geometry(N, K)
        placed.append((off, pb, ksteps))
        for n0 in range(0, (N + 63) // 64 * 64, 64):
            for k0 in range(0, ksteps * 32, 64):
                descs.append((src_off, ld, int(tr), N, K, n0, k0, ksteps, off, pb))
        off += 2 * pb
    return np.array(descs, dtype=DESC_DTYPE), placed, off


def descs_to_device(descs, device):
    return torch.from_numpy(descs.view(np.uint8).reshape(-1).copy()).to(device)


class Planes:
    """Planes of a few stand-alone weight tensors (tests, micro-benchmarks).  The train step uses engine.FlatStore's."""

    def __init__(self, weights, transposed=False):
        from . import ops
        dev = weights[0].device
        flat = torch.cat([w.reshape(-1) for w in weights])
        mats, o = [], 0
        for w in weights:
            rows, cols = w.shape
            mats.append((o, cols, int(transposed), cols if transposed else rows, rows if transposed else cols))
            o += w.numel()
        descs, self.placed, total = plan(mats)
        self.buf = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.descs = descs_to_device(descs, dev)
        self.src = flat
        ops.retile_weights(flat, self.buf, self.descs, len(descs))

    def ref(self, i, row0=0):
        """(hi-plane pointer, plane bytes) of matrix i, optionally from output column `row0` (a multiple of 64) on."""
        off, pb, ksteps = self.placed[i]
        assert row0 % 64 == 0
        return self.buf.data_ptr() + off + (row0 // 16) * ksteps * 1024, pb
