"""Thin Python wrappers over the C ABI (include/unast_hip.h): pointer marshalling only, no arithmetic.

Every function enqueues HIP kernels on torch's current stream.  Tensors are fp32 CUDA tensors owned by torch
(device-memory plumbing); shapes are validated here and again on the C side.
"""
import torch

from . import config
from ._lib import lib, check

OP_KC, OP_KC_CONV, OP_RC, OP_RC_CONV_DGRAD, OP_RC_CONV_WGRAD = 0, 1, 2, 3, 4


def _p(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (raw getter: ~0.3 us instead of ~9 us)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_F32 = torch.float32
_LN_WS = {}


def deterministic_sums():
    """True with config.DETERMINISTIC_SUMS (utils.set_deterministic(True, fixed_sums=True), or UNAST_DETERMINISTIC_SUMS=1): every fp32 sum
    of the step is then formed in a fixed order -- the two-kernel attention backward instead of the one-pass kernel's dQ atomics, bias
    gradients by unast_colsum_det instead of the weight-gradient GEMM's row-sum atomics, unast_embed_bwd_det, ungrouped weight gradients --
    so that two runs of a step, eager or replayed, agree to the bit.  (fp64 accumulators -- BatchNorm statistics, loss sums, the gradient
    norm -- stay atomic: their order only moves bits far below fp32 resolution.)  Off by default, also in the parity mode: the golden and
    oracle tests are to pin the kernels the train step runs with."""
    return bool(config.DETERMINISTIC_SUMS)


def _f32(t, name):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 CUDA tensor (got %s on %s)" % (name, t.dtype, t.device))


# Flat parameter stores whose weights also exist in the pre-split operand format: (base address, bytes, split base).
_WEIGHT_SPANS = []


def register_weight_span(base_ptr, nbytes, split_ptr, transposed_lookup=None, planes_lookup=None):
    _WEIGHT_SPANS[:] = [s for s in _WEIGHT_SPANS if s[0] != base_ptr]
    _WEIGHT_SPANS.append((base_ptr, nbytes, split_ptr, transposed_lookup, planes_lookup))


def unregister_weight_span(base_ptr):
    _WEIGHT_SPANS[:] = [s for s in _WEIGHT_SPANS if s[0] != base_ptr]


def _weight_planes(W, transposed=False):
    """(hi-plane pointer, plane bytes) of the tiled bf16 planes of a stored weight view (engine.FlatStore.planes_of), or None."""
    ptr = W.data_ptr()
    for base, nbytes, split, _, lookup in _WEIGHT_SPANS:
        if lookup is not None and base <= ptr < base + nbytes:
            return lookup(W, transposed)
    return None


def _presplit_ptr(ptr):
    for base, nbytes, split, _, _p in _WEIGHT_SPANS:
        if base <= ptr < base + nbytes:
            return split + (ptr - base)
    return 0


def _transposed_weight(W):
    """(pointer, row stride) of W^T in the pre-split format if W is a stored weight that has one (engine.FlatStore.dgrad_T)."""
    ptr = W.data_ptr()
    for base, nbytes, split, lookup, _p in _WEIGHT_SPANS:
        if lookup is not None and base <= ptr < base + nbytes:
            return lookup(W)
    return None


def gemm(a_mode, b_mode, A, lda, B, ldb, C, ldc, M, N, K, conv=(0, 0, 0, 0), bias=None, R=None, ldr=0, G=None,
         ldg=0, gate_scale=1.0, alpha=1.0, beta=0, act=0, drop_p=0.0, seed=0, stream_id=0, splitk=1, nsplit=None, atomic=False, rowsum_a=None, kb_valid=0, tile_wn=0,
         b_ptr=None, out_split=False, colstats=None):
    """out_split: C is written in the pre-split operand format (for the attention kernels).  b_ptr: B given as a raw device pointer to a PRE-SPLIT operand (a transposed weight copy); `B` is then only a shape/dtype witness."""
    if A.dtype is not _F32 or B.dtype is not _F32 or C.dtype is not _F32 or not C.is_cuda:       # (epilogue operands are produced by this package)
        for t, n in ((A, "A"), (B, "B"), (C, "C"), (bias, "bias"), (R, "R"), (G, "G")):
            _f32(t, n)
    ws, ws_n = None, 0
    if splitk > 1 and not atomic:
        ws_n = splitk * M * ((N + 3) // 4 * 4)
        ws = torch.empty(ws_n, dtype=torch.float32, device=C.device)          # split-K partial slabs (caching allocator)
    bp, presplit = _p(B), 0
    if b_ptr is not None:
        bp, presplit = b_ptr, 1
    elif _WEIGHT_SPANS and config.PRESPLIT_WEIGHTS and a_mode != OP_RC:      # forward / dgrad forms: B may be a stored weight
        sp = _presplit_ptr(bp)
        if sp:
            bp, presplit = sp, 1
    check(lib().unast_gemm(a_mode, b_mode, nsplit or config.NSPLIT, _p(A), lda, bp, ldb, _p(C), ldc, M, N, K, kb_valid,
                           conv[0], conv[1], conv[2], conv[3], _p(bias), _p(R), ldr, _p(G), ldg, gate_scale,
                           alpha, beta, act, drop_p, seed & 0xFFFFFFFF, stream_id, splitk, _p(ws), ws_n, _p(rowsum_a), tile_wn, presplit,
                           int(out_split), _p(colstats), _stream()), "unast_gemm")


import os as _os
SPLITK_TARGET_BLOCKS = int(_os.environ.get("UNAST_SPLITK_TARGET", "256"))      # one workgroup per CU: 36.4 vs 36.8 ms/step at 320 (same box)
SPLITK_MIN_KSTEPS = int(_os.environ.get("UNAST_SPLITK_MINK", "10"))


def _splitk_for(M, N, K):
    """Split the long token reduction of a weight gradient so that ~SPLITK_TARGET_BLOCKS workgroups are in flight, each
    with at least SPLITK_MIN_KSTEPS 32-deep k-steps."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ksteps = (K + 31) // 32
    # every split writes and re-reads an [M,N] slab: for a small gradient (<= 4 tiles) one workgroup per CU is the optimum, a
    # larger one amortises the slabs over more operand bytes and gains from two per CU (round-1 tools/bench_wgrad.py: 768x256 over
    # 25600 tokens 76 -> 57 us at 42 instead of 21 splits; 256x256 39.7 us at 64 splits, 53.7 at 128)
    target = SPLITK_TARGET_BLOCKS * (2 if tiles >= 8 else 1)
    want = max(1, target // tiles)
    return max(1, min(want, ksteps // SPLITK_MIN_KSTEPS))


def _panel_ok(M, K, *tensors, N=None):
    if not (config.PANEL_GEMM and config.NSPLIT == 3 and M >= config.PANEL_MIN_ROWS and K >= 4 and K % 4 == 0):
        return False
    if K > 256 and not (config.PANEL_KSTREAM and K % 64 == 0 and N is not None and N % 256 == 0):      # the K-streamed form: 256-column output tiles
        return False
    for t in tensors:
        if t is not None and (t.stride(-1) != 1 or t.data_ptr() % 16 or t.stride(0) % 4):
            return False
        if t is not None and (M + 127) // 128 * 128 * t.stride(0) * 4 >= 0x7FFFFFF0:      # 32-bit buffer offsets (unast_panel_gemm refuses it too)
            return False
    return True


def linear_fwd(x2d, W, bias, out, act=0, drop_p=0.0, seed=0, stream_id=0, R=None, out_split=False, ln=None, gate_bits=None):
    """out[M,N] = epi(x2d[M,K] @ W[N,K]^T + bias).  ln = (gamma, beta, y, mean, rstd, eps): LayerNorm of `out` fused behind the GEMM when
    the row-panel kernel serves the shape (returns True), a separate launch otherwise.  gate_bits: uint8 buffer that receives one
    keep bit per output element (panel kernel only; see linear_dgrad)."""
    M, K = x2d.shape
    N = W.shape[0]
    fuse_ln = ln is not None and config.PANEL_LN and N == 256 and not out_split and act == 0
    deep = K > 256                                  # K-streamed form: plain / residual / LayerNorm epilogues (dropout inside the LayerNorm one only)
    if _panel_ok(M, K, x2d, out, R, N=N) and W.stride(1) == 1 and (R is None or fuse_ln or deep) and \
            not (deep and (act or out_split or gate_bits is not None or (drop_p > 0 and not fuse_ln))):       # (K <= 256: a residual operand is served by the LayerNorm epilogue only)
        wp = _weight_planes(W)
        if wp is not None:
            panel_gemm(x2d, wp, out, N, bias=bias, R=R, act=act, drop_p=drop_p, seed=seed, stream_id=stream_id, out_split=out_split,
                       ln=ln if fuse_ln else None, rows_per_wg=config.PANEL_ROWS, gate_bits=gate_bits)
            if ln is not None and not fuse_ln:
                layernorm_fwd(out, ln[0], ln[1], ln[2], ln[3], ln[4], ln[5])
            return out
    if gate_bits is not None:
        raise RuntimeError("linear_fwd: gate_bits needs the row-panel kernel (check ops.panel_serves first)")
    gemm(OP_KC, OP_KC, x2d, x2d.stride(0), W, W.stride(0), out, out.stride(0), M, N, K, bias=bias, act=act,
         drop_p=drop_p, seed=seed, stream_id=stream_id, R=R, ldr=(R.stride(0) if R is not None else 0), out_split=out_split)
    if ln is not None:
        layernorm_fwd(out, ln[0], ln[1], ln[2], ln[3], ln[4], ln[5])
    return out


def gate_bits_bytes(M, N):
    """Bytes of the keep-bit buffer of an [M, N] activation (unast_panel_gemm gate_bits): whole 128-row panels, one bit per element."""
    return (M + 127) // 128 * 128 * ((N + 15) // 16 * 16) // 8


def panel_serves(M, K, W, transposed=False):
    """True when linear_fwd (transposed=False) / linear_dgrad (True) of this shape and stored weight run on the row-panel kernel."""
    return _panel_ok(M, K, N=(W.shape[1] if transposed else W.shape[0])) and W.stride(1) == 1 and _weight_planes(W, transposed) is not None


def linear_dgrad(dy2d, W, dx, R=None, G=None, gate_scale=1.0, beta=0, out_split=False, gate_bits=None):
    """dx[M,K] = (dy2d[M,N] @ W[N,K]) gated by G>0, + R.  gate_bits: the keep bits linear_fwd(..., gate_bits=) wrote for the [M,K]
    activation that G would be (panel kernel only): the gate is read from them with scalar loads instead of from G."""
    M, N = dy2d.shape
    K = W.shape[1]
    deep = N > 256                                  # K-streamed form (contraction over N): residual operand allowed, no gate, no split output
    if N % 4 == 0 and G is None and _panel_ok(M, N, dy2d, dx, R, N=K) and beta == 0 and W.stride(1) == 1 and \
            (R is None if not deep else (gate_bits is None and not out_split)):
        wp = _weight_planes(W, transposed=True)                # dX[M,K] = dY[M,N] (W^T)[K,N]^T: W^T planes, contraction over N
        if wp is not None:
            panel_gemm(dy2d, wp, dx, K, R=R, gate_scale=gate_scale, out_split=out_split, rows_per_wg=config.PANEL_ROWS, gate_bits=gate_bits)
            return dx
    if gate_bits is not None:
        raise RuntimeError("linear_dgrad: gate_bits needs the row-panel kernel (check ops.panel_serves first)")
    Np = (N + 3) // 4 * 4          # dy2d is a view of a zero-padded buffer when N % 4 != 0 (logits 46->48, head 81->84, fc2 1->4)
    if Np != N and dy2d.stride(0) < Np:
        raise ValueError("linear_dgrad: dy must live in a zero-padded buffer with row stride >= %d" % Np)
    wt = _transposed_weight(W) if config.DGRAD_TRANSPOSED else None
    if wt is not None:             # dX = dY (W^T)^T with W^T stored pre-split and K-contiguous: the forward GEMM's operand form
        gemm(OP_KC, OP_KC, dy2d, dy2d.stride(0), W, wt[1], dx, dx.stride(0), M, K, Np, R=R,
             ldr=(R.stride(0) if R is not None else 0), G=G, ldg=(G.stride(0) if G is not None else 0),
             gate_scale=gate_scale, beta=beta, b_ptr=wt[0], out_split=out_split)
        return dx
    gemm(OP_KC, OP_RC, dy2d, dy2d.stride(0), W, W.stride(0), dx, dx.stride(0), M, K, Np, R=R,
         ldr=(R.stride(0) if R is not None else 0), G=G, ldg=(G.stride(0) if G is not None else 0),
         gate_scale=gate_scale, beta=beta, kb_valid=N, out_split=out_split)
    return dx


LNBWD_FUSED = [0]         # launches of the fused form below (tests look at it)


def linear_dgrad_lnbwd(dy2d, W, R, z, mean, rstd, gamma, dz, dz_drop, dgamma, dbeta, drop_p=0.0, seed=0, stream_id=0):
    """The input-gradient GEMM that ends a sub-layer's backward together with the PREVIOUS sub-layer's LayerNorm backward:
    dz = LNbwd(R + dy2d @ W; z, mean, rstd, gamma), dz_drop = dropout(dz), dgamma / dbeta += column sums -- one launch of the K-streamed
    row-panel kernel (csrc/panel.hip kpanel_kernel<2>) instead of linear_dgrad + layernorm_bwd.  Returns False (and does nothing) when the
    kernel does not serve the shape: contraction dy2d.shape[1] > 256 and a multiple of 64, 256 output columns, at least
    config.PANEL_MIN_ROWS rows, tiled planes of W^T, 16-byte operands."""
    M, N = dy2d.shape
    if not (config.PANEL_LNBWD and W.shape[1] == 256 and N > 256 and W.stride(1) == 1 and z.shape[1] == 256 and
            _panel_ok(M, N, dy2d, dz, R, z, dz_drop, N=256) and gamma.data_ptr() % 16 == 0):
        return False
    wp = _weight_planes(W, transposed=True)
    if wp is None:
        return False
    part, part_n = None, 0
    if dgamma is not None:
        part_n = lib().unast_panel_gemm_lnbwd_ws_floats(M)
        part = torch.empty(part_n, dtype=torch.float32, device=dy2d.device)
    PANEL_LAUNCHES[1] += 1
    LNBWD_FUSED[0] += 1
    check(lib().unast_panel_gemm_lnbwd(_p(dy2d), dy2d.stride(0), wp[0], wp[1], M, N, _p(R), R.stride(0) if R is not None else 0,
                                       _p(z), z.stride(0), _p(mean), _p(rstd), _p(gamma), _p(dz), dz.stride(0),
                                       _p(dz_drop), dz_drop.stride(0) if dz_drop is not None else 0, _p(part), part_n,
                                       drop_p if dz_drop is not None else 0.0, seed & 0xFFFFFFFF, stream_id, _stream()), "unast_panel_gemm_lnbwd")
    if dgamma is not None:      # the reduction of the parameter-gradient partials is off the backward chain: companion stream
        nblk = (M + 127) // 128
        _on_wgrad_stream(lambda: check(lib().unast_layernorm_partials_finalize(_p(part), nblk, 256, _p(dgamma), _p(dbeta), _stream()),
                                       "unast_layernorm_partials_finalize"), M if config.LN_FINALIZE_OFFLOAD else 0, dgamma.data_ptr(), part)
    return True


# Set by engine.side_streams: returns the companion stream for weight gradients issued from the current stream, or None.
WGRAD_SIDE = None


# Stream choice per gradient buffer for the current optimizer phase (address of dW -> offloaded?).  The first weight-gradient
# launch into a buffer decides, later ones follow: the token count of a weight's gradient may differ between sub-steps (other
# batch, other memory length) and may straddle the gate, but its accumulations must all be ordered on ONE stream (defensive: the
# sub-step joins also wait for the companion streams).  Cleared by FlatStore.zero_grad.
_WGRAD_CHOICE = {}


def reset_wgrad_choices():
    _WGRAD_CHOICE.clear()


def _on_wgrad_stream(launch, tokens, key, *operands):
    """Runs `launch()` on the weight-gradient companion stream of the current stream (if any): that stream first waits for
    everything enqueued so far on the current one (the producers of dy / x), the operands are handed to the caching allocator
    with record_stream, and nobody waits for the result until the optimizer joins the streams.  All weight / bias gradient
    updates of a module come through here, so their read-modify-writes stay ordered on one stream."""
    w = WGRAD_SIDE() if WGRAD_SIDE is not None else None
    if w is not None:
        off = _WGRAD_CHOICE.get(key)
        if off is None:
            off = _WGRAD_CHOICE[key] = tokens >= config.WGRAD_STREAM_MIN_TOKENS
        if not off:
            w = None
    if w is None:
        return launch()
    from . import engine
    engine.wait(w, torch.cuda.current_stream())
    for t in operands:
        if t is not None:
            t.record_stream(w)
    with torch.cuda.stream(w):
        jitter()
        return launch()


def _linear_wgrad_now(dy2d, x2d, dW, db=None):
    M, N = dy2d.shape
    K = x2d.shape[1]
    sk = _splitk_for(N, K, M)
    if db is not None and deterministic_sums():          # the bias gradient in a fixed order, behind the weight gradient on its stream
        def both():
            gemm(OP_RC, OP_RC, dy2d, dy2d.stride(0), x2d, x2d.stride(0), dW, dW.stride(0), N, K, M, beta=1, splitk=sk)
            colsum(dy2d, db)
        _on_wgrad_stream(both, M, dW.data_ptr(), dy2d, x2d)
        return dW
    _on_wgrad_stream(lambda: gemm(OP_RC, OP_RC, dy2d, dy2d.stride(0), x2d, x2d.stride(0), dW, dW.stride(0), N, K, M, beta=1, splitk=sk,
                                  rowsum_a=db), M, dW.data_ptr(), dy2d, x2d)
    return dW


import ctypes as _ct


class _WgradItem(_ct.Structure):          # include/unast_hip.h: unast_wgrad_item
    _fields_ = [("A", _ct.c_void_p), ("lda", _ct.c_int), ("B", _ct.c_void_p), ("ldb", _ct.c_int), ("C", _ct.c_void_p), ("ldc", _ct.c_int),
                ("rowsum_a", _ct.c_void_p), ("M", _ct.c_int), ("N", _ct.c_int), ("K", _ct.c_int)]


_BATCH = None            # list of pending (dy2d, x2d, dW, db) while inside `wgrad_batch()`
GROUP_MAX = 8
WGRAD_GROUP_TARGET = int(_os.environ.get("UNAST_WGRAD_GROUP_TARGET", "256"))      # workgroups per grouped launch: 29.42 vs 29.65 ms/step at 512 (three alternating runs each, same box; round 2 preferred 512 when one stream bounded the step)


class wgrad_batch:
    """While active, eligible weight gradients (linear_wgrad) are collected and issued as ONE grouped launch at exit
    (unast_wgrad_group): the operands of a backward closure stay alive until it returns, and nobody reads a weight gradient
    before the optimizer, so deferring them to the closure's end changes no result."""

    def __enter__(self):
        global _BATCH
        self.outer = _BATCH is not None
        if not self.outer and config.WGRAD_GROUP:
            _BATCH = []
        return self

    def __exit__(self, *exc):
        global _BATCH
        if self.outer or _BATCH is None:
            return
        items, _BATCH = _BATCH, None
        if exc[0] is None:
            _flush_wgrads(items)


def _groupable(dy2d, x2d, dW, db):
    M, N = dy2d.shape
    K = x2d.shape[1]
    return (N % 128 == 0 and K % 128 == 0 and M % 32 == 0 and dy2d.stride(1) == 1 and x2d.stride(1) == 1 and dW.stride(1) == 1
            and dy2d.stride(0) % 4 == 0 and x2d.stride(0) % 4 == 0 and dW.stride(0) % 4 == 0
            and (dy2d.data_ptr() | x2d.data_ptr() | dW.data_ptr()) % 16 == 0 and M * max(dy2d.stride(0), x2d.stride(0)) < (1 << 29))


def _flush_wgrads(items):
    # the stream choice is per gradient buffer (see _WGRAD_CHOICE): problems bound for the companion stream form one group, the rest another
    w_side = WGRAD_SIDE() if WGRAD_SIDE is not None else None
    groups = {}
    for it in items:
        dy2d, x2d, dW, db = it
        key = dW.data_ptr()
        off = False
        if w_side is not None:
            off = _WGRAD_CHOICE.get(key)
            if off is None:
                off = _WGRAD_CHOICE[key] = dy2d.shape[0] >= config.WGRAD_STREAM_MIN_TOKENS
        groups.setdefault(bool(off), []).append(it)
    for off, its in groups.items():
        for i in range(0, len(its), GROUP_MAX):
            chunk = its[i:i + GROUP_MAX]
            if deterministic_sums():                       # (the grouped kernel takes the bias gradients by atomics)
                for it in chunk:
                    _linear_wgrad_now(*it)
            elif len(chunk) == 1:
                _linear_wgrad_now(*chunk[0])
            else:
                _launch_group(chunk, off)


def _launch_group(chunk, offload):
    n = len(chunk)
    arr = (_WgradItem * n)()
    operands = []
    for j, (dy2d, x2d, dW, db) in enumerate(chunk):
        a = arr[j]
        a.A, a.lda, a.B, a.ldb, a.C, a.ldc = dy2d.data_ptr(), dy2d.stride(0), x2d.data_ptr(), x2d.stride(0), dW.data_ptr(), dW.stride(0)
        a.rowsum_a = db.data_ptr() if db is not None else None
        a.M, a.N, a.K = dy2d.shape[1], x2d.shape[1], dy2d.shape[0]
        operands += [dy2d, x2d]
    ptr = _ct.addressof(arr)              # `arr` stays referenced by the closure below until the launch has been issued
    ws_n = lib().unast_wgrad_group_ws_floats(n, ptr, WGRAD_GROUP_TARGET)
    if ws_n <= 0:
        raise RuntimeError("unast_wgrad_group_ws_floats failed")

    shapes = tuple((int(a.M), int(a.N), int(a.K), int(a.rowsum_a is not None)) for a in arr)

    def launch():
        import sys
        sys.modules[__name__].wgrad_group(shapes, n, ptr, ws_n, chunk[0][0].device)       # looked up at call time: bench.py wraps it with HIP events
    if offload:
        w = WGRAD_SIDE()
        from . import engine
        engine.wait(w, torch.cuda.current_stream())
        for t in operands:
            t.record_stream(w)
        with torch.cuda.stream(w):
            launch()
    else:
        launch()


def wgrad_group(shapes, n, items_ptr, ws_n, device):
    """The grouped launch itself (GEMM + reduction on the current stream); `shapes` = ((out, in, tokens, has_bias), ...) for bookkeeping."""
    ws = torch.empty(ws_n, dtype=torch.float32, device=device)
    check(lib().unast_wgrad_group(config.NSPLIT, n, items_ptr, _p(ws), ws_n, WGRAD_GROUP_TARGET, _stream()), "unast_wgrad_group")


def linear_wgrad(dy2d, x2d, dW, db=None):
    """dW[N,K] += dy2d[M,N]^T @ x2d[M,K]; optionally db[N] += sum_rows dy2d (fused).  Always accumulates.  Inside
    `wgrad_batch()` eligible problems are deferred to the batch's grouped launch."""
    if _BATCH is not None and _groupable(dy2d, x2d, dW, db):
        _BATCH.append((dy2d, x2d, dW, db))
        return dW
    return _linear_wgrad_now(dy2d, x2d, dW, db)


_PANEL_TRACE = _os.environ.get("UNAST_PANEL_TRACE", "0") == "1"       # debugging: print and synchronise every row-panel launch


def retile_weights(src_flat, dst_bytes, descs_dev, ndesc):
    """Tiled bf16 planes (unast_amd.planes) of the matrices the descriptors name, from the fp32 buffer `src_flat`."""
    check(lib().unast_retile_weights(_p(src_flat), _p(dst_bytes), _p(descs_dev), ndesc, _stream()), "unast_retile_weights")


PANEL_LAUNCHES = [0, 0]          # (activation-stationary, K-streamed) launches so far: tests assert which kernel served them


def panel_gemm(A, wplanes, C, N, bias=None, R=None, G=None, gate_scale=1.0, act=0, drop_p=0.0, seed=0, stream_id=0, out_split=False,
               ln=None, rows_per_wg=0, K=None, gate_bits=None):
    """C[M,N] = epi(A[M,K] W[N,K]^T) with W given as tiled planes `wplanes` = (hi-plane pointer, plane bytes).
    ln = (gamma, beta, Y, mean, rstd, eps): LayerNorm epilogue (N = 256): C receives the pre-norm sum z, Y the normalised rows."""
    M = A.shape[0]
    K = A.shape[1] if K is None else K
    g = b = Y = mean = rstd = None
    eps, ldy = 0.0, 0
    if ln is not None:
        g, b, Y, mean, rstd, eps = ln
        ldy = Y.stride(0)
    PANEL_LAUNCHES[0 if K <= 256 else 1] += 1
    if _PANEL_TRACE:
        print("panel_gemm M=%d N=%d K=%d lda=%d ldc=%d bias=%s R=%s G=%s act=%d drop=%.2f split=%d ln=%s bits=%s rows=%d" % (
            M, N, K, A.stride(0), C.stride(0), bias is not None, R is not None, G is not None, act, drop_p, out_split, ln is not None,
            None if gate_bits is None else gate_bits.numel(), rows_per_wg), flush=True)
    check(lib().unast_panel_gemm(_p(A), A.stride(0), wplanes[0], wplanes[1], _p(C), C.stride(0), M, N, K, _p(bias), _p(R),
                                 R.stride(0) if R is not None else 0, _p(G), G.stride(0) if G is not None else 0, gate_scale, act,
                                 drop_p, seed & 0xFFFFFFFF, stream_id, int(out_split), _p(g), _p(b), _p(Y), ldy, _p(mean), _p(rstd), eps,
                                 _p(gate_bits), rows_per_wg, _stream()), "unast_panel_gemm")
    if _PANEL_TRACE:
        torch.cuda.synchronize()
    return C


def conv_fwd(x3d, Wp, bias, out, pad_left, colstats=None):
    """x3d [B,T,Cin], Wp [Cout,5,Cin] (tap-major physical layout), out [B,T,Cout].  colstats: float64 [2*Cout], zeroed by the
    caller, receives the per-channel sum and sum of squares of `out` (the batch statistics of the BatchNorm that follows)."""
    B, T, Cin = x3d.shape
    Cout = Wp.shape[0]
    gemm(OP_KC_CONV, OP_KC, x3d, x3d.stride(1), Wp, 5 * Cin, out, out.stride(1), B * T, Cout, 5 * Cin,
         conv=(T, Cin, 0, pad_left), bias=bias, colstats=colstats)
    return out


def conv_dgrad(dy3d, Wp, dx, pad_left, beta=0):
    B, T, Cout = dy3d.shape
    Cin = Wp.shape[2]
    gemm(OP_KC_CONV, OP_RC_CONV_DGRAD, dy3d, dy3d.stride(1), Wp, 4, dx, dx.stride(1), B * T, Cin, 5 * Cout,
         conv=(T, Cout, Cout, 4 - pad_left), beta=beta)
    return dx


def conv_wgrad(dy3d, x3d, dWp, pad_left, db=None):
    """dWp[Cout,5,Cin] += sum_t dy[t,o] x[t+j-pad_left,c]; optionally db[Cout] += sum_t dy (fused)."""
    B, T, Cout = dy3d.shape
    Cin = x3d.shape[2]
    sk = _splitk_for(Cout, 5 * Cin, B * T)
    if db is not None and deterministic_sums() and dy3d.is_contiguous():
        def both():
            gemm(OP_RC, OP_RC_CONV_WGRAD, dy3d, dy3d.stride(1), x3d, x3d.stride(1), dWp, 5 * Cin, Cout, 5 * Cin, B * T,
                 conv=(T, 0, Cin, pad_left), beta=1, splitk=sk)
            colsum(dy3d.view(B * T, Cout), db)
        _on_wgrad_stream(both, B * T, dWp.data_ptr(), dy3d, x3d)
        return dWp
    _on_wgrad_stream(lambda: gemm(OP_RC, OP_RC_CONV_WGRAD, dy3d, dy3d.stride(1), x3d, x3d.stride(1), dWp, 5 * Cin, Cout, 5 * Cin, B * T,
                                  conv=(T, 0, Cin, pad_left), beta=1, splitk=sk, rowsum_a=db), B * T, dWp.data_ptr(), dy3d, x3d)
    return dWp


# ---- attention ---------------------------------------------------------------------------------------------
def attn_fwd(Q, K, V, O, LSE, lens_k, B, H, Tq, Tk, causal, drop_p=0.0, seed=0, stream_id=0, nsplit=None, qkv_split=False):
    """Q/K/V/O are 2-D views [B*T, >=H*64] (column slices of projection outputs are fine)."""
    check(lib().unast_attn_fwd(nsplit or config.NSPLIT, _p(Q), Q.stride(0), _p(K), K.stride(0), _p(V), V.stride(0), _p(O), O.stride(0),
                               _p(LSE), _p(lens_k), B, H, Tq, Tk, 64, int(causal), 0.125, drop_p, seed & 0xFFFFFFFF, stream_id,
                               int(qkv_split), _stream()), "unast_attn_fwd")


def attn_bwd(Q, K, V, O, dO, LSE, delta_ws, dQ, dK, dV, lens_k, B, H, Tq, Tk, causal, drop_p=0.0, seed=0, stream_id=0, nsplit=None, qkv_split=False, lens_q=None):
    """lens_q (int32 [B], one-pass backward only): queries t >= lens_q[b] have a zero dO by the caller's guarantee; their tiles are skipped."""
    fused = config.ATTN_FUSED_BWD and not deterministic_sums()          # (the one-pass kernel adds dQ with fp32 atomics)
    check(lib().unast_attn_bwd(nsplit or config.NSPLIT, _p(Q), Q.stride(0), _p(K), K.stride(0), _p(V), V.stride(0), _p(O), O.stride(0),
                               _p(dO), dO.stride(0), _p(LSE), _p(delta_ws), _p(dQ), dQ.stride(0), _p(dK), dK.stride(0), _p(dV),
                               dV.stride(0), _p(lens_k), B, H, Tq, Tk, 64, int(causal), 0.125, drop_p, seed & 0xFFFFFFFF, stream_id,
                               (2 if config.ATTN_BWD_TERMS == 2 else 1) if fused else 0, int(qkv_split),
                               _p(lens_q) if fused else None, _stream()), "unast_attn_bwd")


# ---- normalisation -----------------------------------------------------------------------------------------
def layernorm_fwd(z, gamma, beta, y, mean, rstd, eps=1e-5):
    rows, C = z.shape
    check(lib().unast_layernorm_fwd(_p(z), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, C, eps, _stream()), "unast_layernorm_fwd")


def layernorm_bwd(dy, z, gamma, mean, rstd, dz, dz_drop=None, dgamma=None, dbeta=None, drop_p=0.0, seed=0, stream_id=0):
    rows, C = z.shape
    ws, ws_n = None, 0
    if dgamma is not None:
        ws_n = _LN_WS.get((rows, C))
        if ws_n is None:
            ws_n = _LN_WS[(rows, C)] = lib().unast_layernorm_bwd_ws_floats(rows, C)
        ws = torch.empty(ws_n, dtype=torch.float32, device=z.device)
    check(lib().unast_layernorm_bwd(_p(dy), _p(z), _p(gamma), _p(mean), _p(rstd), _p(dz), _p(dz_drop), _p(dgamma), _p(dbeta), _p(ws), ws_n,
                                    rows, C, drop_p, seed & 0xFFFFFFFF, stream_id, 0 if dgamma is not None else 1, _stream()), "unast_layernorm_bwd")
    if dgamma is not None:      # the reduction of the parameter-gradient partials is off the backward chain: companion stream
        _on_wgrad_stream(lambda: check(lib().unast_layernorm_bwd_finalize(_p(ws), ws_n, rows, C, _p(dgamma), _p(dbeta), _stream()),
                                       "unast_layernorm_bwd_finalize"), rows if config.LN_FINALIZE_OFFLOAD else 0, dgamma.data_ptr(), ws)


def colsum(x2d, out):
    rows, C = x2d.shape
    if deterministic_sums():
        ws_n = lib().unast_colsum_det_ws_floats(rows, C)
        ws = torch.empty(ws_n, dtype=torch.float32, device=x2d.device)
        check(lib().unast_colsum_det(_p(x2d), x2d.stride(0), rows, C, _p(out), _p(ws), ws_n, _stream()), "unast_colsum_det")
        return
    check(lib().unast_colsum_f32(_p(x2d), x2d.stride(0), rows, C, _p(out), _stream()), "unast_colsum_f32")


def bn_fwd(x2d, gamma, beta, y, mean, rstd, running_mean, running_var, ws, act, drop_p=0.0, seed=0, stream_id=0, eps=1e-5, momentum=0.1, have_sums=False,
           num_batches_tracked=None):
    rows, C = x2d.shape
    check(lib().unast_bn_fwd(_p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(running_mean), _p(running_var), _p(ws), rows, C,
                             eps, momentum, act, drop_p, seed & 0xFFFFFFFF, stream_id, int(have_sums), _p(num_batches_tracked), _stream()), "unast_bn_fwd")


def bn_eval_fwd(x2d, gamma, beta, running_mean, running_var, y, mean, rstd, act, eps=1e-5):
    rows, C = x2d.shape
    check(lib().unast_bn_eval_fwd(_p(x2d), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), _p(mean), _p(rstd), rows, C,
                                  eps, act, _stream()), "unast_bn_eval_fwd")


def bn_bwd(dy_inout, x2d, mean, rstd, gamma, beta, dx, dgamma, dbeta, ws, act, drop_p=0.0, seed=0, stream_id=0, ws_zeroed=False):
    rows, C = x2d.shape
    check(lib().unast_bn_bwd(_p(dy_inout), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx), _p(dgamma), _p(dbeta), _p(ws), rows, C,
                             act, drop_p, seed & 0xFFFFFFFF, stream_id, int(ws_zeroed), _stream()), "unast_bn_bwd")


# ---- embedding / positional encoding / masks ---------------------------------------------------------------
def embed_fwd(ids, E, out, T, shift_sos=-1, drop_p=0.0, seed=0, stream_id=0, noise_p=0.0, noise_stream=0):
    rows = ids.numel()
    check(lib().unast_embed_fwd(_p(ids), _p(E), _p(out), rows, T, E.shape[1], shift_sos, drop_p, seed & 0xFFFFFFFF, stream_id, noise_p,
                                noise_stream, _stream()), "unast_embed_fwd")


def embed_bwd(ids, dout, dE, T, shift_sos=-1, drop_p=0.0, seed=0, stream_id=0, noise_p=0.0, noise_stream=0, padding_idx=0):
    rows = ids.numel()
    fn = lib().unast_embed_bwd_det if deterministic_sums() else lib().unast_embed_bwd
    check(fn(_p(ids), _p(dout), _p(dE), rows, T, dE.shape[1], dE.shape[0], shift_sos, padding_idx, drop_p,
             seed & 0xFFFFFFFF, stream_id, noise_p, noise_stream, _stream()), "unast_embed_bwd")


def posenc_fwd(x2d, pe, y, T, scale, drop_p=0.0, seed=0, stream_id=0):
    rows, D = x2d.shape
    check(lib().unast_posenc_fwd(_p(x2d), _p(pe), _p(y), rows, T, D, scale, drop_p, seed & 0xFFFFFFFF, stream_id, _stream()), "unast_posenc_fwd")


def posenc_bwd(dy, gate, dx, scale, drop_p=0.0, seed=0, stream_id=0):
    rows, D = dy.shape
    check(lib().unast_posenc_bwd(_p(dy), _p(gate), _p(dx), rows, D, scale, drop_p, seed & 0xFFFFFFFF, stream_id, _stream()), "unast_posenc_bwd")


def rowmask(x2d, y, p, seed, stream_id):
    rows, D = x2d.shape
    check(lib().unast_rowmask(_p(x2d), _p(y), rows, D, p, seed & 0xFFFFFFFF, stream_id, _stream()), "unast_rowmask")


def add_inplace(a, b):
    check(lib().unast_add_inplace(_p(a), _p(b), a.numel(), _stream()), "unast_add_inplace")


def sum2(dst, a, b=None):
    """dst = a + b (b None: dst = a), contiguous fp32 tensors of one size."""
    check(lib().unast_sum2(_p(dst), _p(a), _p(b), dst.numel(), _stream()), "unast_sum2")


def argmax_rows(x2d, cols, out_i64):
    check(lib().unast_argmax_rows(_p(x2d), x2d.stride(0), x2d.shape[0], cols, _p(out_i64), _stream()), "unast_argmax_rows")


def mask_by_len(x3d, lens_i64):
    B, T = x3d.shape[0], x3d.shape[1]
    D = x3d.numel() // (B * T)
    check(lib().unast_mask_by_len(_p(x3d), _p(lens_i64), B, T, D, _stream()), "unast_mask_by_len")


PRO_NONE, PRO_LN, PRO_LN_DROP, PRO_EMBED, PRO_POSENC = 0, 1, 2, 3, 4
def decode_linear(X, W, bias, Y, act=0, drop_p=0.0, seed=0, stream_id=0, R=None, ln=None, ln_drop=None, embed=None, posenc=None, xn_out=None,
                  cache=None, split_col=0, pos=None, x_frames=None):
    """Y = epilogue(X' W^T) for the rows of one decoding position (csrc/decode.hip).  X' = X, or
    ln=(gamma, beta): LayerNorm(X);  + ln_drop=(p, stream): dropout of that;
    embed=(tokens [B,T], table, pe, scale, (p1, stream1), (p2, stream2)): dropout(dropout(table[tokens[:, pos]]) * scale + pe[pos]);
    posenc=(pe, scale, (p2, stream2)): dropout(X * scale + pe[pos]).
    x_frames: X is a [B, T, K] buffer and the rows are its slice at the device-resident position `pos`."""
    N, K = W.shape
    assert W.stride(1) == 1
    pro, g, b, tokens, ld_tok, emb, pe, scale, d1, d2 = PRO_NONE, None, None, None, 0, None, None, 1.0, (0.0, 0), (0.0, 0)
    if ln is not None:
        g, b = ln
        pro = PRO_LN
        if ln_drop is not None and ln_drop[0] > 0:
            pro, d2 = PRO_LN_DROP, ln_drop
    elif embed is not None:
        tokens, emb, pe, scale, d1, d2 = embed
        pro, ld_tok = PRO_EMBED, tokens.stride(0)
    elif posenc is not None:
        pe, scale, d2 = posenc
        pro = PRO_POSENC
    if x_frames is not None:
        X, ldx, xps, M = x_frames, x_frames.stride(0), x_frames.stride(1), x_frames.shape[0]
        assert x_frames.stride(2) == 1 and x_frames.shape[2] == K
    elif X is not None:
        assert X.shape[1] == K and X.stride(1) == 1
        ldx, xps, M = X.stride(0), 0, X.shape[0]
    else:
        ldx, xps, M = 0, 0, tokens.shape[0]
    check(lib().unast_decode_linear(_p(X), ldx, xps, _p(W), W.stride(0), _p(bias), _p(Y), Y.stride(0) if Y is not None else 0, M, N, K, act,
                                    float(drop_p), seed, stream_id, _p(R), R.stride(0) if R is not None else 0,
                                    pro, _p(g), _p(b), 1e-5, _p(tokens), ld_tok, _p(emb), _p(pe), float(scale),
                                    float(d1[0]), d1[1], float(d2[0]), d2[1],
                                    _p(xn_out), xn_out.stride(0) if xn_out is not None else 0,
                                    _p(cache), cache.stride(1) if cache is not None else 0, cache.shape[1] if cache is not None else 0, split_col, _p(pos),
                                    _stream()), "unast_decode_linear")


def decode_attn(Q, K, V, rows_per_seq, O, H, lens=None, stop_lens=None, pos=None, drop_p=0.0, seed=0, stream_id=0):
    """Single-query attention over cached K/V rows ([B*rows_per_seq, ld] views), head dim 64; valid keys: lens[b], or
    min(stop_lens[b] + 1, pos + 1) with both on the device."""
    B = Q.shape[0]
    assert K.stride(0) == V.stride(0) and Q.shape[1] == 64 * H
    check(lib().unast_decode_attn(_p(Q), Q.stride(0), _p(K), _p(V), K.stride(0), rows_per_seq, _p(lens), _p(stop_lens), _p(pos), _p(O), O.stride(0), B, H, 0.125,
                                  float(drop_p), seed, stream_id, _stream()), "unast_decode_attn")


def decode_end_text(logits, V, tokens, stop_lens, max_len, eos, pos, epoch):
    check(lib().unast_decode_end_text(_p(logits), logits.stride(0), V, logits.shape[0], _p(tokens), tokens.stride(0), _p(stop_lens), max_len, eos, _p(pos),
                                      _p(epoch), _stream()), "unast_decode_end_text")


def decode_end_speech(head, M, outputs, stops, stop_lens, max_len, pos, epoch):
    check(lib().unast_decode_end_speech(_p(head), head.stride(0), M, head.shape[0], _p(outputs), outputs.stride(0), _p(stops), stops.stride(0), _p(stop_lens),
                                        max_len, _p(pos), _p(epoch), _stream()), "unast_decode_end_speech")


def scale_inplace(a, alpha):
    check(lib().unast_scale_inplace(_p(a), float(alpha), a.numel(), _stream()), "unast_scale_inplace")


_JITTER = [0]


def jitter():
    """config.STREAM_JITTER > 0: a spin of pseudo-random length (0 .. STREAM_JITTER us, a third of the calls none) on the current stream."""
    if not config.STREAM_JITTER:
        return
    _JITTER[0] = (_JITTER[0] * 1103515245 + 12345 + config.STREAM_JITTER_SEED) & 0x7FFFFFFF
    r = (_JITTER[0] >> 8) % 3000
    if r >= 2000:
        return
    check(lib().unast_spin(int(r * config.STREAM_JITTER / 2000), _stream()), "unast_spin")


def shift_frames(mel3d, out3d):
    """out[b,0] = 0, out[b,t] = mel[b,t-1] (the decoder input of teacher forcing)."""
    B, T, M = mel3d.shape
    check(lib().unast_shift_frames(_p(mel3d), _p(out3d), B, T, M, _stream()), "unast_shift_frames")
    return out3d


def add_strided(dst2d, src2d, cols):
    """dst2d[:, :cols] += src2d[:, :cols] (row strides may differ)."""
    rows = dst2d.shape[0]
    check(lib().unast_add_strided(_p(dst2d), dst2d.stride(0), _p(src2d), src2d.stride(0), rows, cols, _stream()), "unast_add_strided")


def specaugment(mel, lens_i32, out, seed, stream_id, freq_mask=20, time_mask=100):
    B, T, M = mel.shape
    ws = torch.empty(B, dtype=torch.float32, device=mel.device)
    check(lib().unast_specaugment(_p(mel), _p(lens_i32), _p(out), _p(ws), B, T, M, freq_mask, time_mask, seed & 0xFFFFFFFF, stream_id, _stream()),
          "unast_specaugment")


def disc_gather(t_hid, s_hid, t_len, s_len, perm, out, out_len):
    B, Tt, D = t_hid.shape
    Ts = s_hid.shape[1]
    check(lib().unast_disc_gather(_p(t_hid), _p(s_hid), _p(t_len), _p(s_len), _p(perm), _p(out), _p(out_len), B, Tt, Ts, D, _stream()),
          "unast_disc_gather")


def disc_scatter(dout, perm, dt_hid, ds_hid):
    B, Tt, D = dt_hid.shape
    Ts = ds_hid.shape[1]
    check(lib().unast_disc_scatter(_p(dout), _p(perm), _p(dt_hid), _p(ds_hid), B, Tt, Ts, D, _stream()), "unast_disc_scatter")


# ---- losses ------------------------------------------------------------------------------------------------
def speech_loss_fwd(gold, head, post, lens_i32, eos_weight, ws, loss):
    B, T, M = gold.shape
    check(lib().unast_speech_loss_fwd(_p(gold), _p(head), head.stride(-2), _p(post), _p(lens_i32), B, T, M, eos_weight, _p(ws), _p(loss),
                                      _stream()), "unast_speech_loss_fwd")


def speech_loss_bwd(gold, head, post, lens_i32, eos_weight, gscale, d_head, d_post):
    B, T, M = gold.shape
    check(lib().unast_speech_loss_bwd(_p(gold), _p(head), head.stride(-2), _p(post), _p(lens_i32), B, T, M, eos_weight, _p(gscale),
                                      _p(d_head), _p(d_post), _stream()), "unast_speech_loss_bwd")


def text_loss_fwd(logits2d, gold, V, eos_weight, ws, loss):
    check(lib().unast_text_loss_fwd(_p(logits2d), logits2d.stride(0), _p(gold), logits2d.shape[0], V, eos_weight, _p(ws), _p(loss),
                                    _stream()), "unast_text_loss_fwd")


def text_head_loss(x2d, W, bias, gold, V, eos_weight, gscale, logits2d, dlogits2d, ws, loss):
    """TextPostnet.fc1 + text_loss + its gradient with respect to the logits in one launch (csrc/loss.hip text_head_loss_kernel)."""
    check(lib().unast_text_head_loss(_p(x2d), x2d.stride(0), _p(W), _p(bias), _p(gold), x2d.shape[0], x2d.shape[1], V, float(eos_weight), float(gscale),
                                     _p(logits2d), _p(dlogits2d), logits2d.stride(0), _p(ws), _p(loss), _stream()), "unast_text_head_loss")


def speech_head_loss(x2d, W, bias, gold2d, lens, B, T, M, eos_weight, gscale, head, d_head, ws):
    """[linear_project | stop_linear] + the pre-net MSE and stop BCE of speech_loss + their gradient in one launch (csrc/loss.hip)."""
    check(lib().unast_speech_head_loss(_p(x2d), x2d.stride(0), _p(W), _p(bias), _p(gold2d), _p(lens), B, T, x2d.shape[1], M, float(eos_weight), float(gscale),
                                       _p(head), _p(d_head), head.stride(0), _p(ws), _stream()), "unast_speech_head_loss")


def speech_post_loss(gold2d, post2d, lens, B, T, M, gscale, d_post, ws, loss):
    """The post-net MSE term, its gradient and the finished speech_loss scalar (after speech_head_loss on the same stream)."""
    check(lib().unast_speech_post_loss(_p(gold2d), _p(post2d), _p(lens), B, T, M, float(gscale), _p(d_post), _p(ws), _p(loss), _stream()), "unast_speech_post_loss")


def text_loss_bwd(logits2d, gold, V, eos_weight, ws, gscale, dlogits):
    check(lib().unast_text_loss_bwd(_p(logits2d), logits2d.stride(0), _p(gold), logits2d.shape[0], V, eos_weight, _p(ws), _p(gscale),
                                    _p(dlogits), _stream()), "unast_text_loss_bwd")


def disc_targets(perm, B, flip, out, smoothing=0.1):
    check(lib().unast_disc_targets(_p(perm), perm.numel(), B, int(flip), smoothing, _p(out), _stream()), "unast_disc_targets")


def randperm(n, seed, stream_id, device):
    out = torch.empty(n, dtype=torch.int64, device=device)
    check(lib().unast_randperm(_p(out), n, seed & 0xFFFFFFFF, stream_id, _stream()), "unast_randperm")
    return out


def bce_logits(logits, ldx, targets, n, loss=None, gscale=None, dlogits=None, ldd=1):
    check(lib().unast_bce_logits(_p(logits), ldx, _p(targets), n, _p(gscale), _p(loss), _p(dlogits), ldd, _stream()), "unast_bce_logits")


# ---- LSTM discriminator ------------------------------------------------------------------------------------
def lstm_fwd(xproj, whh, b_ih, b_hh, lens_i32, y, gates, cs, hprev, hfinal, ndir, whh_dir_stride, bias_dir_stride):
    Bd, T = xproj.shape[0], xproj.shape[1]
    check(lib().unast_lstm_fwd(_p(xproj), _p(whh), _p(b_ih), _p(b_hh), _p(lens_i32), _p(y), _p(gates), _p(cs), _p(hprev), _p(hfinal), Bd, T,
                               ndir, 64, whh_dir_stride, bias_dir_stride, _stream()), "unast_lstm_fwd")


def lstm_bwd(dy, dhfinal, whh, gates, cs, lens_i32, dgates, ndir, whh_dir_stride):
    Bd, T = gates.shape[0], gates.shape[1]
    check(lib().unast_lstm_bwd(_p(dy), _p(dhfinal), _p(whh), _p(gates), _p(cs), _p(lens_i32), _p(dgates), Bd, T, ndir, 64, whh_dir_stride,
                               _stream()), "unast_lstm_bwd")


def leaky_dropout(x, dy, out, slope, drop_p=0.0, seed=0, stream_id=0):
    rows = x.shape[0]
    D = x.numel() // rows
    check(lib().unast_leaky_dropout(_p(x), _p(dy), _p(out), rows, D, slope, drop_p, seed & 0xFFFFFFFF, stream_id, _stream()),
          "unast_leaky_dropout")


# ---- optimizer ---------------------------------------------------------------------------------------------
_MSE_WS = {}


def masked_mse(gold, pred, mask):
    """src/train.py:100-103 on flat fp32 tensors; returns a device scalar.  The kernel's workspace (two sums and an arrival counter,
    zero on entry, left zero) is one per (device, stream): calls on different streams may overlap, calls on one stream are ordered."""
    dev = gold.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _MSE_WS.get(key)
    if ws is None:
        ws = _MSE_WS[key] = torch.zeros(3, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)                  # filled once: nobody may run ahead of the fill
    if config.DEBUG_WORKSPACES and bool((ws != 0).any()):
        raise RuntimeError("masked_mse: workspace not zero on entry (an earlier call on this stream died midway?)")
    out = torch.empty((), dtype=torch.float32, device=dev)
    check(lib().unast_masked_mse(_p(gold), _p(pred), _p(mask), gold.numel(), _p(ws), _p(out), _stream()), "unast_masked_mse")
    return out


def sumsq(g, out):
    check(lib().unast_sumsq(_p(g), g.numel(), _p(out), _stream()), "unast_sumsq")


def scalar_combine(xs, div, out):
    """out = (xs[0] + xs[1] + xs[2]) / div over 1 to 3 device scalars."""
    a = xs[0]
    b = xs[1] if len(xs) > 1 else None
    c = xs[2] if len(xs) > 2 else None
    check(lib().unast_scalar_combine(_p(a), _p(b), _p(c), float(div), _p(out), _stream()), "unast_scalar_combine")
    return out


def adamw(p, g, m, v, sumsq_scalar, max_norm, lr, beta1, beta2, eps, wd, step, split_out=None, decoupled=True, dev_hyper=None, zero_grad=False):
    """dev_hyper: float32 [3] device tensor {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read by the kernel instead of lr / step."""
    check(lib().unast_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(sumsq_scalar), max_norm, lr, beta1, beta2, eps, wd, step,
                            _p(split_out), int(bool(decoupled)), _p(dev_hyper), int(bool(zero_grad)), _stream()), "unast_adamw")


def adam_hyper(lr, beta1, beta2, step):
    """The three floats unast_adamw derives from (lr, step) on the host -- {lr, 1 - beta1^t, sqrt(1 - beta2^t)} with the betas
    rounded to fp32 first, as the C entry point receives them -- for the device-resident block a captured step reads."""
    import ctypes
    import math
    b1, b2 = ctypes.c_float(beta1).value, ctypes.c_float(beta2).value
    return [float(lr), 1.0 - b1 ** int(step), math.sqrt(1.0 - b2 ** int(step))]


def transpose_split(src_flat, dst_flat, tiles_i32, ntiles):
    check(lib().unast_transpose_split(_p(src_flat), _p(dst_flat), _p(tiles_i32), ntiles, _stream()), "unast_transpose_split")


def split_f32(src, dst):
    """dst = src in the GEMM's pre-split operand format (16-B chunks [hi x4 | lo x4] of bf16; same byte offsets)."""
    check(lib().unast_split_f32(_p(src), _p(dst), src.numel(), _stream()), "unast_split_f32")


_STEP_STATE = {}
STEP_STATE_WORDS = 64


def step_state():
    """Per-device block of 64 32-bit words in device memory that captured (HIP-graph) work reads at replay time:
    word 0 = the RNG epoch mixed into every dropout / noise stream (unast_set_rng_epoch), words 4.. = float triples
    {lr, 1 - beta1^t, sqrt(1 - beta2^t)} for unast_adamw's dev_hyper.  One small host-to-device copy refreshes all of it."""
    dev = torch.cuda.current_device()
    c = _STEP_STATE.get(dev)
    if c is None:
        torch.cuda.synchronize()
        c = torch.zeros(STEP_STATE_WORDS, dtype=torch.int32, device="cuda:%d" % dev)
        torch.cuda.synchronize()
        check(lib().unast_set_rng_epoch(c.data_ptr()), "unast_set_rng_epoch")
        _STEP_STATE[dev] = c
    return c


def rng_epoch_counter():
    """The device counter mixed into every dropout / noise stream (unast_set_rng_epoch): int32 [1], 0 outside graph replay."""
    return step_state()[0:1]


def set_step_state(epoch, hyper_by_slot):
    """One tiny launch: RNG epoch (word 0) and the hyper-parameter triples {slot: [lr, bc1, bc2_sqrt]} (words 4 + 3*slot ..)."""
    import ctypes
    import struct
    words = (ctypes.c_uint * 16)()
    words[0] = epoch & 0xFFFFFFFF
    n = 4
    for slot, vals in hyper_by_slot.items():
        if 4 + 3 * slot + 3 > 16:
            raise ValueError("set_step_state: at most 4 optimizer ranges")
        for j, v in enumerate(vals):
            words[4 + 3 * slot + j] = struct.unpack("<I", struct.pack("<f", float(v)))[0]
        n = max(n, 4 + 3 * slot + 3)
    check(lib().unast_set_words(_p(step_state()), ctypes.addressof(words), n, _stream()), "unast_set_words")


def hyper_slot(i):
    """float32 [3] view of the i-th optimizer hyper-parameter triple inside step_state()."""
    return step_state().view(torch.float32)[4 + 3 * i: 7 + 3 * i]
