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


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t, name):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 CUDA tensor (got %s on %s)" % (name, t.dtype, t.device))


def gemm(a_mode, b_mode, A, lda, B, ldb, C, ldc, M, N, K, conv=(0, 0, 0, 0), bias=None, R=None, ldr=0, G=None,
         ldg=0, gate_scale=1.0, alpha=1.0, beta=0, act=0, drop_p=0.0, seed=0, stream_id=0, splitk=1, nsplit=None):
    for t, n in ((A, "A"), (B, "B"), (C, "C"), (bias, "bias"), (R, "R"), (G, "G")):
        _f32(t, n)
    check(lib().unast_gemm(a_mode, b_mode, nsplit or config.NSPLIT, _p(A), lda, _p(B), ldb, _p(C), ldc, M, N, K,
                           conv[0], conv[1], conv[2], conv[3], _p(bias), _p(R), ldr, _p(G), ldg, gate_scale,
                           alpha, beta, act, drop_p, seed & 0xFFFFFFFF, stream_id, splitk, _stream()), "unast_gemm")


def _splitk_for(M, N, K):
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    want = max(1, 768 // tiles)
    return max(1, min(want, K // 128 if K >= 128 else 1))


def linear_fwd(x2d, W, bias, out, act=0, drop_p=0.0, seed=0, stream_id=0, R=None):
    """out[M,N] = epi(x2d[M,K] @ W[N,K]^T + bias)."""
    M, K = x2d.shape
    N = W.shape[0]
    gemm(OP_KC, OP_KC, x2d, x2d.stride(0), W, W.stride(0), out, out.stride(0), M, N, K, bias=bias, act=act,
         drop_p=drop_p, seed=seed, stream_id=stream_id, R=R, ldr=(R.stride(0) if R is not None else 0))
    return out


def linear_dgrad(dy2d, W, dx, R=None, G=None, gate_scale=1.0, beta=0):
    """dx[M,K] = (dy2d[M,N] @ W[N,K]) gated by G>0, + R."""
    M, N = dy2d.shape
    K = W.shape[1]
    gemm(OP_KC, OP_RC, dy2d, dy2d.stride(0), W, W.stride(0), dx, dx.stride(0), M, K, N, R=R,
         ldr=(R.stride(0) if R is not None else 0), G=G, ldg=(G.stride(0) if G is not None else 0),
         gate_scale=gate_scale, beta=beta)
    return dx


def linear_wgrad(dy2d, x2d, dW):
    """dW[N,K] += dy2d[M,N]^T @ x2d[M,K]  (always accumulates; split-K with fp32 atomics)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    sk = _splitk_for(N, K, M)
    gemm(OP_RC, OP_RC, dy2d, dy2d.stride(0), x2d, x2d.stride(0), dW, dW.stride(0), N, K, M, beta=1, splitk=sk)
    return dW


def conv_fwd(x3d, Wp, bias, out, pad_left):
    """x3d [B,T,Cin], Wp [Cout,5,Cin] (tap-major physical layout), out [B,T,Cout]."""
    B, T, Cin = x3d.shape
    Cout = Wp.shape[0]
    gemm(OP_KC_CONV, OP_KC, x3d, x3d.stride(1), Wp, 5 * Cin, out, out.stride(1), B * T, Cout, 5 * Cin,
         conv=(T, Cin, 0, pad_left), bias=bias)
    return out


def conv_dgrad(dy3d, Wp, dx, pad_left, beta=0):
    B, T, Cout = dy3d.shape
    Cin = Wp.shape[2]
    gemm(OP_KC_CONV, OP_RC_CONV_DGRAD, dy3d, dy3d.stride(1), Wp, 4, dx, dx.stride(1), B * T, Cin, 5 * Cout,
         conv=(T, Cout, Cout, 4 - pad_left), beta=beta)
    return dx


def conv_wgrad(dy3d, x3d, dWp, pad_left):
    """dWp[Cout,5,Cin] += sum_t dy[t,o] x[t+j-pad_left,c]."""
    B, T, Cout = dy3d.shape
    Cin = x3d.shape[2]
    sk = _splitk_for(Cout, 5 * Cin, B * T)
    gemm(OP_RC, OP_RC_CONV_WGRAD, dy3d, dy3d.stride(1), x3d, x3d.stride(1), dWp, 5 * Cin, Cout, 5 * Cin, B * T,
         conv=(T, 0, Cin, pad_left), beta=1, splitk=sk)
    return dWp
