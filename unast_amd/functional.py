"""Forward + hand-written backward of every block on the UNAST train-step hot path, as sequences of HIP kernel
launches (unast_amd.ops).  Each function runs the forward immediately and, when a Tape is given, records one closure
that consumes the gradient of its output Var(s) and produces the gradients of its inputs / parameters.

Reference semantics per block are cited next to each function (file:line in /root/reference).
Activations are fp32 [B*T, C] row-major (token-major); `lens` are int32 device tensors.
"""
import math

import torch

from . import config, ops
from .engine import Var, acc
from .utils import is_deterministic

LN_EPS = 1e-5


class Ctx:
    """Per-call context: parameter store, dropout configuration, RNG stream allocation."""

    def __init__(self, store, training, seed):
        self.st = store
        store.sync_split()
        self.P = store.phys
        self.training = training
        self.stochastic = training and not is_deterministic()      # dropout sites active
        self.noisy = not is_deterministic()                         # noise_fn ignores model.training (src/utils.py:40-49)
        self.seed = seed & 0xFFFFFFFF
        self._stream = 0

    def stream(self):
        self._stream += 1
        return self._stream

    def p(self, rate):
        return float(rate) if self.stochastic else 0.0


def _empty(*shape, like=None, device=None):
    return torch.empty(*shape, dtype=torch.float32, device=(like.device if like is not None else device))


def _bias_grad(cx, dy2d, name_or_view):
    g = cx.st.g(name_or_view) if isinstance(name_or_view, str) else name_or_view
    if g is not None:
        ops.colsum(dy2d, g)


# ---------------------------------------------------------------------------------------------------------------
# Transformer sub-layers (torch.nn.TransformerEncoderLayer / DecoderLayer, post-LN, ReLU; src/module.py:273-274,
# 286-287; exact math SURVEY.md Appendix A)
# ---------------------------------------------------------------------------------------------------------------
def _layernorm(cx, tape, z, pre, x_res_grad_sink):
    """y = LN(z).  Backward returns (dz, dz_dropped) through `x_res_grad_sink(dz, dzd)`."""
    rows, C = z.shape
    y = _empty(rows, C, like=z)
    mean = _empty(rows, like=z)
    rstd = _empty(rows, like=z)
    ops.layernorm_fwd(z, cx.P[pre + "weight"], cx.P[pre + "bias"], y, mean, rstd, LN_EPS)
    return y, mean, rstd


def _ln_backward(cx, out, z, mean, rstd, pre_norm, p, seed, stream_id):
    """(dz, dz_dropped-or-dz) of the LayerNorm that produced `out`: already there when the consuming sub-layer's last GEMM ran it in its
    epilogue (out.pre, _input_grad below), otherwise the stand-alone kernel on out.g."""
    if out.pre is not None:
        dz, dzd = out.pre
        out.pre = None
        return dz, (dzd if p > 0 else dz)
    st = cx.st
    dz = _empty(*z.shape, like=z)
    dzd = _empty(*z.shape, like=z) if p > 0 else None
    ops.layernorm_bwd(out.g, z, cx.P[pre_norm + "weight"], mean, rstd, dz, dzd, st.g(pre_norm + "weight"), st.g(pre_norm + "bias"),
                      drop_p=p, seed=seed, stream_id=stream_id)
    return dz, (dzd if p > 0 else dz)


def _input_grad(cx, x, dy2d, W, dz_res):
    """The last GEMM of a sub-layer's backward: dx = dz_res + dy2d @ W, accumulated into x -- or, when x is the output of a LayerNorm
    inside the stack whose only consumer this sub-layer is (x.ln, set by the stack loops) and nothing else has contributed to x.g, that
    LayerNorm's backward in the same launch (ops.linear_dgrad_lnbwd), its result left in x.pre for the producing sub-layer's closure."""
    ln = x.ln
    if ln is not None and x.g is None and x.pre is None:
        z, mean, rstd, pre_norm, p, seed, sid = ln
        st = cx.st
        dz = _empty(*z.shape, like=z)
        dzd = _empty(*z.shape, like=z) if p > 0 else None
        if ops.linear_dgrad_lnbwd(dy2d, W, dz_res, z, mean, rstd, cx.P[pre_norm + "weight"], dz, dzd, st.g(pre_norm + "weight"), st.g(pre_norm + "bias"),
                                  drop_p=p, seed=seed, stream_id=sid):
            x.pre = (dz, dzd)
            return
    dx = _empty(dy2d.shape[0], W.shape[1], like=dy2d)
    ops.linear_dgrad(dy2d, W, dx, R=dz_res)
    acc(x, dx)


def attn_sublayer(cx, tape, x, mem, lens_k, causal, pre_attn, pre_norm, B, Tq, Tk, H, drop, pad_free_grads=False):
    """y = LN(x + dropout(MHA(x, mem or x)))  with attention-probability dropout `drop` as well.
    x: Var [B*Tq, E]; mem: Var [B*Tk, E] or None for self-attention."""
    E = x.v.shape[1]
    Nq, Nk = B * Tq, B * Tk
    W, bias = cx.P[pre_attn + "in_proj_weight"], cx.P[pre_attn + "in_proj_bias"]
    Wo, bo = cx.P[pre_attn + "out_proj.weight"], cx.P[pre_attn + "out_proj.bias"]
    p = cx.p(drop)
    s_attn, s_out = cx.stream(), cx.stream()
    # Q / K / V (and dO below) are read by the attention kernels only: the projections store them pre-split (hi/lo bf16 chunks), so
    # the K/V tiles every query block stages -- and Q / dO in the backward -- need no fp32 -> hi/lo conversion there
    ps = config.ATTN_PRESPLIT
    if mem is None:
        qkv = _empty(Nq, 3 * E, like=x.v)
        ops.linear_fwd(x.v, W, bias, qkv, out_split=ps)
        Q, K, V = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    else:
        q = _empty(Nq, E, like=x.v)
        kv = _empty(Nk, 2 * E, like=x.v)
        ops.linear_fwd(x.v, W[:E], bias[:E], q, out_split=ps)
        ops.linear_fwd(mem.v, W[E:], bias[E:], kv, out_split=ps)
        Q, K, V = q, kv[:, :E], kv[:, E:]
    O = _empty(Nq, E, like=x.v)
    LSE = _empty(B, H, Tq, like=x.v)
    ops.attn_fwd(Q, K, V, O, LSE, lens_k, B, H, Tq, Tk, causal, drop_p=p, seed=cx.seed, stream_id=s_attn, qkv_split=ps)
    z = _empty(Nq, E, like=x.v)
    y, mean, rstd = _empty(Nq, E, like=x.v), _empty(Nq, like=x.v), _empty(Nq, like=x.v)
    # z = x + drop(O Wo^T + bo), y = LayerNorm(z): one launch where the row-panel GEMM serves the shape, GEMM + LayerNorm otherwise
    ops.linear_fwd(O, Wo, bo, z, drop_p=p, seed=cx.seed, stream_id=s_out, R=x.v, ln=(cx.P[pre_norm + "weight"], cx.P[pre_norm + "bias"], y, mean, rstd, LN_EPS))
    out = Var(y)
    if tape is not None:
        seed = cx.seed

        out.ln = (z, mean, rstd, pre_norm, p, seed, s_out)      # (used only if a stack loop declares the next sub-layer the sole consumer: _sole)

        def bwd():
            if out.g is None and out.pre is None:
                return
            with ops.wgrad_batch():            # out-proj + in-proj weight gradients: one grouped launch at the end
                bwd_body()

        def bwd_body():
            st = cx.st
            dz, da = _ln_backward(cx, out, z, mean, rstd, pre_norm, p, seed, s_out)
            gWo = st.g(pre_attn + "out_proj.weight")
            if gWo is not None:
                ops.linear_wgrad(da, O, gWo, db=st.g(pre_attn + "out_proj.bias"))
            dO = _empty(Nq, E, like=z)
            ops.linear_dgrad(da, Wo, dO, out_split=ps)
            delta = _empty(B, H, Tq, like=z)
            gW, gb = st.g(pre_attn + "in_proj_weight"), st.g(pre_attn + "in_proj_bias")
            if mem is None:
                dqkv = _empty(Nq, 3 * E, like=z)
                ops.attn_bwd(Q, K, V, O, dO, LSE, delta, dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:], lens_k, B, H, Tq, Tk, causal,
                             drop_p=p, seed=seed, stream_id=s_attn, qkv_split=ps, lens_q=lens_k if (pad_free_grads and config.ENC_SKIP_PAD_GRADS) else None)
                if gW is not None:
                    ops.linear_wgrad(dqkv, x.v, gW, db=gb)
                _input_grad(cx, x, dqkv, W, dz)
            else:
                dq = _empty(Nq, E, like=z)
                dkv = _empty(Nk, 2 * E, like=z)
                ops.attn_bwd(Q, K, V, O, dO, LSE, delta, dq, dkv[:, :E], dkv[:, E:], lens_k, B, H, Tq, Tk, causal,
                             drop_p=p, seed=seed, stream_id=s_attn, qkv_split=ps)
                if gW is not None:
                    ops.linear_wgrad(dq, x.v, gW[:E], db=gb[:E])
                    ops.linear_wgrad(dkv, mem.v, gW[E:], db=gb[E:])
                _input_grad(cx, x, dq, W[:E], dz)           # (contraction over 256: the K-streamed kernel does not serve it; plain path)
                dmem = _empty(Nk, E, like=z)
                ops.linear_dgrad(dkv, W[E:], dmem)
                acc(mem, dmem)
        tape.record(bwd)
    return out


def cross_attn_sublayer_pair(cx, tape, x2, mems, lens_ks, Tks, pre_attn, pre_norm, B, Tq, H, drop):
    """The cross-attention sub-layer of TWO decoder calls of one query shape and the same weights (the auto-encoder and the supervised
    sub-step's decoder, /root/reference/src/train.py:609-628): y = LN(x + dropout(MHA(x, mem_h))) for the two row blocks of x2 [2 B Tq, E]
    with their own memories mems[h] (Var [B Tk_h, E]).  Everything that does not see the memory runs once over both blocks -- the query
    projection, the out-projection + LayerNorm, their backward GEMMs and weight gradients; the K/V projections and the attention core
    run per block."""
    E = x2.v.shape[1]
    N = B * Tq
    W, bias = cx.P[pre_attn + "in_proj_weight"], cx.P[pre_attn + "in_proj_bias"]
    Wo, bo = cx.P[pre_attn + "out_proj.weight"], cx.P[pre_attn + "out_proj.bias"]
    p = cx.p(drop)
    s_attn, s_out = (cx.stream(), cx.stream()), cx.stream()
    ps = config.ATTN_PRESPLIT
    q2 = _empty(2 * N, E, like=x2.v)
    ops.linear_fwd(x2.v, W[:E], bias[:E], q2, out_split=ps)
    kvs = []
    for h in range(2):
        kv = _empty(B * Tks[h], 2 * E, like=x2.v)
        ops.linear_fwd(mems[h].v, W[E:], bias[E:], kv, out_split=ps)
        kvs.append(kv)
    O2 = _empty(2 * N, E, like=x2.v)
    LSE2 = _empty(2, B, H, Tq, like=x2.v)
    for h in range(2):
        ops.attn_fwd(q2[h * N:(h + 1) * N], kvs[h][:, :E], kvs[h][:, E:], O2[h * N:(h + 1) * N], LSE2[h], lens_ks[h], B, H, Tq, Tks[h], False,
                     drop_p=p, seed=cx.seed, stream_id=s_attn[h], qkv_split=ps)
    z = _empty(2 * N, E, like=x2.v)
    y, mean, rstd = _empty(2 * N, E, like=x2.v), _empty(2 * N, like=x2.v), _empty(2 * N, like=x2.v)
    ops.linear_fwd(O2, Wo, bo, z, drop_p=p, seed=cx.seed, stream_id=s_out, R=x2.v, ln=(cx.P[pre_norm + "weight"], cx.P[pre_norm + "bias"], y, mean, rstd, LN_EPS))
    out = Var(y)
    if tape is not None:
        seed = cx.seed

        out.ln = (z, mean, rstd, pre_norm, p, seed, s_out)

        def bwd():
            if out.g is None and out.pre is None:
                return
            with ops.wgrad_batch():
                st = cx.st
                dz, da = _ln_backward(cx, out, z, mean, rstd, pre_norm, p, seed, s_out)
                gWo = st.g(pre_attn + "out_proj.weight")
                if gWo is not None:
                    ops.linear_wgrad(da, O2, gWo, db=st.g(pre_attn + "out_proj.bias"))
                dO = _empty(2 * N, E, like=z)
                ops.linear_dgrad(da, Wo, dO, out_split=ps)
                gW, gb = st.g(pre_attn + "in_proj_weight"), st.g(pre_attn + "in_proj_bias")
                dq2 = _empty(2 * N, E, like=z)
                dkvs = []
                for h in range(2):
                    delta = _empty(B, H, Tq, like=z)
                    dkv = _empty(B * Tks[h], 2 * E, like=z)
                    ops.attn_bwd(q2[h * N:(h + 1) * N], kvs[h][:, :E], kvs[h][:, E:], O2[h * N:(h + 1) * N], dO[h * N:(h + 1) * N], LSE2[h], delta,
                                 dq2[h * N:(h + 1) * N], dkv[:, :E], dkv[:, E:], lens_ks[h], B, H, Tq, Tks[h], False,
                                 drop_p=p, seed=seed, stream_id=s_attn[h], qkv_split=ps)
                    dkvs.append(dkv)
                    dmem = _empty(B * Tks[h], E, like=z)
                    ops.linear_dgrad(dkv, W[E:], dmem)
                    acc(mems[h], dmem)
                if gW is not None:
                    ops.linear_wgrad(dkvs[0], mems[0].v, gW[E:], db=gb[E:])
                    ops.linear_wgrad(dq2, x2.v, gW[:E], db=gb[:E])
                _input_grad(cx, x2, dq2, W[:E], dz)          # (contraction over 256: plain path)
            # the second memory's K/V weight gradient accumulates into the SAME rows of in_proj_weight as the first's: the problems of a
            # grouped launch run side by side, so it goes out on its own, behind the group (same stream: ops keeps a gradient buffer's stream)
            gW = cx.st.g(pre_attn + "in_proj_weight")
            if gW is not None:
                ops.linear_wgrad(dkvs[1], mems[1].v, gW[E:], db=cx.st.g(pre_attn + "in_proj_bias")[E:])
        tape.record(bwd)
    return out


def decoder_stack_pair(cx, tape, x2, lens_q2, mems, lens_ks, Tks, pre, L, B, Tq, H, drop):
    """decoder_stack over two calls of one query shape at once: causal self-attention and the feed-forward block over 2B sequences, the
    cross-attention per call on its own memory (cross_attn_sublayer_pair)."""
    for i in range(L):
        lp = "%s%d." % (pre, i)
        x2 = attn_sublayer(cx, tape, x2, None, lens_q2, True, lp + "self_attn.", lp + "norm1.", 2 * B, Tq, Tq, H, drop)
        x2 = cross_attn_sublayer_pair(cx, tape, x2, mems, lens_ks, Tks, lp + "multihead_attn.", lp + "norm2.", B, Tq, H, drop)
        x2 = ffn_sublayer(cx, tape, x2, lp, lp + "norm3.", drop)
    x2.ln = None                # the stack's output has consumers outside it: its LayerNorm backward stays a launch of its own
    return x2


def speech_decode_pair(cx, tape, m, mels, lens_q2, mems, lens_ks, Tks, loss_hints):
    """SpeechTransformer.decode_sequence of TWO calls of one target shape (the auto-encoder's and the TTS decoder of one generator phase):
    front ends per call into one buffer, the decoder stack once over both (decoder_stack_pair), heads + post-net + loss terms per call
    (the post-net's BatchNorm statistics stay per call, first a, then b).  Returns [(head Var, post Var)] per call."""
    B, T, M = mels[0].shape
    a = m.args
    N = B * T
    E = cx.P["speech_m.prenet.layer.fc2.linear_layer.weight"].shape[0]
    buf = _empty(2 * N, E, like=mels[0])
    halves = [speech_decode_front(cx, tape, m, mels[h], True, out=buf[h * N:(h + 1) * N]) for h in range(2)]
    x2 = _stack_rows(cx, tape, halves, buf)
    y2 = decoder_stack_pair(cx, tape, x2, lens_q2, mems, lens_ks, Tks, "speech_m.decoder.transformer_decoder.layers.", a.num_layers, B, T, a.nhead, a.d_drop)
    # the two tails read row blocks of y2 and put their d(loss)/dx straight into row blocks of ONE gradient buffer
    gbuf = {}
    xs = [Var(y2.v[h * N:(h + 1) * N]) for h in range(2)]
    if tape is not None:
        def join():                                   # recorded before the tails => runs after them: hand the buffer to the stack
            if "g" in gbuf:
                for h in range(2):
                    if xs[h].g is None:
                        gbuf["g"][h * N:(h + 1) * N].zero_()
                    elif xs[h].g.data_ptr() != gbuf["g"][h * N:(h + 1) * N].data_ptr():
                        ops.sum2(gbuf["g"][h * N:(h + 1) * N], xs[h].g.contiguous())
                acc(y2, gbuf["g"])
        tape.record(join)

    def dx_block(h):
        def get():
            if "g" not in gbuf:
                gbuf["g"] = _empty(2 * N, E, like=y2.v)
            return gbuf["g"][h * N:(h + 1) * N]
        return get
    outs = [speech_decode_tail(cx, tape, m, xs[h], mels[h], postnet=True, loss_hint=loss_hints[h], dx_out=dx_block(h)) for h in range(2)]
    return outs


def ffn_sublayer(cx, tape, x, pre, pre_norm, drop):
    """y = LN(x + dropout(W2 dropout(relu(W1 x + b1)) + b2))."""
    N, E = x.v.shape
    W1, b1, W2, b2 = cx.P[pre + "linear1.weight"], cx.P[pre + "linear1.bias"], cx.P[pre + "linear2.weight"], cx.P[pre + "linear2.bias"]
    F = W1.shape[0]
    p = cx.p(drop)
    s1, s2 = cx.stream(), cx.stream()
    h = _empty(N, F, like=x.v)
    # keep bits of h (relu > 0 and not dropped) for the backward's gate, when both GEMMs that touch them run on the row-panel kernel
    bits = None
    if tape is not None and config.PANEL_GATE_BITS and F % 64 == 0 and F <= 1024 and ops.panel_serves(N, E, W1) and ops.panel_serves(N, E, W2, transposed=True):
        bits = torch.empty(ops.gate_bits_bytes(N, F), dtype=torch.uint8, device=x.v.device)
    ops.linear_fwd(x.v, W1, b1, h, act=1, drop_p=p, seed=cx.seed, stream_id=s1, gate_bits=bits)
    z = _empty(N, E, like=x.v)
    y, mean, rstd = _empty(N, E, like=x.v), _empty(N, like=x.v), _empty(N, like=x.v)
    ops.linear_fwd(h, W2, b2, z, drop_p=p, seed=cx.seed, stream_id=s2, R=x.v, ln=(cx.P[pre_norm + "weight"], cx.P[pre_norm + "bias"], y, mean, rstd, LN_EPS))
    out = Var(y)
    if tape is not None:
        seed = cx.seed

        out.ln = (z, mean, rstd, pre_norm, p, seed, s2)

        def bwd():
            if out.g is None and out.pre is None:
                return
            with ops.wgrad_batch():            # linear2 + linear1 weight gradients: one grouped launch at the end
                bwd_body()

        def bwd_body():
            st = cx.st
            dz, da = _ln_backward(cx, out, z, mean, rstd, pre_norm, p, seed, s2)
            g2 = st.g(pre + "linear2.weight")
            if g2 is not None:
                ops.linear_wgrad(da, h, g2, db=st.g(pre + "linear2.bias"))
            du = _empty(N, F, like=z)
            gsc = 1.0 / (1.0 - p) if p > 0 else 1.0
            if bits is not None:
                ops.linear_dgrad(da, W2, du, gate_bits=bits, gate_scale=gsc)                      # relu' and dropout mask from the keep bits
            else:
                ops.linear_dgrad(da, W2, du, G=h, gate_scale=gsc)                                 # ... from h > 0
            g1 = st.g(pre + "linear1.weight")
            if g1 is not None:
                ops.linear_wgrad(du, x.v, g1, db=st.g(pre + "linear1.bias"))
            _input_grad(cx, x, du, W1, dz)
        tape.record(bwd)
    return out


def encoder_stack(cx, tape, x, lens, pre, L, B, T, H, drop):
    """src/module.py:270-280 (TransformerEncoder): all-False attn mask + key padding mask."""
    for i in range(L):
        lp = "%s%d." % (pre, i)
        x = attn_sublayer(cx, tape, x, None, lens, False, lp + "self_attn.", lp + "norm1.", B, T, T, H, drop, pad_free_grads=True)
        x = ffn_sublayer(cx, tape, x, lp, lp + "norm2.", drop)
    x.ln = None                 # the stack's output has consumers outside it (decoder memory, discriminator): plain LayerNorm backward
    return x


def decoder_stack(cx, tape, x, lens_q, mem, lens_k, pre, L, B, Tq, Tk, H, drop):
    """src/module.py:283-293 (TransformerDecoder): causal + tgt padding self-attention, memory-padding cross-attention."""
    for i in range(L):
        lp = "%s%d." % (pre, i)
        x = attn_sublayer(cx, tape, x, None, lens_q, True, lp + "self_attn.", lp + "norm1.", B, Tq, Tq, H, drop)
        x = attn_sublayer(cx, tape, x, mem, lens_k, False, lp + "multihead_attn.", lp + "norm2.", B, Tq, Tk, H, drop)
        x = ffn_sublayer(cx, tape, x, lp, lp + "norm3.", drop)
    x.ln = None
    return x


# ---------------------------------------------------------------------------------------------------------------
# Positional encoding (src/module.py:249-267; dropout fixed at 0.1)
# ---------------------------------------------------------------------------------------------------------------
def posenc(cx, tape, x, pe, T, gate=None, out=None):
    """out: an [N, Dm] row block of a larger buffer that receives the result (the two halves of a paired encoder call)."""
    N, Dm = x.v.shape
    p = cx.p(0.1)
    s = cx.stream()
    y = out if out is not None else _empty(N, Dm, like=x.v)
    scale = math.sqrt(Dm)
    ops.posenc_fwd(x.v, pe, y, T, scale, drop_p=p, seed=cx.seed, stream_id=s)
    out = Var(y)
    if tape is not None:
        seed = cx.seed

        def bwd():
            if out.g is None:
                return
            dx = _empty(N, Dm, like=y)
            ops.posenc_bwd(out.g, gate, dx, scale, drop_p=p, seed=seed, stream_id=s)
            acc(x, dx)
        tape.record(bwd)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Text side
# ---------------------------------------------------------------------------------------------------------------
class ZeroPool:
    """fp64 scratch of a whole conv + BatchNorm stack, zeroed by ONE fill launch: per stage 2C doubles for the conv GEMM's column
    sums (forward) and 2C for the backward's reduction -- instead of a fill per stage and pass (17 + 14 launches per step)."""

    def __init__(self, stages, C, device):
        self.buf = torch.zeros(stages * 4 * C, dtype=torch.float64, device=device)
        self.C, self.i = C, 0

    def take(self, Cout):
        if Cout != self.C or (self.i + 1) * 4 * self.C > self.buf.numel():
            return torch.zeros(4 * Cout, dtype=torch.float64, device=self.buf.device)
        w = self.buf[self.i * 4 * self.C:(self.i + 1) * 4 * self.C]
        self.i += 1
        return w


def conv_bn_act(cx, tape, x, B, T, conv_pre, bn_pre, pad_left, act, drop, bn_buffers, residual=None, x_ld_view=None, pool=None):
    """dropout(act(BN_train(conv1d_k5(x)))) — one stage of TextPrenet.forward_fcn / SpeechPostnet.forward
    (src/module.py:162-165, 223-230).  x: Var [B*T, Cin]."""
    Wp, b = cx.P[conv_pre + "conv.weight"], cx.P[conv_pre + "conv.bias"]
    Cout, _, Cin = Wp.shape
    x3 = x.v.view(B, T, -1)[..., :Cin] if x.v.shape[1] != Cin else x.v.view(B, T, Cin)
    c = _empty(B, T, Cout, like=x.v)
    fused_stats = cx.training and config.CONV_BN_STATS
    # [0, 2C): sum / sum of squares per channel out of the conv GEMM's epilogue; [2C, 4C): the backward's column sums (both pre-zeroed)
    ws4 = pool.take(Cout) if (pool is not None and cx.training) else torch.zeros(4 * Cout, dtype=torch.float64, device=x.v.device)
    ws, ws_b = ws4[:2 * Cout], ws4[2 * Cout:]
    ops.conv_fwd(x3, Wp, b, c, pad_left, colstats=ws if fused_stats else None)
    p = cx.p(drop)
    s = cx.stream()
    N = B * T
    y = _empty(N, Cout, like=x.v)
    mean, rstd = _empty(Cout, like=x.v), _empty(Cout, like=x.v)
    rm, rv = bn_buffers[bn_pre + "running_mean"], bn_buffers[bn_pre + "running_var"]
    gamma, beta = cx.P[bn_pre + "weight"], cx.P[bn_pre + "bias"]
    if cx.training:
        ops.bn_fwd(c.view(N, Cout), gamma, beta, y, mean, rstd, rm, rv, ws, act, drop_p=p, seed=cx.seed, stream_id=s, have_sums=fused_stats,
                   num_batches_tracked=bn_buffers[bn_pre + "num_batches_tracked"])
    else:
        if tape is not None:
            raise NotImplementedError("eval-mode BatchNorm has no backward on this path; evaluate() runs under torch.no_grad()")
        ops.bn_eval_fwd(c.view(N, Cout), gamma, beta, rm, rv, y, mean, rstd, act)
    out = Var(y)
    if tape is not None:
        seed = cx.seed

        def bwd():
            if out.g is None:
                return
            st = cx.st
            dy = out.g if out.g.is_contiguous() else out.g.contiguous()
            dc = _empty(N, Cout, like=y)
            ops.bn_bwd(dy, c.view(N, Cout), mean, rstd, gamma, beta, dc, st.g(bn_pre + "weight"), st.g(bn_pre + "bias"), ws_b, act,
                       drop_p=p, seed=seed, stream_id=s, ws_zeroed=True)
            gW = st.g(conv_pre + "conv.weight")
            dc3 = dc.view(B, T, Cout)
            if gW is not None:
                ops.conv_wgrad(dc3, x3, gW, pad_left, db=st.g(conv_pre + "conv.bias"))
            dx = _empty(B, T, Cin, like=y)
            ops.conv_dgrad(dc3, Wp, dx, pad_left)
            acc(x, dx.view(N, Cin))
        tape.record(bwd)
    return out


def text_embed(cx, tape, ids, T, drop, noise, shift_sos):
    """prenet.emb_dropout(prenet.embed(ids)) [+ noise_fn]  (src/network.py:427-438); shift_sos builds [SOS, ids[:-1]]."""
    E = cx.P["text_m.prenet.embed.weight"]
    N = ids.numel()
    p = cx.p(drop)
    pn = 0.3 if (noise and cx.noisy) else 0.0
    s, sn = cx.stream(), cx.stream()
    y = _empty(N, E.shape[1], device=E.device)
    ops.embed_fwd(ids, E, y, T, shift_sos=shift_sos, drop_p=p, seed=cx.seed, stream_id=s, noise_p=pn, noise_stream=sn)
    out = Var(y)
    if tape is not None:
        seed = cx.seed

        def bwd():
            gE = cx.st.g("text_m.prenet.embed.weight")
            if out.g is None or gE is None:
                return
            ops.embed_bwd(ids, out.g, gE, T, shift_sos=shift_sos, drop_p=p, seed=seed, stream_id=s, noise_p=pn, noise_stream=sn)
        tape.record(bwd)
    return out


def text_frontend(cx, tape, m, ids, noise, out=None):
    """Embedding (+ noise_fn), the three conv + BatchNorm + ReLU stages and the positional encoding in front of the text encoder stack
    (src/network.py:427-434): everything of TextTransformer.encode that sees the batch as a whole (BatchNorm statistics)."""
    B, T = ids.shape
    a = m.args
    x = text_embed(cx, tape, ids, T, a.t_pre_drop, noise, -1)
    pool = ZeroPool(3, cx.P["text_m.prenet.conv1.conv.weight"].shape[0], ids.device) if cx.training else None
    for i in (1, 2, 3):
        x = conv_bn_act(cx, tape, x, B, T, "text_m.prenet.conv%d." % i, "text_m.prenet.batch_norm%d." % i, 2, 1, a.t_pre_drop, m.buffers_dict, pool=pool)
    return posenc(cx, tape, x, m.pe, T, out=out)


def text_encode(cx, tape, m, ids, lens, noise):
    """TextTransformer.encode (src/network.py:427-444)."""
    B, T = ids.shape
    a = m.args
    x = text_frontend(cx, tape, m, ids, noise)
    return encoder_stack(cx, tape, x, lens, "text_m.encoder.transformer_encoder.layers.", a.num_layers, B, T, a.nhead, a.e_drop)


def _stack_rows(cx, tape, halves, buf):
    """Var over `buf` [2N, E], whose two row blocks the front ends of a paired encoder call have just written (posenc out=): the
    gradient that comes back for the whole buffer is handed to the halves as row-block views (no copy either way)."""
    x2 = Var(buf)
    if tape is not None:
        N = halves[0].v.shape[0]

        def bwd():
            if x2.g is None:
                return
            for i, h in enumerate(halves):
                acc(h, x2.g[i * N:(i + 1) * N])
        tape.record(bwd)                                # recorded after the halves' own closures => runs before them
    return x2


def text_encode_pair(cx, tape, m, ids_a, noise_a, ids_b, noise_b, lens2):
    """TextTransformer.encode of TWO batches of one shape (the auto-encoder and the supervised sub-step of one generator phase, same
    weights: /root/reference/src/train.py:609-628) with the encoder stack run ONCE over 2B sequences.  What the reference keeps per call
    stays per half: the conv front end with its BatchNorm batch statistics and running-stat updates (first a, then b), the dropout /
    noise streams.  Returns the stack's output Var [2 B T, E] (rows of a, then rows of b)."""
    B, T = ids_a.shape
    a = m.args
    E = cx.P["text_m.prenet.embed.weight"].shape[1]
    buf = _empty(2 * B * T, E, device=ids_a.device)
    xa = text_frontend(cx, tape, m, ids_a, noise_a, out=buf[:B * T])
    xb = text_frontend(cx, tape, m, ids_b, noise_b, out=buf[B * T:])
    x2 = _stack_rows(cx, tape, (xa, xb), buf)
    return encoder_stack(cx, tape, x2, lens2, "text_m.encoder.transformer_encoder.layers.", a.num_layers, 2 * B, T, a.nhead, a.e_drop)


FUSED_STATS = {"text_head": 0, "text_grad_direct": 0, "text_grad_general": 0, "speech_head": 0, "speech_grad_direct": 0, "speech_grad_general": 0}      # launches so far (tests)
FUSED_LOSSES = {}      # storage address of a head's output buffer -> what the fused head + loss launch left for the loss call (train.text_loss / speech_loss)


def text_decode(cx, tape, m, ids, lens_q, mem, lens_k, Tk, shift=True, loss_hint=None):
    """TextTransformer.decode_sequence (src/network.py:483-493) incl. TextPostnet (src/module.py:233-246).
    Returns Var logits buffer [B*T, 48] (46 valid columns).  shift=False: `ids` are the decoder inputs as they are
    (TextTransformer.decode, src/network.py:446-450)."""
    B, T = ids.shape
    a = m.args
    x = text_embed(cx, tape, ids, T, a.t_pre_drop, False, 1 if shift else -1)          # SOS_IDX = 1
    x = posenc(cx, tape, x, m.pe, T)
    x = decoder_stack(cx, tape, x, lens_q, mem, lens_k, "text_m.decoder.transformer_decoder.layers.", a.num_layers, B, T, Tk, a.nhead, a.d_drop)
    return text_decode_tail(cx, tape, m, x, loss_hint=loss_hint)


def text_decode_pair(cx, tape, m, ids2, lens_q2, mems, lens_ks, Tks, loss_hints):
    """TextTransformer.decode_sequence of TWO calls of one target shape (the auto-encoder's text decoder and the ASR decoder of one generator
    phase): embeddings + positional encoding per call into one buffer, the decoder stack once over both (decoder_stack_pair), TextPostnet
    head + loss per call.  Returns the two logits Vars."""
    B, T = ids2[0].shape
    a = m.args
    N = B * T
    E = cx.P["text_m.prenet.embed.weight"].shape[1]
    buf = _empty(2 * N, E, device=ids2[0].device)
    halves = [posenc(cx, tape, text_embed(cx, tape, ids2[h], T, a.t_pre_drop, False, 1), m.pe, T, out=buf[h * N:(h + 1) * N]) for h in range(2)]
    x2 = _stack_rows(cx, tape, halves, buf)
    y2 = decoder_stack_pair(cx, tape, x2, lens_q2, mems, lens_ks, Tks, "text_m.decoder.transformer_decoder.layers.", a.num_layers, B, T, a.nhead, a.d_drop)
    gbuf = {}
    xs = [Var(y2.v[h * N:(h + 1) * N]) for h in range(2)]
    if tape is not None:
        def join():                                   # recorded before the tails => runs after them
            if "g" in gbuf:
                for h in range(2):
                    blk = gbuf["g"][h * N:(h + 1) * N]
                    if xs[h].g is None:
                        blk.zero_()
                    elif xs[h].g.data_ptr() != blk.data_ptr():
                        ops.sum2(blk, xs[h].g.contiguous())
                acc(y2, gbuf["g"])
        tape.record(join)

    def dx_block(h):
        def get():
            if "g" not in gbuf:
                gbuf["g"] = _empty(2 * N, E, like=y2.v)
            return gbuf["g"][h * N:(h + 1) * N]
        return get
    return [text_decode_tail(cx, tape, m, xs[h], loss_hint=loss_hints[h], dx_out=dx_block(h)) for h in range(2)]


def text_decode_tail(cx, tape, m, x, loss_hint=None, dx_out=None):
    """TextPostnet (dropout + fc1, src/module.py:233-246) on decoder states x (Var [B*T, E]), with the loss in the head's launch when the
    step announced it.  dx_out: where the backward puts d(loss)/dx (a row block of a paired call's gradient buffer)."""
    a = m.args
    N, E = x.v.shape
    p = cx.p(a.t_post_drop)
    s = cx.stream()
    if p > 0:
        xd = _empty(N, E, like=x.v)
        ops.leaky_dropout(x.v, None, xd, 1.0, drop_p=p, seed=cx.seed, stream_id=s)
    else:
        xd = x.v
    W, b = cx.P["text_m.postnet.fc1.weight"], cx.P["text_m.postnet.fc1.bias"]
    V = W.shape[0]
    ldl = (V + 3) // 4 * 4
    if loss_hint is not None and tape is not None and E == 256 and V <= 48 and config.FUSED_HEAD_LOSS:
        # loss_hint = (gold [B, T] int64, eos_weight, gscale, loss workspace): head GEMM, cross-entropy and its gradient in one launch
        gold, eos_w, gscale, ws = loss_hint
        logits = torch.empty(N, ldl, dtype=torch.float32, device=x.v.device)
        dlogits = torch.empty(N, ldl, dtype=torch.float32, device=x.v.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.v.device)
        goldc = gold.contiguous().view(-1)
        ops.text_head_loss(xd, W, b, goldc, V, eos_w, gscale, logits, dlogits, ws, loss)
        FUSED_STATS["text_head"] += 1
        FUSED_LOSSES[logits.untyped_storage().data_ptr()] = dict(kind="text", loss=loss, dlogits=dlogits, gold=goldc, eos_weight=float(eos_w), gscale=float(gscale), ws=ws)
    else:
        logits = torch.zeros(N, ldl, dtype=torch.float32, device=x.v.device)
        ops.linear_fwd(xd, W, b, logits[:, :V])
    out = Var(logits)
    if tape is not None:
        seed = cx.seed

        def bwd():
            if out.g is None:
                return
            st = cx.st
            dl = out.g
            gW = st.g("text_m.postnet.fc1.weight")
            if gW is not None:
                ops.linear_wgrad(dl[:, :V], xd, gW, db=st.g("text_m.postnet.fc1.bias"))
            last = dx_out() if dx_out is not None else None
            dxd = last if (last is not None and p <= 0) else _empty(N, E, like=xd)
            ops.linear_dgrad(dl[:, :V], W, dxd)
            if p > 0:
                dx = last if last is not None else _empty(N, E, like=xd)
                ops.leaky_dropout(x.v, dxd, dx, 1.0, drop_p=p, seed=seed, stream_id=s)
            else:
                dx = dxd
            acc(x, dx)
        tape.record(bwd)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Speech side
# ---------------------------------------------------------------------------------------------------------------
def speech_prenet(cx, tape, m, mel2d, T, out=None):
    """SpeechPrenet (src/module.py:76-110; ONE dropout) followed by PositionalEncoding."""
    a = m.args
    N = mel2d.shape[0]
    W1, b1 = cx.P["speech_m.prenet.layer.fc1.linear_layer.weight"], cx.P["speech_m.prenet.layer.fc1.linear_layer.bias"]
    W2, b2 = cx.P["speech_m.prenet.layer.fc2.linear_layer.weight"], cx.P["speech_m.prenet.layer.fc2.linear_layer.bias"]
    p = cx.p(a.s_pre_drop)
    s = cx.stream()
    h1 = _empty(N, W1.shape[0], like=mel2d)
    ops.linear_fwd(mel2d, W1, b1, h1, act=1, drop_p=p, seed=cx.seed, stream_id=s)
    h2 = _empty(N, W2.shape[0], like=mel2d)
    ops.linear_fwd(h1, W2, b2, h2, act=1)
    h2v = Var(h2)
    if tape is not None:
        def bwd():
            st = cx.st
            d2 = h2v.g                                  # already gated by relu'(h2) in posenc's backward
            g2 = st.g("speech_m.prenet.layer.fc2.linear_layer.weight")
            if d2 is None or g2 is None:
                return
            ops.linear_wgrad(d2, h1, g2, db=st.g("speech_m.prenet.layer.fc2.linear_layer.bias"))
            du = _empty(N, W1.shape[0], like=h1)
            ops.linear_dgrad(d2, W2, du, G=h1, gate_scale=(1.0 / (1.0 - p) if p > 0 else 1.0))
            ops.linear_wgrad(du, mel2d, st.g("speech_m.prenet.layer.fc1.linear_layer.weight"), db=st.g("speech_m.prenet.layer.fc1.linear_layer.bias"))
        tape.record(bwd)                                # recorded before posenc's closure => runs after it
    y = posenc(cx, tape, h2v, m.pe, T, gate=h2, out=out)
    return y


def speech_frontend(cx, tape, m, mel, noise, out=None):
    """noise_fn on the raw mel, SpeechPrenet and the positional encoding in front of the speech encoder stack (src/network.py:203-206)."""
    B, T, M = mel.shape
    mel2d = mel.reshape(B * T, M)
    if noise and cx.noisy:
        noised = _empty(B * T, M, like=mel2d)
        ops.rowmask(mel2d, noised, 0.3, cx.seed, cx.stream())
        mel2d = noised
    return speech_prenet(cx, tape, m, mel2d, T, out=out)


def speech_encode(cx, tape, m, mel, lens, noise):
    """SpeechTransformer.encode (src/network.py:203-208)."""
    B, T, M = mel.shape
    a = m.args
    x = speech_frontend(cx, tape, m, mel, noise)
    return encoder_stack(cx, tape, x, lens, "speech_m.encoder.transformer_encoder.layers.", a.num_layers, B, T, a.nhead, a.e_drop)


def speech_encode_pair(cx, tape, m, mel_a, noise_a, mel_b, noise_b, lens2):
    """SpeechTransformer.encode of two batches of one shape with the encoder stack run once over 2B sequences (see text_encode_pair; the
    speech front end has no batch statistics, its dropout / noise streams stay per half)."""
    B, T, M = mel_a.shape
    a = m.args
    E = cx.P["speech_m.prenet.layer.fc2.linear_layer.weight"].shape[0]
    buf = _empty(2 * B * T, E, like=mel_a)
    xa = speech_frontend(cx, tape, m, mel_a, noise_a, out=buf[:B * T])
    xb = speech_frontend(cx, tape, m, mel_b, noise_b, out=buf[B * T:])
    x2 = _stack_rows(cx, tape, (xa, xb), buf)
    return encoder_stack(cx, tape, x2, lens2, "speech_m.encoder.transformer_encoder.layers.", a.num_layers, 2 * B, T, a.nhead, a.e_drop)


def speech_decode_front(cx, tape, m, mel, shift=True, out=None):
    """The decoder input of SpeechTransformer.decode_sequence: go-frame shift (src/network.py:254-262), SpeechPrenet, positional encoding."""
    B, T, M = mel.shape
    N = B * T
    if shift:
        if M % 4 == 0 and mel.is_contiguous() and mel.data_ptr() % 16 == 0:
            tgt = ops.shift_frames(mel, torch.empty_like(mel))      # [zero frame, mel[:-1]] in one launch
        else:
            tgt = torch.zeros_like(mel)
            tgt[:, 1:] = mel[:, :-1]
    else:
        tgt = mel.contiguous()
    return speech_prenet(cx, tape, m, tgt.view(N, M), T, out=out)


def speech_decode(cx, tape, m, mel, lens_q, mem, lens_k, Tk, shift=True, postnet=True, loss_hint=None):
    """SpeechTransformer.decode_sequence (src/network.py:254-269) + SpeechPostnet (src/module.py:155-171).
    Returns (head Var [B*T, 84]: cols 0..79 pre-net mel, col 80 stop logit; post Var [B*T, 80]).
    shift=False feeds `mel` as the decoder input as it is, postnet=False stops at the heads (SpeechTransformer.decode,
    src/network.py:210-214; no tape)."""
    B, T, M = mel.shape
    a = m.args
    x = speech_decode_front(cx, tape, m, mel, shift)
    x = decoder_stack(cx, tape, x, lens_q, mem, lens_k, "speech_m.decoder.transformer_decoder.layers.", a.num_layers, B, T, Tk, a.nhead, a.d_drop)
    return speech_decode_tail(cx, tape, m, x, mel, postnet=postnet, loss_hint=loss_hint)


def speech_decode_tail(cx, tape, m, x, mel, postnet=True, loss_hint=None, dx_out=None):
    """The heads ([linear_project | stop_linear]) and the SpeechPostnet on decoder states x (Var [B*T, E]); mel only gives shapes / device.
    dx_out: where the backward puts d(loss)/dx (a row block of a paired call's gradient buffer) instead of a buffer of its own."""
    B, T, M = mel.shape
    a = m.args
    N = B * T
    E = x.v.shape[1]
    st = cx.st
    Wh = st.span("speech_m.postnet.linear_project.weight", "speech_m.postnet.stop_linear.weight", (M + 1, E))
    bh = st.span("speech_m.postnet.linear_project.bias", "speech_m.postnet.stop_linear.bias", (M + 1,))
    ldh = (M + 1 + 3) // 4 * 4
    fused = None
    if loss_hint is not None and tape is not None and postnet and E == 256 and M % 4 == 0 and M + 1 <= 96 and config.FUSED_HEAD_LOSS and \
            Wh.data_ptr() % 16 == 0 and bh.data_ptr() % 16 == 0:
        # loss_hint = (gold mel [B, T, M], lengths int32 [B], eos_weight, gscale, loss workspace): the heads, the pre-net MSE, the stop BCE and
        # their gradient in one launch; the post-net term and the scalar follow behind the post-net (speech_post_loss below)
        gold, glens, eos_w, gscale, ws = loss_hint
        goldc = gold.contiguous()
        head = torch.empty(N, ldh, dtype=torch.float32, device=mel.device)
        d_head = torch.empty(N, ldh, dtype=torch.float32, device=mel.device)
        ops.speech_head_loss(x.v, Wh, bh, goldc.view(N, M), glens, B, T, M, eos_w, gscale, head, d_head, ws)
        FUSED_STATS["speech_head"] += 1
        fused = dict(kind="speech", gold=goldc, lens=glens, eos_weight=float(eos_w), gscale=float(gscale), ws=ws, d_head=d_head)
    else:
        head = torch.zeros(N, ldh, dtype=torch.float32, device=mel.device)
        ops.linear_fwd(x.v, Wh, bh, head[:, :M + 1])
    headv = Var(head)
    pre = Var(head)                                                 # postnet input = columns [0,M) of head (row stride ldh)
    postv = Var(None)
    if tape is not None:
        def bwd_head():
            # d(head) = loss part (headv.g) + postnet-input part (pre.g, [N,M]) + residual part of post = pre + postnet(pre)
            dh = headv.g
            if dh is None:
                dh = torch.zeros(N, ldh, dtype=torch.float32, device=mel.device)
            if pre.g is not None:
                ops.add_strided(dh, pre.g, M)
            if postv.g is not None:
                ops.add_strided(dh, postv.g, M)
            gW = st.gspan("speech_m.postnet.linear_project.weight", "speech_m.postnet.stop_linear.weight", (M + 1, E))
            if gW is not None:
                ops.linear_wgrad(dh[:, :M + 1], x.v, gW, db=st.gspan("speech_m.postnet.linear_project.bias", "speech_m.postnet.stop_linear.bias", (M + 1,)))
            dx = dx_out() if dx_out is not None else _empty(N, E, like=x.v)
            ops.linear_dgrad(dh[:, :M + 1], Wh, dx)
            acc(x, dx)
        tape.record(bwd_head)                                       # runs after every postnet closure
    if not postnet:
        return headv, None
    pool = ZeroPool(4, cx.P["speech_m.postnet.conv1.conv.weight"].shape[0], mel.device) if cx.training else None
    y = conv_bn_act(cx, tape, pre, B, T, "speech_m.postnet.conv1.", "speech_m.postnet.pre_batchnorm.", 4, 2, a.s_post_drop, m.buffers_dict, pool=pool)
    for i in range(3):
        y = conv_bn_act(cx, tape, y, B, T, "speech_m.postnet.conv_list.%d." % i, "speech_m.postnet.batch_norm_list.%d." % i, 4, 2,
                        a.s_post_drop, m.buffers_dict, pool=pool)
    Wp2, b2 = cx.P["speech_m.postnet.conv2.conv.weight"], cx.P["speech_m.postnet.conv2.conv.bias"]
    post = _empty(B, T, M, like=mel)
    C = y.v.shape[1]
    ops.gemm(ops.OP_KC_CONV, ops.OP_KC, y.v, C, Wp2, 5 * C, post, M, N, M, 5 * C, conv=(T, C, 0, 4), bias=b2, R=head, ldr=ldh)   # + residual pre
    postv.v = post.view(N, M)
    if fused is not None:
        fused["d_post"] = torch.empty(B, T, M, dtype=torch.float32, device=mel.device)
        fused["loss"] = torch.empty(1, dtype=torch.float32, device=mel.device)
        ops.speech_post_loss(fused["gold"].view(N, M), postv.v, fused["lens"], B, T, M, fused["gscale"], fused["d_post"], fused["ws"], fused["loss"])
        FUSED_LOSSES[head.untyped_storage().data_ptr()] = fused
    if tape is not None:
        def bwd_conv2():
            if postv.g is None:
                return
            dp = postv.g if postv.g.is_contiguous() else postv.g.contiguous()
            postv.g = dp
            gW = st.g("speech_m.postnet.conv2.conv.weight")
            dp3 = dp.view(B, T, M)
            if gW is not None:
                ops.conv_wgrad(dp3, y.v.view(B, T, C), gW, 4, db=st.g("speech_m.postnet.conv2.conv.bias"))
            dy = _empty(B, T, C, like=dp)
            ops.conv_dgrad(dp3, Wp2, dy, 4)
            acc(y, dy.view(N, C))
        tape.record(bwd_conv2)                                      # recorded last => runs first
    return headv, postv


# ---------------------------------------------------------------------------------------------------------------
# LSTM discriminator (src/network.py:172-186, src/module.py:297-336)
# ---------------------------------------------------------------------------------------------------------------
def lstm_discriminator(cx, tape, m, x, lens, Bd, T, need_input_grad=True):
    """x: Var [Bd*T, d_in] (zero padded), lens int32 [Bd].  Returns Var logits buffer [Bd, 4] (column 0 valid)."""
    st = cx.st
    Hh, L, ndir = m.hidden, m.num_layers, m.num_dir
    G4 = 4 * Hh
    dev = x.v.device
    p_inter = cx.p(m.dropout_p) if L > 1 else 0.0
    saved = []
    inp = x
    hfin = None
    for l in range(L):
        names = ["discriminator.rnn.rnn.%s_l%d" % (k, l) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        last = [n + ("_reverse" if ndir == 2 else "") for n in names]
        din = inp.v.shape[1]
        Wih = st.span(names[0], last[0], (ndir * G4, din))
        Whh = st.span(names[1], last[1], (ndir * G4, Hh))
        bih = st.span(names[2], last[2], (ndir * G4,))
        bhh = st.span(names[3], last[3], (ndir * G4,))
        xproj = _empty(Bd * T, ndir * G4, device=dev)
        ops.linear_fwd(inp.v, Wih, None, xproj)
        y = _empty(Bd, T, ndir * Hh, device=dev)                          # (the recurrence kernels write zeros at the padded steps themselves)
        gates = _empty(Bd, T, ndir, G4, device=dev)
        cs = _empty(Bd, T, ndir, Hh, device=dev)
        hprev = _empty(Bd, T, ndir, Hh, device=dev)
        hfin = _empty(Bd, ndir * Hh, device=dev)
        ops.lstm_fwd(xproj.view(Bd, T, ndir * G4), Whh, bih, bhh, lens, y, gates, cs, hprev, hfin, ndir, G4 * Hh, G4)
        yv = Var(y.view(Bd * T, ndir * Hh))
        rec = dict(l=l, names=names, last=last, Wih=Wih, Whh=Whh, inp=inp, y=yv, gates=gates, cs=cs, hprev=hprev, din=din, drop=None)
        saved.append(rec)
        if l + 1 < L:
            if p_inter > 0:
                s = cx.stream()
                yd = _empty(Bd * T, ndir * Hh, device=dev)
                ops.leaky_dropout(yv.v, None, yd, 1.0, drop_p=p_inter, seed=cx.seed, stream_id=s)
                rec["drop"] = s
                inp = Var(yd)
                rec["next_in"] = inp
            else:
                inp = yv
                rec["next_in"] = inp
    # head: reduce_h_W on [h_fwd | h_bwd] of the top layer -> LeakyReLU -> Dropout -> fc2
    p_head = cx.p(m.dropout_p)
    if ndir == 2:
        Wr, br = cx.P["discriminator.rnn.reduce_h_W.weight"], cx.P["discriminator.rnn.reduce_h_W.bias"]
        r = _empty(Bd, Hh, device=dev)
        ops.linear_fwd(hfin, Wr, br, r)
    else:
        r = hfin
    s_head = cx.stream()
    a = _empty(Bd, Hh, device=dev)
    ops.leaky_dropout(r, None, a, m.relu_slope, drop_p=p_head, seed=cx.seed, stream_id=s_head)
    W2, b2 = cx.P["discriminator.fc2.weight"], cx.P["discriminator.fc2.bias"]
    logit = torch.zeros(Bd, 4, dtype=torch.float32, device=dev)
    ops.linear_fwd(a, W2, b2, logit[:, :1])
    out = Var(logit)
    if tape is not None:
        seed = cx.seed

        def bwd():
            if out.g is None:
                return
            with ops.wgrad_batch():            # the input-projection gradients of both LSTM layers: one grouped launch
                bwd_body()

        def bwd_body():
            dl = out.g                                             # [Bd,4], column 0 valid
            g2 = st.g("discriminator.fc2.weight")
            if g2 is not None:
                ops.linear_wgrad(dl[:, :1], a, g2, db=st.g("discriminator.fc2.bias"))
            da = _empty(Bd, Hh, device=dev)
            ops.linear_dgrad(dl[:, :1], W2, da)
            dr = _empty(Bd, Hh, device=dev)
            ops.leaky_dropout(r, da, dr, m.relu_slope, drop_p=p_head, seed=seed, stream_id=s_head)
            if ndir == 2:
                gr = st.g("discriminator.rnn.reduce_h_W.weight")
                if gr is not None:
                    ops.linear_wgrad(dr, hfin, gr, db=st.g("discriminator.rnn.reduce_h_W.bias"))
                dhf = _empty(Bd, ndir * Hh, device=dev)
                ops.linear_dgrad(dr, Wr, dhf)
            else:
                dhf = dr
            dy = None
            for rec in reversed(saved):
                dg = _empty(Bd, T, ndir, G4, device=dev)
                ops.lstm_bwd(dy, dhf, rec["Whh"], rec["gates"], rec["cs"], lens, dg, ndir, G4 * Hh)
                dhf = None                                          # only the top layer's final state feeds the head
                dg2 = dg.view(Bd * T, ndir * G4)
                names, last = rec["names"], rec["last"]
                gWih = st.gspan(names[0], last[0], (ndir * G4, rec["din"]))
                if gWih is not None:
                    ops.linear_wgrad(dg2, rec["inp"].v, gWih, db=st.gspan(names[2], last[2], (ndir * G4,)))
                    gWhh = st.gspan(names[1], last[1], (ndir * G4, Hh))
                    hp = rec["hprev"].view(Bd * T, ndir * Hh)
                    for d in range(ndir):
                        ops.linear_wgrad(dg2[:, d * G4:(d + 1) * G4], hp[:, d * Hh:(d + 1) * Hh], gWhh[d * G4:(d + 1) * G4])
                    ops.colsum(dg2, st.gspan(names[3], last[3], (ndir * G4,)))
                need_dx = rec["l"] > 0 or need_input_grad
                if not need_dx:
                    break
                dxin = _empty(Bd * T, rec["din"], device=dev)
                ops.linear_dgrad(dg2, rec["Wih"], dxin)
                if rec["l"] > 0:
                    prev = saved[rec["l"] - 1]
                    if prev["drop"] is not None:
                        dyp = _empty(Bd * T, ndir * Hh, device=dev)
                        ops.leaky_dropout(prev["y"].v, dxin, dyp, 1.0, drop_p=p_inter, seed=seed, stream_id=prev["drop"])
                        dy = dyp.view(Bd, T, ndir * Hh)
                    else:
                        dy = dxin.view(Bd, T, ndir * Hh)
                else:
                    acc(x, dxin)
        tape.record(bwd)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Inference-only helpers used by unast_amd.inference (no tape)
# ---------------------------------------------------------------------------------------------------------------
def speech_prenet_step(cx, m, frames, pos_t):
    """SpeechPrenet on the frames [B, T, num_mels] at the device-resident position pos_t (one decoder input per sequence)."""
    a = m.args
    N = frames.shape[0]
    W1, b1 = cx.P["speech_m.prenet.layer.fc1.linear_layer.weight"], cx.P["speech_m.prenet.layer.fc1.linear_layer.bias"]
    W2, b2 = cx.P["speech_m.prenet.layer.fc2.linear_layer.weight"], cx.P["speech_m.prenet.layer.fc2.linear_layer.bias"]
    h1 = _empty(N, W1.shape[0], like=frames)
    ops.decode_linear(None, W1, b1, h1, act=1, drop_p=cx.p(a.s_pre_drop), seed=cx.seed, stream_id=cx.stream(), x_frames=frames, pos=pos_t)
    h2 = _empty(N, W2.shape[0], like=frames)
    ops.decode_linear(h1, W2, b2, h2, act=1)
    return h2


def text_postnet(cx, m, hid3d):
    """TextPostnet (src/module.py:233-246) on a [B,T,E] tensor: fc1(dropout(x)) -> [B,T,V]; no tape."""
    B, T, E = hid3d.shape
    N = B * T
    x = hid3d.reshape(N, E)
    p = cx.p(m.args.t_post_drop)
    if p > 0:
        xd = _empty(N, E, like=x)
        ops.leaky_dropout(x, None, xd, 1.0, drop_p=p, seed=cx.seed, stream_id=cx.stream())
        x = xd
    W, b = cx.P["text_m.postnet.fc1.weight"], cx.P["text_m.postnet.fc1.bias"]
    V = W.shape[0]
    ldl = (V + 3) // 4 * 4
    logits = torch.zeros(N, ldl, dtype=torch.float32, device=x.device)
    ops.linear_fwd(x, W, b, logits[:, :V])
    return logits.view(B, T, ldl)[..., :V]


def speech_postnet_residual(cx, m, mel3d, residual=True):
    """mel + SpeechPostnet(mel) for a [B,T,M] tensor (src/network.py:246; BN in the model's current mode); residual=False:
    SpeechPostnet(mel) alone (SpeechTransformer.postprocess, src/network.py:216-217).  No tape."""
    B, T, M = mel3d.shape
    a = m.args
    N = B * T
    x = Var(mel3d.reshape(N, M))
    y = conv_bn_act(cx, None, x, B, T, "speech_m.postnet.conv1.", "speech_m.postnet.pre_batchnorm.", 4, 2, a.s_post_drop, m.buffers_dict)
    for i in range(3):
        y = conv_bn_act(cx, None, y, B, T, "speech_m.postnet.conv_list.%d." % i, "speech_m.postnet.batch_norm_list.%d." % i, 4, 2,
                        a.s_post_drop, m.buffers_dict)
    Wp2, b2 = cx.P["speech_m.postnet.conv2.conv.weight"], cx.P["speech_m.postnet.conv2.conv.bias"]
    post = _empty(B, T, M, like=mel3d)
    C = y.v.shape[1]
    ops.gemm(ops.OP_KC_CONV, ops.OP_KC, y.v, C, Wp2, 5 * C, post, M, N, M, 5 * C, conv=(T, C, 0, 4), bias=b2, R=x.v if residual else None, ldr=M)
    return post
