"""UNAST task graph, transformer autoencoders and discriminators with the reference's API (src/network.py), executed
by hand-written HIP kernels.

Public surface kept:
  UNAST(text_m, speech_m, discriminator=None, teacher=None): text_ae, speech_ae, tts, asr, num_params
  TextTransformer / SpeechTransformer(args): encode, decode_sequence, forward, preprocess
  LSTMDiscriminator(d_in, hidden, out=1, bidirectional=False, num_layers=1, dropout=.2, relu=.2).forward(out, out_len)
  Discriminator(enc_dim, hidden=1024, out_classes=1, dropout=.2, relu=.2).forward(enc_output)
`state_dict()` keys/shapes equal the reference's (unast_amd.spec).  The `masks` tuple returned by encode() is opaque
to callers in the reference (only passed back into decode_sequence); here it carries int32 lengths instead of bool
mask tensors — the kernels test `t < len[b]` themselves (no B x T host loop, cf. src/utils.py:77-83).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import functional as F
from . import ops
from .engine import Var, acc, run_segment, FlatStore, on_stream, ddp_hook
from .module import (SpeechPrenet, SpeechPostnet, TextPrenet, TextPostnet, PositionalEncoding, TransformerEncoder,
                     TransformerDecoder, RNNEncoder)
from .spec import state_dict_spec  # noqa: F401  (re-export)
from .utils import PAD_IDX, SOS_IDX, EOS_IDX, lens_i32, next_seed  # noqa: F401

_HP_KEYS = ("num_mels", "s_pre_hid", "s_pre_drop", "s_post_drop", "t_emb_dim", "t_pre_drop", "t_post_drop", "hidden", "e_in",
            "e_drop", "d_drop", "num_layers", "nhead", "ffn_dim")


def _hp(args):
    hp = SimpleNamespace(**{k: getattr(args, k) for k in _HP_KEYS if hasattr(args, k)})
    if hp.hidden != hp.e_in:
        raise ValueError("hidden must equal e_in (the heads read the decoder output; src/network.py:196 vs src/module.py:152-153)")
    return hp


def _as_padded(t, rows, ld, cols):
    """Recover (or build) the [rows, ld] padded buffer behind a [..., cols] view; device-memory plumbing only."""
    if t is None:
        return None
    if t.dim() >= 2 and t.stride(-1) == 1 and t.stride(-2) == ld and t.storage_offset() % 4 == 0 and \
            t.untyped_storage().nbytes() // 4 >= t.storage_offset() + rows * ld:
        lead = t.shape[:-1].numel()
        if lead == rows and (t.dim() == 2 or t.stride(0) == t.shape[1] * ld):
            return t.as_strided((rows, ld), (ld, 1))
    buf = torch.zeros(rows, ld, dtype=torch.float32, device=t.device)
    buf[:, :cols].copy_(t.reshape(rows, cols))
    return buf


class _Side(nn.Module):
    """Shared plumbing of TextTransformer / SpeechTransformer / LSTMDiscriminator: canonical name prefix, parameter
    store lookup (the enclosing UNAST's store when wrapped), BN buffers by canonical name."""
    _prefix = ""

    def _root(self):
        r = self.__dict__.get("_unast_root")
        return r() if r is not None else None

    def _store(self):
        root = self._root()
        if root is not None:
            return root._store()
        st = self.__dict__.get("_unast_store")
        dev = next(self.parameters()).device
        if st is None or st.device != dev:
            st = FlatStore(self, prefix=self._prefix)
            self.__dict__["_unast_store"] = st
        return st

    @property
    def buffers_dict(self):
        d = self.__dict__.get("_bufs")
        dev = next(self.parameters()).device
        if d is None or d["__dev"] != dev:
            d = {self._prefix + n: b for n, b in self.named_buffers()}
            d["__dev"] = dev
            self.__dict__["_bufs"] = d
        return d

    def _ctx(self):
        return F.Ctx(self._store(), self.training, next_seed())


class AutoEncoderNet(_Side):
    """src/network.py:12-86 (abstract interface)."""

    def preprocess(self, input_, input_lens):
        raise Exception("Please use a subclass for text or speech")


def _lens_of_pad_mask(mask, T, device):
    """Lengths behind a key-padding mask.  The kernels mask by length, so a bool mask [B,T] (True = padded; the reference's form,
    src/network.py:411-413) must be a suffix mask, which is all the reference ever builds; int lengths (this package's masks[1])
    pass through; None = nothing padded."""
    if mask is None:
        raise ValueError("pass the key-padding mask (bool [B,T]) or the lengths (int [B])")
    if mask.dtype != torch.bool:
        return lens_i32(mask, device)
    if mask.shape[1] > 1 and bool((mask[:, 1:] < mask[:, :-1]).any()):
        raise ValueError("key-padding masks must pad a suffix of each sequence (lengths are what the attention kernels take)")
    return (T - mask.sum(dim=1)).to(device=device, dtype=torch.int32)


def _out3d(tape, var2d, B, T):
    """Expose a [B*T, C] Var as a [B, T, C] output Var (a view; its gradient flows back as a view)."""
    o = Var(var2d.v.view(B, T, var2d.v.shape[1]))
    if tape is not None:
        def bwd():
            if o.g is not None:
                acc(var2d, (o.g if o.g.is_contiguous() else o.g.contiguous()).view(B * T, -1))
        tape.record(bwd)
    return o


def _pair_outputs(tape, var2d, B, T):
    """The two row blocks of a [2 B T, C] Var (a paired encoder call's stack output) as [B, T, C] output Vars, each twice: the value and
    a second alias of it for its second consumer (see _alias: the decoder and the discriminator both read an encoder output, on two
    streams).  The four gradients arrive separately; the stack's backward wants one buffer: each block = value gradient + alias gradient,
    summed straight into its rows by one kernel per block (no memcpy node in a captured step, no separate accumulate pass)."""
    N, C = B * T, var2d.v.shape[1]
    outs = []
    for i in range(2):
        blk = var2d.v[i * N:(i + 1) * N].view(B, T, C)
        outs += [Var(blk), Var(blk.view(B, T, C))]
    if tape is not None:
        def bwd():
            if all(o.g is None for o in outs):
                return
            g = torch.empty_like(var2d.v)
            for i in range(2):
                parts = [o.g if o.g.is_contiguous() else o.g.contiguous() for o in outs[2 * i:2 * i + 2] if o.g is not None]
                dst = g[i * N:(i + 1) * N]
                if not parts:
                    dst.zero_()
                else:
                    ops.sum2(dst, parts[0], parts[1] if len(parts) > 1 else None)
            acc(var2d, g)
        tape.record(bwd)
    return outs


def _alias(tape, var):
    """A second output of the same activation (a view) for a second consumer.  Each of the two then has exactly one user in
    torch's autograd graph and THIS call's backward adds the two incoming gradients itself (on its own stream, after autograd
    has ordered it behind both producers).  Letting autograd sum gradients that arrive from different streams in a node's
    input buffer works eagerly but takes hipStreamEndCapture down when the step is captured into a HIP graph (ROCm 7.2;
    the round-2 capture probe, cases aedisc_* vs discdetach_ms, in the git history), and the encoder output has two users on two streams: the
    decoder (cross-attention memory) and the discriminator."""
    o = Var(var.v.view(var.v.shape))
    if tape is not None:
        def bwd():
            if o.g is not None:
                acc(var, o.g if o.g.is_contiguous() else o.g.contiguous())
        tape.record(bwd)                                   # recorded after var's own closure => runs before it
    return o


def _mem_in(tape, mem, B, Tk):
    """[B,Tk,E] input Var -> [B*Tk,E] Var; its closure (recorded first => runs last) hands the gradient back."""
    memv = Var(mem.v.contiguous().view(B * Tk, -1))
    if tape is not None:
        tape.record(lambda: setattr(mem, "g", None if memv.g is None else memv.g.view(B, Tk, -1)))
    return memv


def _speech_outputs(tape, head, post, B, T, M, device):
    """(pre, post, stop) output Vars of a speech decoder call over its head buffer [B T, ldh] and post-net output [B T, M]; their closure
    (recorded last => runs first) turns the three incoming gradients back into d(head) and d(post)."""
    ldh = head.v.shape[1]
    h3 = head.v.view(B, T, ldh)
    o_pre, o_post, o_stop = Var(h3[..., :M]), Var(post.v.view(B, T, M)), Var(h3[..., M])
    if tape is not None:
        def bwd():
            if o_post.g is not None:
                post.g = (o_post.g if o_post.g.is_contiguous() else o_post.g.contiguous()).view(B * T, M)
            gp, gs = o_pre.g, o_stop.g
            if gp is None and gs is None:
                return
            if gp is not None and gs is not None and gp.stride(-1) == 1 and gp.stride(-2) == ldh and \
                    gs.data_ptr() == gp.data_ptr() + 4 * M and gs.stride(-1) == ldh:
                head.g = gp.as_strided((B * T, ldh), (ldh, 1))          # both are views of one d_head buffer
            else:
                buf = torch.zeros(B * T, ldh, dtype=torch.float32, device=device)
                if gp is not None:
                    buf[:, :M].copy_(gp.reshape(B * T, M))
                if gs is not None:
                    buf[:, M].copy_(gs.reshape(B * T))
                head.g = buf
        tape.record(bwd)
    return [o_pre, o_post, o_stop]


class TextTransformer(AutoEncoderNet):
    """src/network.py:417-500."""
    _prefix = "text_m."

    def __init__(self, args):
        super().__init__()
        if args.t_emb_dim != args.e_in:
            raise ValueError("t_emb_dim must equal e_in (decoder input skips the conv prenet; src/network.py:435-438)")
        self.prenet = TextPrenet(args.t_emb_dim, args.e_in, p=args.t_pre_drop)
        self.pos_emb = PositionalEncoding(args.e_in)
        self.encoder = TransformerEncoder(args.e_in, args.nhead, args.ffn_dim, args.e_drop, args.num_layers)
        self.decoder = TransformerDecoder(args.e_in, args.nhead, args.ffn_dim, args.d_drop, args.num_layers)
        self.postnet = TextPostnet(args.hidden, args.t_post_drop)
        self.args = _hp(args)

    @property
    def pe(self):
        return self.pos_emb.pe[0]

    @on_stream("text")
    def encode(self, input_, input_lens, noise_in=False):
        B, T = input_.shape
        lens = lens_i32(input_lens, input_.device)
        cx = self._ctx()
        ids = input_.contiguous()

        def run(tape, dummy):
            o = _out3d(tape, F.text_encode(cx, tape, self, ids, lens, noise_in), B, T)
            return [o, _alias(tape, o)]
        enc, enc_hid = run_segment(run, ddp_hook("text_enc", cx.st), cx.st.dummy)
        return enc, (None, lens, enc_hid)

    @on_stream("text")
    def encode_pair(self, in_a, lens_a, noise_a, in_b, lens_b, noise_b):
        """encode(in_a, lens_a, noise_a) and encode(in_b, lens_b, noise_b) of two batches of ONE shape as a single call: the front ends
        per batch (BatchNorm statistics and running-stat updates in the order a, b, as two calls would make them), the encoder stack once
        over both (unast_amd.functional.text_encode_pair).  Returns the two (enc_outputs, masks) pairs of the separate calls."""
        if in_a.shape != in_b.shape:
            raise ValueError("encode_pair: the two batches must have one shape")
        B, T = in_a.shape
        la, lb = lens_i32(lens_a, in_a.device), lens_i32(lens_b, in_b.device)
        lens2 = torch.cat([la, lb])
        cx = self._ctx()
        ids_a, ids_b = in_a.contiguous(), in_b.contiguous()

        def run(tape, dummy):
            return _pair_outputs(tape, F.text_encode_pair(cx, tape, self, ids_a, noise_a, ids_b, noise_b, lens2), B, T)
        ea, ha, eb, hb = run_segment(run, ddp_hook("text_enc", cx.st), cx.st.dummy)
        return (ea, (None, la, ha)), (eb, (None, lb, hb))

    @on_stream("text")
    def decode_sequence(self, tgt, tgt_lens, enc_outputs, masks, teacher_ratio=1, loss_hint=None):
        """loss_hint (not in the reference's signature; unast_amd.train passes it): (gold, eos_weight, gscale, workspace) of the text_loss call
        that will follow on this call's logits -- the head GEMM then computes that loss and its gradient in the same launch."""
        B, T = tgt.shape
        Tk = enc_outputs.shape[1]
        lens_q = lens_i32(tgt_lens, tgt.device)
        lens_k = masks[1]
        cx = self._ctx()
        ids = tgt.contiguous()
        V = self.postnet.fc1.weight.shape[0]

        def run(tape, dummy, mem):
            memv = _mem_in(tape, mem, B, Tk)
            out = F.text_decode(cx, tape, self, ids, lens_q, memv, lens_k, Tk, loss_hint=loss_hint)
            ldl = out.v.shape[1]
            o = Var(out.v.view(B, T, ldl)[..., :V])
            if tape is not None:
                def bwd():
                    if o.g is not None:
                        out.g = _as_padded(o.g, B * T, ldl, V)
                tape.record(bwd)                                                    # recorded last => runs first
            return [o]
        return run_segment(run, ddp_hook("text_dec", cx.st), cx.st.dummy, enc_outputs)

    @on_stream("text")
    def decode_pair(self, tgt_a, lens_a, enc_a, masks_a, hint_a, tgt_b, lens_b, enc_b, masks_b, hint_b):
        """Two decode_sequence calls of one target shape as a single call (see SpeechTransformer.decode_pair): the decoder stack once over
        both, cross-attention per call on its own memory, head + loss per call.  Returns the two logits tensors."""
        if tgt_a.shape != tgt_b.shape:
            raise ValueError("decode_pair: the two targets must have one shape")
        B, T = tgt_a.shape
        Tks = (enc_a.shape[1], enc_b.shape[1])
        lq = torch.cat([lens_i32(lens_a, tgt_a.device), lens_i32(lens_b, tgt_b.device)])
        lens_ks = (masks_a[1], masks_b[1])
        cx = self._ctx()
        ids2 = (tgt_a.contiguous(), tgt_b.contiguous())
        V = self.postnet.fc1.weight.shape[0]

        def run(tape, dummy, mem_a, mem_b):
            mems = (_mem_in(tape, mem_a, B, Tks[0]), _mem_in(tape, mem_b, B, Tks[1]))
            res = []
            for out in F.text_decode_pair(cx, tape, self, ids2, lq, mems, lens_ks, Tks, (hint_a, hint_b)):
                ldl = out.v.shape[1]
                o = Var(out.v.view(B, T, ldl)[..., :V])
                if tape is not None:
                    def bwd(o=o, out=out, ldl=ldl):
                        if o.g is not None:
                            out.g = _as_padded(o.g, B * T, ldl, V)
                    tape.record(bwd)
                res.append(o)
            return res
        return run_segment(run, ddp_hook("text_dec", cx.st), cx.st.dummy, enc_a, enc_b)

    @on_stream("text")
    def decode(self, tgt, tgt_lens, tgt_pad_mask, enc_outputs, enc_mask):
        """src/network.py:446-450: one uncached generation step -- the decoder over all of `tgt` (token ids [B,T], fed as they are),
        logits [B,1,V] of its last position.  Forward only (no tape); generation proper runs on the K/V cache (infer_sequence)."""
        B, T = tgt.shape
        Tk = enc_outputs.shape[1]
        lens_q = _lens_of_pad_mask(tgt_pad_mask, T, tgt.device)
        lens_k = _lens_of_pad_mask(enc_mask, Tk, tgt.device)
        cx = self._ctx()
        with torch.no_grad():
            out = F.text_decode(cx, None, self, tgt.contiguous(), lens_q, Var(enc_outputs.detach().contiguous().view(B * Tk, -1)), lens_k, Tk, shift=False)
        V = self.postnet.fc1.weight.shape[0]
        return out.v.view(B, T, -1)[:, -1:, :V]

    @on_stream("text")
    def postprocess(self, out):
        """src/network.py:452-453: TextPostnet on decoder states [B,T,E] -> logits [B,T,V].  Forward only (the train step's post-net is
        part of decode_sequence's segment)."""
        with torch.no_grad():
            return F.text_postnet(self._ctx(), self, out.detach())

    def forward(self, text, text_len, noise_in=False, teacher_ratio=1, ret_enc_hid=False):
        enc_outputs, masks = self.encode(text, text_len, noise_in)
        dec_out = self.decode_sequence(text, text_len, enc_outputs, masks)
        if ret_enc_hid:
            return dec_out, masks[2]                       # the encoder output once more, for its second user (see _alias)
        return dec_out

    def infer_sequence(self, memory, masks, max_len=300):
        """src/network.py:455-481 with a K/V cache (unast_amd.inference).  Returns (tokens [B,T], stop_lens [B])."""
        return self.generation(memory, masks, max_len).run()

    infer_max_len = 300      # the cap the cross-model paths generate with (= infer_sequence's default)

    def generation(self, memory, masks, max_len=None):
        """infer_sequence as a resumable unast_amd.inference._Generation (lets UNAST.cm_both_in run two of them in lock-step)."""
        max_len = max_len or self.infer_max_len
        from .inference import text_generation
        return text_generation(self, self._ctx(), memory.detach(), masks[1], max_len)


class SpeechTransformer(AutoEncoderNet):
    """src/network.py:188-276."""
    _prefix = "speech_m."

    def __init__(self, args):
        super().__init__()
        self.prenet = SpeechPrenet(args.num_mels, args.s_pre_hid, args.e_in, p=args.s_pre_drop)
        self.pos_emb = PositionalEncoding(args.e_in)
        self.encoder = TransformerEncoder(args.e_in, args.nhead, args.ffn_dim, args.e_drop, args.num_layers)
        self.decoder = TransformerDecoder(args.e_in, args.nhead, args.ffn_dim, args.d_drop, args.num_layers)
        self.postnet = SpeechPostnet(args.num_mels, args.hidden, p=args.s_post_drop)
        self.args = _hp(args)

    @property
    def pe(self):
        return self.pos_emb.pe[0]

    @on_stream("speech")
    def encode(self, input_, input_lens, noise_in=False):
        B, T, M = input_.shape
        lens = lens_i32(input_lens, input_.device)
        cx = self._ctx()
        mel = input_.detach().contiguous()

        def run(tape, dummy):
            o = _out3d(tape, F.speech_encode(cx, tape, self, mel, lens, noise_in), B, T)
            return [o, _alias(tape, o)]
        enc, enc_hid = run_segment(run, ddp_hook("speech_enc", cx.st), cx.st.dummy)
        return enc, (None, lens, enc_hid)

    @on_stream("speech")
    def encode_pair(self, in_a, lens_a, noise_a, in_b, lens_b, noise_b):
        """Two encode calls of one shape as one (see TextTransformer.encode_pair): front ends per batch, the encoder stack once over both."""
        if in_a.shape != in_b.shape:
            raise ValueError("encode_pair: the two batches must have one shape")
        B, T, M = in_a.shape
        la, lb = lens_i32(lens_a, in_a.device), lens_i32(lens_b, in_b.device)
        lens2 = torch.cat([la, lb])
        cx = self._ctx()
        mel_a, mel_b = in_a.detach().contiguous(), in_b.detach().contiguous()

        def run(tape, dummy):
            return _pair_outputs(tape, F.speech_encode_pair(cx, tape, self, mel_a, noise_a, mel_b, noise_b, lens2), B, T)
        ea, ha, eb, hb = run_segment(run, ddp_hook("speech_enc", cx.st), cx.st.dummy)
        return (ea, (None, la, ha)), (eb, (None, lb, hb))

    @on_stream("speech")
    def decode_sequence(self, tgt, tgt_lens, enc_outputs, masks, teacher_ratio=1, loss_hint=None):
        """loss_hint (not in the reference's signature): (gold mel, lengths, eos_weight, gscale, workspace) of the speech_loss call that will
        follow on this call's outputs -- the head GEMM then computes the pre-net and stop terms and their gradient in the same launch."""
        B, T, M = tgt.shape
        Tk = enc_outputs.shape[1]
        lens_q = lens_i32(tgt_lens, tgt.device)
        lens_k = masks[1]
        cx = self._ctx()
        mel = tgt.detach().contiguous()

        def run(tape, dummy, mem):
            memv = _mem_in(tape, mem, B, Tk)
            head, post = F.speech_decode(cx, tape, self, mel, lens_q, memv, lens_k, Tk, loss_hint=loss_hint)
            return _speech_outputs(tape, head, post, B, T, M, mel.device)
        pre, post, stop = run_segment(run, ddp_hook("speech_dec", cx.st), cx.st.dummy, enc_outputs)
        return pre, post, stop, tgt_lens

    @on_stream("speech")
    def decode_pair(self, tgt_a, lens_a, enc_a, masks_a, hint_a, tgt_b, lens_b, enc_b, masks_b, hint_b):
        """decode_sequence(tgt_a, lens_a, enc_a, masks_a) and decode_sequence(tgt_b, lens_b, enc_b, masks_b) of two targets of ONE shape as
        a single call: the decoder stack runs once over both (self-attention and feed-forward over 2B sequences, cross-attention per call on
        its own memory), front ends, heads, post-net (BatchNorm statistics: first a, then b) and loss terms per call
        (unast_amd.functional.speech_decode_pair).  Returns the two (pre, post, stop, lens) tuples of the separate calls."""
        if tgt_a.shape != tgt_b.shape:
            raise ValueError("decode_pair: the two targets must have one shape")
        B, T, M = tgt_a.shape
        Tks = (enc_a.shape[1], enc_b.shape[1])
        lq = torch.cat([lens_i32(lens_a, tgt_a.device), lens_i32(lens_b, tgt_b.device)])
        lens_ks = (masks_a[1], masks_b[1])
        cx = self._ctx()
        mels = (tgt_a.detach().contiguous(), tgt_b.detach().contiguous())

        def run(tape, dummy, mem_a, mem_b):
            mems = (_mem_in(tape, mem_a, B, Tks[0]), _mem_in(tape, mem_b, B, Tks[1]))
            outs = F.speech_decode_pair(cx, tape, self, mels, lq, mems, lens_ks, Tks, (hint_a, hint_b))
            res = []
            for (head, post) in outs:
                res += _speech_outputs(tape, head, post, B, T, M, mels[0].device)
            return res
        pa, qa, sa, pb, qb, sb = run_segment(run, ddp_hook("speech_dec", cx.st), cx.st.dummy, enc_a, enc_b)
        return (pa, qa, sa, lens_a), (pb, qb, sb, lens_b)

    @on_stream("speech")
    def decode(self, tgt, tgt_lens, tgt_pad_mask, enc_outputs, enc_mask):
        """src/network.py:210-214: one uncached generation step -- the decoder over all of `tgt` ([B,T,M] frames, fed as they are),
        (mel [B,1,M], stop logit [B,1,1]) of its last position.  Forward only; generation proper runs on the K/V cache."""
        B, T, M = tgt.shape
        Tk = enc_outputs.shape[1]
        lens_q = _lens_of_pad_mask(tgt_pad_mask, T, tgt.device)
        lens_k = _lens_of_pad_mask(enc_mask, Tk, tgt.device)
        cx = self._ctx()
        with torch.no_grad():
            head, _ = F.speech_decode(cx, None, self, tgt.detach(), lens_q, Var(enc_outputs.detach().contiguous().view(B * Tk, -1)), lens_k, Tk,
                                      shift=False, postnet=False)
        h = head.v.view(B, T, -1)[:, -1:, :]
        return h[..., :M], h[..., M:M + 1]

    @on_stream("speech")
    def postprocess(self, out):
        """src/network.py:216-217: SpeechPostnet on [B,T,M] frames (the residual term; callers add it to `out`).  Forward only."""
        with torch.no_grad():
            return F.speech_postnet_residual(self._ctx(), self, out.detach().contiguous(), residual=False)

    def forward(self, mel, mel_len, noise_in=False, teacher_ratio=1, ret_enc_hid=False):
        enc_outputs, masks = self.encode(mel, mel_len, noise_in)
        pre_pred, post_pred, stop_pred, stop_lens = self.decode_sequence(mel, mel_len, enc_outputs, masks)
        if ret_enc_hid:
            return pre_pred, post_pred, stop_pred, masks[2]
        return pre_pred, post_pred, stop_pred

    def infer_sequence(self, memory, masks, max_len=815):
        """src/network.py:219-252 with a K/V cache.  Returns (pre [B,T,M], post [B,T,M], stop [B,T], stop_lens [B])."""
        return self.generation(memory, masks, max_len).run()

    infer_max_len = 815

    def generation(self, memory, masks, max_len=None):
        max_len = max_len or self.infer_max_len
        from .inference import speech_generation
        return speech_generation(self, self._ctx(), memory.detach(), masks[1], max_len,
                                 lambda cx, fr, pos: F.speech_prenet_step(cx, self, fr, pos), lambda cx, mel: F.speech_postnet_residual(cx, self, mel))


class LSTMDiscriminator(_Side):
    """src/network.py:172-186."""
    _prefix = "discriminator."

    def __init__(self, d_in, hidden, out=1, bidirectional=False, num_layers=1, dropout=.2, relu=.2):
        super().__init__()
        if hidden != 64 or out != 1:
            raise NotImplementedError("the persistent LSTM kernel is built for hidden=64, out=1 (disc_hid of every reference config)")
        self.num_dir = 2 if bidirectional else 1
        self.num_layers = num_layers
        self.hidden = hidden
        self.rnn = RNNEncoder(d_in, hidden, bidirectional=bidirectional, num_layers=num_layers, dropout=dropout)
        self.fc2 = nn.Linear(hidden, out)
        self.dropout_p, self.relu_slope = dropout, relu

    @on_stream("disc")
    def forward(self, out, out_len):
        Bd, T, Dm = out.shape
        lens = lens_i32(out_len, out.device)
        cx = self._ctx()
        need_dx = out.requires_grad and torch.is_grad_enabled()

        def run(tape, dummy, x):
            xv = Var(x.v.contiguous().view(Bd * T, Dm))
            if tape is not None:
                tape.record(lambda: setattr(x, "g", None if xv.g is None else xv.g.view(Bd, T, Dm)))
            lg = F.lstm_discriminator(cx, tape, self, xv, lens, Bd, T, need_input_grad=need_dx)
            o = Var(lg.v[:, 0])
            if tape is not None:
                tape.record(lambda: setattr(lg, "g", _as_padded(o.g.unsqueeze(-1), Bd, 4, 1)) if o.g is not None else None)
            return [o]
        return run_segment(run, None, cx.st.dummy, out)


class Discriminator(_Side):
    """src/network.py:154-170 (MLP discriminator of Lample et al.; present in the reference but never built by train.py)."""
    _prefix = "discriminator."

    def __init__(self, enc_dim, hidden=1024, out_classes=1, dropout=.2, relu=.2):
        super().__init__()
        self.fc1 = nn.Linear(enc_dim, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        self.fc3 = nn.Linear(hidden, hidden)
        self.fc4 = nn.Linear(hidden, out_classes)
        self.dropout_p, self.relu_slope, self.out_classes = dropout, relu, out_classes

    @on_stream("disc")
    def forward(self, enc_output):
        shape = enc_output.shape
        N, Dm = shape[:-1].numel(), shape[-1]
        cx = self._ctx()
        st = cx.st
        oc = self.out_classes
        ldo = (oc + 3) // 4 * 4

        def run(tape, dummy, x):
            xv = x.v.contiguous().view(N, Dm)
            acts, pres = [xv], []
            p = cx.p(self.dropout_p)
            streams = []
            h = xv
            for i in (1, 2, 3):
                W, b = cx.P["discriminator.fc%d.weight" % i], cx.P["discriminator.fc%d.bias" % i]
                u = torch.empty(N, W.shape[0], dtype=torch.float32, device=xv.device)
                ops.linear_fwd(h, W, b, u)
                a = torch.empty_like(u)
                s = cx.stream()
                ops.leaky_dropout(u, None, a, self.relu_slope, drop_p=p, seed=cx.seed, stream_id=s)
                pres.append(u); acts.append(a); streams.append(s)
                h = a
            W4, b4 = cx.P["discriminator.fc4.weight"], cx.P["discriminator.fc4.bias"]
            lg = torch.zeros(N, ldo, dtype=torch.float32, device=xv.device)
            ops.linear_fwd(h, W4, b4, lg[:, :oc])
            o = Var(lg[:, :oc].view(*shape[:-1], oc).squeeze(-1) if oc == 1 else lg[:, :oc].view(*shape[:-1], oc))
            if tape is not None:
                seed = cx.seed

                def bwd():
                    if o.g is None:
                        return
                    d = _as_padded(o.g.reshape(N, oc), N, ldo, oc)
                    g4 = st.g("discriminator.fc4.weight")
                    if g4 is not None:
                        ops.linear_wgrad(d[:, :oc], acts[3], g4)
                        ops.colsum(d[:, :oc], st.g("discriminator.fc4.bias"))
                    da = torch.empty_like(acts[3])
                    ops.linear_dgrad(d[:, :oc], W4, da)
                    for i in (3, 2, 1):
                        du = torch.empty_like(da)
                        ops.leaky_dropout(pres[i - 1], da, du, self.relu_slope, drop_p=p, seed=seed, stream_id=streams[i - 1])
                        gW = st.g("discriminator.fc%d.weight" % i)
                        if gW is not None:
                            ops.linear_wgrad(du, acts[i - 1], gW)
                            ops.colsum(du, st.g("discriminator.fc%d.bias" % i))
                        if i == 1 and not x.v.requires_grad:
                            break
                        da = torch.empty_like(acts[i - 1])
                        ops.linear_dgrad(du, cx.P["discriminator.fc%d.weight" % i], da)
                    else:
                        x.g = da.view(shape)
                tape.record(bwd)
            return [o]
        return run_segment(run, None, cx.st.dummy, enc_output)


class UNAST(_Side):
    """src/network.py:88-152."""

    def __init__(self, text_m, speech_m, discriminator=None, teacher=None):
        """NOTE: text_m and speech_m should be same type"""
        super().__init__()
        self.text_m = text_m
        self.speech_m = speech_m
        self.discriminator = discriminator
        self.teacher = teacher
        import weakref
        for sub in (text_m, speech_m, discriminator):
            if sub is not None:
                sub.__dict__["_unast_root"] = weakref.ref(self)

    def _root(self):
        return None

    def _store(self):
        st = self.__dict__.get("_unast_store")
        dev = next(self.parameters()).device
        if st is None or st.device != dev:
            st = FlatStore(self, prefix="")
            self.__dict__["_unast_store"] = st
        return st

    def text_ae(self, text, text_len, ret_enc_hid=False):
        return self.text_m.forward(text, text_len, noise_in=True, teacher_ratio=1, ret_enc_hid=ret_enc_hid)

    def speech_ae(self, mel, mel_len, ret_enc_hid=False):
        return self.speech_m.forward(mel, mel_len, noise_in=True, ret_enc_hid=ret_enc_hid, teacher_ratio=1)

    def tts(self, text, text_len, mel, mel_len, infer=False, ret_enc_hid=False):
        t_e_o, t_masks = self.text_m.encode(text, text_len)
        if not infer:
            pre_pred, post_pred, stop_pred, stop_lens = self.speech_m.decode_sequence(mel, mel_len, t_e_o, t_masks, teacher_ratio=1)
        else:
            pre_pred, post_pred, stop_pred, stop_lens = self.speech_m.infer_sequence(t_e_o, t_masks)
        if ret_enc_hid:
            return pre_pred, post_pred, stop_pred, stop_lens, t_masks[2]
        return pre_pred, post_pred, stop_pred, stop_lens

    def asr(self, text, text_len, mel, mel_len, infer=False, ret_enc_hid=False):
        s_e_o, s_masks = self.speech_m.encode(mel, mel_len)
        if not infer:
            text_pred = self.text_m.decode_sequence(text, text_len, s_e_o, s_masks, teacher_ratio=1)
        else:
            text_pred = self.text_m.infer_sequence(s_e_o, s_masks)
        if ret_enc_hid:
            return text_pred, s_masks[2]
        return text_pred

    def tts_and_asr(self, text, text_len, mel, mel_len, mel_aug, ret_enc_hid=False):
        """`tts(text, text_len, mel, mel_len)` and `asr(text, text_len, mel_aug, mel_len)` of the supervised step
        (src/train.py:231-259) issued encoder-first: with the text side and the speech side on their own HIP streams the
        reference's call order (tts, then asr) makes the speech decoder wait for the text encoder while the speech encoder
        queues behind it, and leaves the text decoder alone at the end.  Same four calls, same arithmetic; only the issue
        order differs (text encoder | speech encoder, then speech decoder | text decoder)."""
        t_e_o, t_masks = self.text_m.encode(text, text_len)
        s_e_o, s_masks = self.speech_m.encode(mel_aug, mel_len)
        pre_pred, post_pred, stop_pred, stop_lens = self.speech_m.decode_sequence(mel, mel_len, t_e_o, t_masks, teacher_ratio=1)
        text_pred = self.text_m.decode_sequence(text, text_len, s_e_o, s_masks, teacher_ratio=1)
        if ret_enc_hid:
            return (pre_pred, post_pred, stop_pred, stop_lens, t_masks[2]), (text_pred, s_masks[2])
        return (pre_pred, post_pred, stop_pred, stop_lens), text_pred

    def cm_text_in(self, text, text_len, ret_enc_hid=False):
        """src/network.py:103-112: text -> (no grad) TTS inference -> speech encoder -> text decoder."""
        with torch.no_grad():
            t_e_o, t_mask = self.text_m.encode(text, text_len)
            _, post_pred, _, pred_lens = self.speech_m.infer_sequence(t_e_o, t_mask, self.speech_m.infer_max_len)
        cm_s_e_o, cm_mask = self.speech_m.encode(post_pred.detach(), pred_lens.detach())
        text_pred = self.text_m.decode_sequence(text, text_len, cm_s_e_o, cm_mask, teacher_ratio=1)
        if ret_enc_hid:
            return text_pred, cm_mask[2], pred_lens
        return text_pred

    def cm_speech_in(self, mel, mel_len, ret_enc_hid=False):
        """src/network.py:114-123: speech -> (no grad) ASR inference -> text encoder -> speech decoder."""
        with torch.no_grad():
            s_e_o, s_mask = self.speech_m.encode(mel, mel_len)
            text_pred, text_pred_len = self.text_m.infer_sequence(s_e_o, s_mask, self.text_m.infer_max_len)
        cm_t_e_o, cm_t_masks = self.text_m.encode(text_pred.detach(), text_pred_len.detach())
        pre_pred, post_pred, stop_pred, stop_lens = self.speech_m.decode_sequence(mel, mel_len, cm_t_e_o, cm_t_masks, teacher_ratio=1)
        if ret_enc_hid:
            return pre_pred, post_pred, stop_pred, cm_t_masks[2], text_pred_len
        return pre_pred, post_pred, stop_pred

    def cm_both_in(self, text, text_len, mel, mel_len, ret_enc_hid=False):
        """cm_speech_in and cm_text_in of one batch (src/network.py:103-123, called back to back by src/train.py:261-294) with
        their two generations -- which do not depend on each other -- decoded in lock-step (unast_amd.inference.run_pair).
        Returns (cm_speech_in's result, cm_text_in's result)."""
        from .inference import run_pair
        with torch.no_grad():
            s_e_o, s_mask = self.speech_m.encode(mel, mel_len)
            t_e_o, t_mask = self.text_m.encode(text, text_len)
            (text_pred, text_pred_len), (_, post_pred, _, pred_lens) = run_pair(self.text_m.generation(s_e_o, s_mask),
                                                                                self.speech_m.generation(t_e_o, t_mask))
        cm_t_e_o, cm_t_masks = self.text_m.encode(text_pred.detach(), text_pred_len.detach())
        pre_pred, post_pred_s, stop_pred, stop_lens = self.speech_m.decode_sequence(mel, mel_len, cm_t_e_o, cm_t_masks, teacher_ratio=1)
        cm_s_e_o, cm_mask = self.speech_m.encode(post_pred.detach(), pred_lens.detach())
        text_out = self.text_m.decode_sequence(text, text_len, cm_s_e_o, cm_mask, teacher_ratio=1)
        if ret_enc_hid:
            return (pre_pred, post_pred_s, stop_pred, cm_t_masks[2], text_pred_len), (text_out, cm_mask[2], pred_lens)
        return (pre_pred, post_pred_s, stop_pred), text_out

    def num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def expose_grads(self):
        """Materialise p.grad (views into the flat gradient buffer) for inspection or third-party optimizers."""
        self._store().expose_grads()
