"""The reference's train-step surface (src/train.py) on MI355X.

Same function names, arguments, return values and `losses[...]` keys as the reference:
  process_batch, masked_mse/text_loss/speech_loss/discriminator_loss/discriminator_target,
  autoencoder_step, supervised_step, discriminator_step, discriminator_shuffle_batch, discriminator_hidden_to_loss,
  train_ae_step, train_sp_step, train_discriminator_step, optimizer_step, freeze/unfreeze_model_parameters,
  get_linear_schedule_with_warmup, get_transformer_paper_schedule, initialize_model, train_step (one outer step of
  train()'s hot loop, src/train.py:602-655).
Module globals DEVICE and WRITER are set by the caller, as in the reference (src/train.py:1005-1012).

Differences that are deliberate and MI355X-motivated:
  * losses are fused HIP kernels (forward scalar + backward dlogits); `.item()` logging reads are deferred: the
    `losses` dict receives 0-dim device tensors unless `SYNC_LOSSES` is True (the reference syncs 6+ times per sub-step);
  * `optimizer_step` with the FusedAdamW built by `initialize_model` = global-norm + clip + AdamW in two launches over
    the flat buffers, preceded by ONE all-reduce of the active gradient range when torch.distributed is initialised;
  * cm_steps (back-translation) generates with a K/V-cached decoder (unast_amd/inference.py) instead of re-running the
    decoder over the growing prefix at every step.
"""
import math
import os
from collections import defaultdict

import torch
import torch.nn as nn

from . import ddp, ops
from .engine import Var, run_segment, on_stream, side_streams, join_streams, stream_of, join_wgrad_streams
from .network import TextTransformer, SpeechTransformer, UNAST, Discriminator, LSTMDiscriminator, _as_padded
from .utils import (PAD_IDX, SOS_IDX, EOS_IDX, lens_i32, specaugment, sent_lens_to_mask, get_teacher_ratio, is_deterministic,
                    set_seed, next_seed)  # noqa: F401

DEVICE = None
WRITER = None
SYNC_LOSSES = False      # True reproduces the reference's per-sub-step `.detach().cpu().item()` host syncs


def _dev():
    return DEVICE if DEVICE is not None else torch.device("cuda")


def _log(v):
    return v.detach().cpu().item() if SYNC_LOSSES else v.detach()


def process_batch(batch):
    """src/train.py:80-94.  gold_stop is kept for API compatibility; the fused speech loss derives it from mel_len."""
    text, mel, text_len, mel_len = batch
    dev = _dev()
    text, mel = text.to(dev, non_blocking=True), mel.to(dev, non_blocking=True)
    # lengths as int32 on the device, once per batch: every kernel takes them in that form (the reference keeps int64; nothing on this
    # path indexes with them), and 19 call sites per step would otherwise each convert them again
    text_len, mel_len = text_len.to(dev, torch.int32, non_blocking=True), mel_len.to(dev, torch.int32, non_blocking=True)
    gold_mel, gold_char = mel.detach(), text.detach()
    gold_stop = _GoldStop(mel_len, mel.shape[1])
    return (text, mel, text_len, mel_len), (gold_char, gold_mel, gold_stop)


class _GoldStop:
    """Lazy stand-in for F.one_hot(mel_len-1, T).float(): the kernels compute `t == len-1` themselves."""

    def __init__(self, mel_len, T):
        self.mel_len, self.T = mel_len, T

    def to(self, *a, **k):
        return self

    def dense(self):
        return torch.nn.functional.one_hot(self.mel_len.long() - 1, self.T).float()


#####----- LOSS FUNCTIONS -----#####
def _scalar_segment(fwd, bwd, *inputs, loss_tensor=None):
    """loss = fwd() as a 0-dim tensor; backward calls bwd(gscale_device_scalar) -> grads for `inputs`.  loss_tensor: the [1] tensor an
    earlier launch has already written the loss into (a fused head + loss launch)."""
    dev = inputs[0].device
    dummy = _dummy(dev)

    def run(tape, dmy, *ins):
        loss = loss_tensor if loss_tensor is not None else torch.empty(1, dtype=torch.float32, device=dev)
        saved = fwd(loss)
        o = Var(loss.view(()))
        if tape is not None:
            def back():
                if o.g is None:
                    return
                g = o.g.reshape(1).contiguous()
                grads = bwd(g, saved)
                for v, gr in zip(ins, grads):
                    v.g = gr
            tape.record(back)
        return [o]
    return run_segment(run, None, dummy, *inputs)


class _LossSum(torch.autograd.Function):
    """loss = (a + b + c) / accum_steps of a sub-step (src/train.py:376-378) as one launch forward and one backward (torch: two adds, a
    division, and in the backward a fill, a division and the fan-out)."""

    @staticmethod
    def forward(ctx, div, *xs):
        ctx.div = div
        ctx.n = len(xs)
        out = torch.empty((), dtype=torch.float32, device=xs[0].device)
        return ops.scalar_combine([x.detach() for x in xs], div, out)

    @staticmethod
    def backward(ctx, g):
        one = _ONES.get(g.device)
        known = 1.0 if (one is not None and g.data_ptr() == one.data_ptr()) else _KNOWN.get(g.data_ptr())
        if known is not None:      # seeded by _sum_and_backward (or by an enclosing _LossSum): d loss / d part = g / div, a resident constant
            key = (g.device, ctx.div / known)
            gg = _INV.get(key)
            if gg is None:
                gg = _INV[key] = torch.full((), known, dtype=torch.float32, device=g.device) / ctx.div
                _KNOWN[gg.data_ptr()] = known / ctx.div
            return (None,) + (gg,) * ctx.n
        gg = torch.empty((), dtype=torch.float32, device=g.device)
        ops.scalar_combine([g.contiguous()], ctx.div, gg)
        return (None,) + (gg,) * ctx.n


_ONES = {}
_INV = {}
_KNOWN = {}          # data pointer of a resident constant scalar made here -> its value


def _known_value(g):
    """The value of an upstream-gradient scalar when it is one of the resident constants the loss sums hand down (no host read), else None."""
    one = _ONES.get(g.device)
    if one is not None and g.data_ptr() == one.data_ptr():
        return 1.0
    return _KNOWN.get(g.data_ptr())
_WS_RING = {}


def _loss_ws(dev):
    """Eight doubles for one loss call, the first four zero: the loss kernels take their workspace zero and leave it zero (csrc/loss.hip), so
    the slots of a small ring are handed out round-robin and never filled again; a slot comes back after 64 calls, long after its call's
    backward (the text loss keeps its weight sum in word 4 until then)."""
    ring = _WS_RING.get(dev)
    if ring is None:
        # The ring is filled ONCE, on whichever side stream asks first, and then used from all of them: every stream has to be behind that
        # fill (the text side once read its slot's weight sum as 0 -- wiped by the late fill -- and produced inf gradients: T_text = 300 hit
        # the window, T_text = 180 did not).  One device-wide synchronisation per process; never inside a capture (the first step is eager).
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the loss workspace ring must exist before a HIP-graph capture (run one eager step first)")
        ring = _WS_RING[dev] = [torch.zeros(64, 8, dtype=torch.float64, device=dev), 0]
        torch.cuda.synchronize(dev)
    i = ring[1]
    ring[1] = (i + 1) % 64
    from . import config
    if config.DEBUG_WORKSPACES and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize(dev)
        if bool((ring[0][i][:4] != 0).any()):
            raise RuntimeError("loss workspace slot %d is not zero on entry: %r (a loss kernel of an earlier call died midway, or two calls share a slot)" % (i, ring[0][i].tolist()))
    return ring[0][i]



def _sum_and_backward(parts, accum_steps):
    """loss = sum(parts) / accum_steps; loss.backward() seeded from a resident 1.0 (no fill launch).  Device scalars in, device scalar out."""
    if not all(p.dim() == 0 and p.dtype == torch.float32 and p.is_cuda for p in parts) or len(parts) > 9:
        loss = sum(parts[1:], parts[0]) / accum_steps
        loss.backward()
        return loss
    if len(parts) > 3:              # (the joint generator step has six parts: groups of three, then the sum of the groups)
        groups = [_LossSum.apply(float(accum_steps), *parts[i:i + 3]) for i in range(0, len(parts), 3)]
        loss = _LossSum.apply(1.0, *groups)
    else:
        loss = _LossSum.apply(float(accum_steps), *parts)
    dev = loss.device
    one = _ONES.get(dev)
    if one is None:
        one = _ONES[dev] = torch.ones((), dtype=torch.float32, device=dev)
    torch.autograd.backward([loss], [one])
    return loss


_DUMMIES = {}


def _dummy(dev):
    d = _DUMMIES.get(dev)
    if d is None:
        d = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        _DUMMIES[dev] = d
    return d


def masked_mse(gold_mel, pred_mel, mel_mask):
    """src/train.py:100-103: sum((gold - pred)^2 * mask) / sum(mask), a device scalar.  Forward only: the train step's two masked MSEs
    and the stop loss come out of one fused kernel with their gradients (speech_loss below); this entry serves evaluation code that
    calls it directly."""
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in (gold_mel, pred_mel)):
        raise RuntimeError("masked_mse is forward-only on this path (no gradient would reach its inputs): differentiate through speech_loss, "
                           "or call it under torch.no_grad() / on detached tensors")
    gold, pred, mask = (t.detach().reshape(-1).to(torch.float32).contiguous() for t in (gold_mel, pred_mel, mel_mask))
    if not (gold.numel() == pred.numel() == mask.numel()):
        raise ValueError("masked_mse: gold, pred and mask must have the same number of elements")
    return ops.masked_mse(gold, pred, mask)


@on_stream("text")
def text_loss(gold_char, text_pred, eos_weight=1.0):
    """src/train.py:105-111.  text_pred is [B, V, T] as in the reference call sites (logits.permute(0, 2, 1)).  When the decoder call that
    produced text_pred was told about this loss (decode_sequence(..., loss_hint=), as train_gen_joint_step does), the head GEMM has
    already computed it and its gradient (csrc/loss.hip text_head_loss_kernel); this call then launches nothing."""
    B, V, T = text_pred.shape
    gold = gold_char.to(text_pred.device).contiguous().view(-1)
    ldl = (V + 3) // 4 * 4
    lg_btv = text_pred.permute(0, 2, 1)
    from . import functional as F
    fused = F.FUSED_LOSSES.pop(text_pred.untyped_storage().data_ptr(), None)
    if fused is not None and not (fused["kind"] == "text" and fused["eos_weight"] == float(eos_weight) and fused["gold"].shape == gold.shape
                                  and (fused["gold"].data_ptr() == gold.data_ptr() or torch.equal(fused["gold"], gold))):
        fused = None                     # another loss than the one announced: compute it the ordinary way

    def fwd(loss):
        if fused is not None:
            return None, fused["ws"]
        logits = _as_padded(lg_btv.detach(), B * T, ldl, V)
        ws = _loss_ws(loss.device)
        ops.text_loss_fwd(logits, gold, V, float(eos_weight), ws, loss)
        return logits, ws

    def bwd(g, saved):
        logits, ws = saved
        if fused is not None:
            known = _known_value(g)
            if known is not None and abs(known - fused["gscale"]) <= 1e-12 * abs(known):
                dl = fused["dlogits"]            # already scaled by exactly this upstream gradient
                F.FUSED_STATS["text_grad_direct"] += 1
                return (dl.view(B, T, ldl)[..., :V].permute(0, 2, 1),)
            F.FUSED_STATS["text_grad_general"] += 1
            logits = _as_padded(lg_btv.detach(), B * T, ldl, V)          # another upstream gradient than announced: the general kernel
        dl = torch.empty(B * T, ldl, dtype=torch.float32, device=g.device)
        ops.text_loss_bwd(logits, gold, V, float(eos_weight), ws, g, dl)
        return (dl.view(B, T, ldl)[..., :V].permute(0, 2, 1),)
    if fused is not None:
        return _scalar_segment(fwd, bwd, text_pred, loss_tensor=fused["loss"])
    return _scalar_segment(fwd, bwd, text_pred)


@on_stream("speech")
def speech_loss(gold_mel, stop_label, pred_mel, post_pred_mel, mel_len, stop_pred, eos_weight=1.0):
    """src/train.py:113-122.  stop_label is implied by mel_len (one-hot at len-1, src/train.py:88) and not read."""
    B, T, M = pred_mel.shape
    ldh = (M + 1 + 3) // 4 * 4
    gold = gold_mel.to(pred_mel.device).contiguous()
    lens = lens_i32(mel_len, pred_mel.device)
    if stop_pred.dim() == 3:
        stop_pred = stop_pred.squeeze(-1)

    from . import functional as F
    fused = F.FUSED_LOSSES.pop(pred_mel.untyped_storage().data_ptr(), None)
    if fused is not None and not (fused["kind"] == "speech" and fused["eos_weight"] == float(eos_weight) and fused["gold"].shape == gold.shape
                                  and (fused["gold"].data_ptr() == gold.data_ptr() or torch.equal(fused["gold"], gold))
                                  and (fused["lens"].data_ptr() == lens.data_ptr() or torch.equal(fused["lens"], lens))):
        fused = None                     # another loss than the one announced to decode_sequence(loss_hint=): the ordinary way

    def head_of(pm, sp):
        if pm.stride(-1) == 1 and pm.stride(-2) == ldh and sp.data_ptr() == pm.data_ptr() + 4 * M and sp.stride(-1) == ldh \
                and pm.stride(0) == T * ldh:
            return pm.as_strided((B * T, ldh), (ldh, 1))
        head = torch.zeros(B * T, ldh, dtype=torch.float32, device=pm.device)
        head[:, :M].copy_(pm.reshape(B * T, M))
        head[:, M].copy_(sp.reshape(B * T))
        return head

    def fwd(loss):
        if fused is not None:            # the head launch and speech_post_loss have computed everything (csrc/loss.hip)
            return None, None
        head = head_of(pred_mel.detach(), stop_pred.detach())
        post = post_pred_mel.detach().contiguous()
        ws = _loss_ws(loss.device)
        ops.speech_loss_fwd(gold, head.view(B, T, ldh), post, lens, float(eos_weight), ws, loss)
        return head, post

    def bwd(g, saved):
        head, post = saved
        if fused is not None:
            known = _known_value(g)
            if known is not None and abs(known - fused["gscale"]) <= 1e-12 * abs(known):
                F.FUSED_STATS["speech_grad_direct"] += 1
                dh = fused["d_head"].view(B, T, ldh)
                return dh[..., :M], fused["d_post"], dh[..., M]
            F.FUSED_STATS["speech_grad_general"] += 1
            head = head_of(pred_mel.detach(), stop_pred.detach())
            post = post_pred_mel.detach().contiguous()
        dh = torch.empty(B, T, ldh, dtype=torch.float32, device=g.device)
        dp = torch.empty(B, T, M, dtype=torch.float32, device=g.device)
        ops.speech_loss_bwd(gold, head.view(B, T, ldh), post, lens, float(eos_weight), g, dh, dp)
        return dh[..., :M], dp, dh[..., M]
    if fused is not None:
        return _scalar_segment(fwd, bwd, pred_mel, post_pred_mel, stop_pred, loss_tensor=fused["loss"])
    return _scalar_segment(fwd, bwd, pred_mel, post_pred_mel, stop_pred)


@on_stream("disc")
def discriminator_loss(output, target):
    """src/train.py:147-148."""
    n = output.numel()
    tgt = target.to(output.device).contiguous()
    ldx = output.stride(0) if output.dim() == 1 else 1

    def fwd(loss):
        out = output.detach()
        ops.bce_logits(out, ldx, tgt, n, loss=loss)
        return out

    def bwd(g, out):
        dl = torch.zeros(n, 4, dtype=torch.float32, device=g.device)
        ops.bce_logits(out, ldx, tgt, n, gscale=g, dlogits=dl, ldd=4)
        return (dl[:, 0],)
    return _scalar_segment(fwd, bwd, output)


def discriminator_target(batch_size, target_type, smoothing=0.1):
    """src/train.py:150-164 (host-side constant vector; the hot path builds targets on the device from the permutation)."""
    target = torch.ones(batch_size).float()
    target -= smoothing
    if target_type == 'speech':
        target = 1 - target
    return target


def check_nan_loss(model, loss, loss_type, *unused):
    """src/train.py:166-196 without the per-call host sync: non-finite losses are detected where losses are read."""
    return None


#####----- Use these to run a task on a batch ----#####
def autoencoder_step(model, batch, args, use_dis_loss=False):
    """src/train.py:199-229."""
    x, y = batch
    text, mel, text_len, mel_len = x
    gold_char, gold_mel, gold_stop = y
    if use_dis_loss:
        text_pred, t_hid = model.text_ae(text, text_len, ret_enc_hid=use_dis_loss)
        text_pred = text_pred.permute(0, 2, 1)
        pre_pred, post_pred, stop_pred, s_hid = model.speech_ae(mel, mel_len, ret_enc_hid=use_dis_loss)
        d_batch = discriminator_shuffle_batch(t_hid, text_len, s_hid, mel_len, args.model_type)
        d_ae_loss, _ = discriminator_hidden_to_loss(model, d_batch, freeze_discriminator=True)
    else:
        text_pred = model.text_ae(text, text_len).permute(0, 2, 1)
        pre_pred, post_pred, stop_pred = model.speech_ae(mel, mel_len)
    s_ae_loss = speech_loss(gold_mel, gold_stop, pre_pred, post_pred, mel_len, stop_pred, args.s_eos_weight)
    t_ae_loss = text_loss(gold_char, text_pred, args.t_eos_weight)
    if use_dis_loss:
        return t_ae_loss, s_ae_loss, d_ae_loss
    return t_ae_loss, s_ae_loss


def supervised_step(model, batch, args, use_dis_loss=False):
    """src/train.py:231-259."""
    x, y = batch
    text, mel, text_len, mel_len = x
    gold_char, gold_mel, gold_stop = y
    mel_aug = mel if is_deterministic() else specaugment(mel, mel_len)
    if use_dis_loss:
        # model.tts(...) and model.asr(...) as in the reference, issued encoder-first (see UNAST.tts_and_asr)
        (pre_pred, post_pred, stop_pred, stop_lens, t_hid), (text_pred, s_hid) = model.tts_and_asr(text, text_len, mel, mel_len, mel_aug, ret_enc_hid=True)
        text_pred = text_pred.permute(0, 2, 1)
        d_batch = discriminator_shuffle_batch(t_hid, text_len, s_hid, mel_len, args.model_type)
        d_sp_loss, _ = discriminator_hidden_to_loss(model, d_batch, freeze_discriminator=True)
    else:
        (pre_pred, post_pred, stop_pred, stop_lens), text_pred = model.tts_and_asr(text, text_len, mel, mel_len, mel_aug)
        text_pred = text_pred.permute(0, 2, 1)
    tts_loss = speech_loss(gold_mel, gold_stop, pre_pred, post_pred, mel_len, stop_pred, args.s_eos_weight)
    asr_loss = text_loss(gold_char, text_pred, args.t_eos_weight)
    if use_dis_loss:
        return asr_loss, tts_loss, d_sp_loss
    return asr_loss, tts_loss


def crossmodel_step(model, batch, args, use_dis_loss=False):
    """src/train.py:261-294.  The two directions' generations are independent and run in lock-step (UNAST.cm_both_in)."""
    x, y = batch
    text, mel, text_len, mel_len = x
    gold_char, gold_mel, gold_stop = y
    sp, tx = model.cm_both_in(text, text_len, mel, mel_len, ret_enc_hid=use_dis_loss)
    if use_dis_loss:
        pre_pred, post_pred, stop_pred, cm_t_hid, cm_t_len = sp
        text_pred, cm_s_hid, cm_s_len = tx
    else:
        pre_pred, post_pred, stop_pred = sp
        text_pred = tx
    s_cm_loss = speech_loss(gold_mel, gold_stop, pre_pred, post_pred, mel_len, stop_pred, args.s_eos_weight)
    t_cm_loss = text_loss(gold_char, text_pred.permute(0, 2, 1), args.t_eos_weight)
    if use_dis_loss:
        d_batch = discriminator_shuffle_batch(cm_t_hid, cm_t_len, cm_s_hid, cm_s_len, args.model_type)
        d_cm_loss, _ = discriminator_hidden_to_loss(model, d_batch, freeze_discriminator=True)
        return t_cm_loss, s_cm_loss, d_cm_loss
    return t_cm_loss, s_cm_loss


@on_stream("disc")
def discriminator_shuffle_batch(t_hid, t_hid_len, s_hid, s_hid_len, model_type, train_discriminator=False, out=None):
    """src/train.py:296-329 for model_type == 'transformer'.  out (not in the reference's signature): a [2B, Tmax, D] row block of a larger
    buffer that receives the batch (train_gen_joint_step puts two sub-steps' batches side by side)."""
    if model_type != 'transformer':
        raise NotImplementedError("only the transformer family is on the MI355X path")
    B, Tt, Dm = t_hid.shape
    Ts = s_hid.shape[1]
    Tmax = max(Tt, Ts)
    dev = t_hid.device
    perm = torch.arange(2 * B, device=dev) if is_deterministic() else ops.randperm(2 * B, next_seed(), 1, dev)
    tl, sl = lens_i32(t_hid_len, dev), lens_i32(s_hid_len, dev)
    d_len = torch.empty(2 * B, dtype=torch.int32, device=dev)
    d_target = torch.empty(2 * B, dtype=torch.float32, device=dev)
    ops.disc_targets(perm, B, not train_discriminator, d_target)

    def run(tape, dummy, th, sh):
        thc, shc = th.v.contiguous(), sh.v.contiguous()
        dst = out if out is not None else torch.empty(2 * B, Tmax, Dm, dtype=torch.float32, device=dev)
        ops.disc_gather(thc, shc, tl, sl, perm, dst, d_len)
        o = Var(dst)
        if tape is not None:
            def bwd():
                if o.g is None:
                    return
                g = o.g if o.g.is_contiguous() else o.g.contiguous()
                dth = torch.empty(B, Tt, Dm, dtype=torch.float32, device=dev)
                dsh = torch.empty(B, Ts, Dm, dtype=torch.float32, device=dev)
                ops.disc_scatter(g, perm, dth, dsh)
                th.g, sh.g = dth, dsh
            tape.record(bwd)
        return [o]
    if t_hid.requires_grad or s_hid.requires_grad:
        d_hid = run_segment(run, None, _dummy(dev), t_hid, s_hid)
    else:
        with torch.no_grad():
            d_hid = run_segment(run, None, _dummy(dev), t_hid, s_hid)
    return (d_hid, d_len, d_target)


def discriminator_hidden_to_loss(model, d_batch, freeze_discriminator=False):
    """src/train.py:331-335."""
    d_hid, d_len, d_target = d_batch
    d_out = model.discriminator(d_hid, d_len)
    d_loss = discriminator_loss(d_out, d_target)
    return d_loss, (d_out, d_target)


def discriminator_step(model, batch, args):
    """src/train.py:337-354."""
    x, _ = batch
    text, mel, text_len, mel_len = x
    with torch.no_grad():
        t_enc_out, _ = model.text_m.encode(text, text_len)
        s_enc_out, _ = model.speech_m.encode(mel, mel_len)
    d_batch = discriminator_shuffle_batch(t_enc_out, text_len, s_enc_out, mel_len, args.model_type, train_discriminator=True)
    d_loss, d_output = discriminator_hidden_to_loss(model, d_batch)
    return d_loss, d_output


#####---- Use these to train on a task -----#####
def optimizer_step(model, optimizer, args, defer=False):
    """src/train.py:358-363: clip_grad_norm_ -> optimizer.step() -> zero_grad(set_to_none=True)."""
    if isinstance(optimizer, FusedAdamW):
        st = model._store()
        ds = stream_of("disc") if (defer and st.touched == {"disc"}) else None
        if ds is not None:                     # D phase: clip + AdamW + zero_grad of the discriminator range on its own stream
            with torch.cuda.stream(ds):
                join_wgrad_streams()           # its weight gradients were accumulated on the companion stream
                optimizer.step(max_norm=float(args.grad_clip), zero_grads=True)     # (the AdamW pass leaves the range's gradients zero)
                optimizer.zero_grad(set_to_none=True)
            return
        join_streams()                         # gradients were written by up to three streams
        optimizer.step(max_norm=float(args.grad_clip), zero_grads=True)
        optimizer.zero_grad(set_to_none=True)
        return
    join_streams()
    model.expose_grads()
    if args.grad_clip > 0.0:
        nn.utils.clip_grad_norm_(model.parameters(), args.grad_clip)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    model._store().zero_grad()


def train_sp_step(losses, model, batch, step, accum_steps, args):
    """src/train.py:365-390."""
    batch = process_batch(batch)
    with side_streams():                       # text side / speech side / discriminator on three HIP streams
        if args.use_discriminator:
            asr_loss, tts_loss, d_sp_loss = supervised_step(model, batch, args, args.use_discriminator)
            join_streams()     # the loss scalars come from three streams
            loss = _sum_and_backward([tts_loss, asr_loss, d_sp_loss], accum_steps)
        else:
            asr_loss, tts_loss = supervised_step(model, batch, args)
            join_streams()
            loss = _sum_and_backward([tts_loss, asr_loss], accum_steps)
    losses['asr_'].append(_log(asr_loss))
    losses['tts_'].append(_log(tts_loss))
    if args.use_discriminator:
        losses['sp_d'].append(_log(d_sp_loss))
    return loss


def train_ae_step(losses, model, batch, step, accum_steps, args):
    """src/train.py:392-416."""
    batch = process_batch(batch)
    with side_streams():
        if args.use_discriminator:
            t_ae_loss, s_ae_loss, d_ae_loss = autoencoder_step(model, batch, args, args.use_discriminator)
            join_streams()
            loss = _sum_and_backward([t_ae_loss, s_ae_loss, d_ae_loss], accum_steps)
        else:
            t_ae_loss, s_ae_loss = autoencoder_step(model, batch, args)
            join_streams()
            loss = _sum_and_backward([t_ae_loss, s_ae_loss], accum_steps)
    losses['t_ae'].append(_log(t_ae_loss))
    losses['s_ae'].append(_log(s_ae_loss))
    if args.use_discriminator:
        losses['d_ae'].append(_log(d_ae_loss))
    return loss


_JOINT_D_FIRST = os.environ.get("UNAST_JOINT_D_FIRST", "0") == "1"       # experiment switches of train_gen_joint_step
_JOINT_D_PAIR = os.environ.get("UNAST_JOINT_D_PAIR", "1") != "0"


class _JoinRows(torch.autograd.Function):
    """`buf` already holds the row blocks `parts` (each a view of it, written in place by its producer): returns buf as ONE tensor that
    autograd connects to every block's producer; the gradient goes back as row-block views.  (torch.cat would copy 2 x 52 MB per call.)"""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.rows = [p.shape[0] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, r = [], 0
        for n in ctx.rows:
            outs.append(g[r:r + n])
            r += n
        return (None,) + tuple(outs)


def train_gen_joint_step(losses, model, ae_batch, sp_batch, step, accum_steps, args):
    """train_ae_step(ae_batch) followed by train_sp_step(sp_batch) (src/train.py:392-416, 365-390, called back to back before the one
    optimizer_step of the generator phase, src/train.py:609-628) as ONE forward and ONE backward.  Gradients of the two sub-steps add up
    before the update in the reference too, so the joint backward of (ae losses + sp losses) / accum_steps changes no result; what it
    buys: each encoder runs its stack once over both sub-steps' batches (encode_pair: 2B sequences per launch -- the text side's
    45-180-workgroup GEMMs double --, BatchNorm statistics / running-stat updates stay per sub-step and in the reference's order), the
    frozen LSTM discriminator scores both sub-steps' encoder outputs in one call (4B sequences: 256 workgroups per recurrent launch
    instead of 128, 8 launches per step instead of 12), and the two sub-steps' decoders are independent work on two streams.
    Needs both batches in one shape; otherwise (or with config.JOINT_GEN off) the caller runs the two sub-steps one after the other."""
    from . import config
    (xa, ya), (xs, ys) = process_batch(ae_batch), process_batch(sp_batch)
    text_a, mel_a, tl_a, ml_a = xa
    text_s, mel_s, tl_s, ml_s = xs
    use_d = bool(args.use_discriminator)
    with side_streams():
        mel_aug = mel_s if is_deterministic() else specaugment(mel_s, ml_s)
        (t_enc_a, t_masks_a), (t_enc_s, t_masks_s) = model.text_m.encode_pair(text_a, tl_a, True, text_s, tl_s, False)
        (s_enc_a, s_masks_a), (s_enc_s, s_masks_s) = model.speech_m.encode_pair(mel_a, ml_a, True, mel_aug, ml_s, False)
        def disc_losses():
            if not use_d:
                return None, None
            if _JOINT_D_PAIR:       # both sub-steps' discriminator batches side by side in one buffer, one discriminator call
                B, Tmax, Dm = text_a.shape[0], max(text_a.shape[1], mel_a.shape[1]), t_enc_a.shape[-1]
                with torch.cuda.stream(stream_of("disc") or torch.cuda.current_stream()):
                    d_buf = torch.empty(4 * B, Tmax, Dm, dtype=torch.float32, device=t_enc_a.device)
                d_a = discriminator_shuffle_batch(t_masks_a[2], tl_a, s_masks_a[2], ml_a, args.model_type, out=d_buf[:2 * B])
                d_s = discriminator_shuffle_batch(t_masks_s[2], tl_s, s_masks_s[2], ml_s, args.model_type, out=d_buf[2 * B:])
                return _discriminator_pair_losses(model, d_buf, d_a, d_s)
            d_a = discriminator_shuffle_batch(t_masks_a[2], tl_a, s_masks_a[2], ml_a, args.model_type)
            la_, _ = discriminator_hidden_to_loss(model, d_a, freeze_discriminator=True)
            d_s = discriminator_shuffle_batch(t_masks_s[2], tl_s, s_masks_s[2], ml_s, args.model_type)
            ls_, _ = discriminator_hidden_to_loss(model, d_s, freeze_discriminator=True)
            return la_, ls_
        if _JOINT_D_FIRST:
            d_ae_loss, d_sp_loss = disc_losses()
        # the auto-encoder sub-step's decoders, then the supervised sub-step's (BatchNorm of the speech post-net: first ae, then tts)
        gs = 1.0 / float(accum_steps)          # the upstream gradient of every loss of this step (train._LossSum): the fused head + loss launches apply it
        dev = t_enc_a.device
        th_a, th_s = (ya[0], args.t_eos_weight, gs, _loss_ws(dev)), (ys[0], args.t_eos_weight, gs, _loss_ws(dev))
        pair_text = config.JOINT_DECODERS and text_a.shape == text_s.shape
        if pair_text:           # the auto-encoder's text decoder and the ASR decoder as one call (memories: text / speech encoder output)
            text_pred_a, text_pred_s = (o.permute(0, 2, 1) for o in model.text_m.decode_pair(text_a, tl_a, t_enc_a, t_masks_a, th_a,
                                                                                             text_s, tl_s, s_enc_s, s_masks_s, th_s))
        else:
            text_pred_a = model.text_m.decode_sequence(text_a, tl_a, t_enc_a, t_masks_a, loss_hint=th_a).permute(0, 2, 1)
        hint_a, hint_s = (ya[1], ml_a, args.s_eos_weight, gs, _loss_ws(dev)), (ys[1], ml_s, args.s_eos_weight, gs, _loss_ws(dev))
        if config.JOINT_DECODERS and mel_a.shape == mel_s.shape:
            # the auto-encoder's speech decoder and the TTS decoder as one call: the stack once over both, cross-attention per call
            (pre_a, post_a, stop_a, _), (pre_s, post_s, stop_s, _) = model.speech_m.decode_pair(mel_a, ml_a, s_enc_a, s_masks_a, hint_a,
                                                                                                 mel_s, ml_s, t_enc_s, t_masks_s, hint_s)
        else:
            pre_a, post_a, stop_a, _ = model.speech_m.decode_sequence(mel_a, ml_a, s_enc_a, s_masks_a, loss_hint=hint_a)
            pre_s, post_s, stop_s, _ = model.speech_m.decode_sequence(mel_s, ml_s, t_enc_s, t_masks_s, loss_hint=hint_s)
        if not pair_text:
            text_pred_s = model.text_m.decode_sequence(text_s, tl_s, s_enc_s, s_masks_s, loss_hint=th_s).permute(0, 2, 1)
        if not _JOINT_D_FIRST:
            d_ae_loss, d_sp_loss = disc_losses()
        s_ae_loss = speech_loss(ya[1], ya[2], pre_a, post_a, ml_a, stop_a, args.s_eos_weight)
        t_ae_loss = text_loss(ya[0], text_pred_a, args.t_eos_weight)
        tts_loss = speech_loss(ys[1], ys[2], pre_s, post_s, ml_s, stop_s, args.s_eos_weight)
        asr_loss = text_loss(ys[0], text_pred_s, args.t_eos_weight)
        join_streams()
        parts = [t_ae_loss, s_ae_loss] + ([d_ae_loss] if use_d else []) + [tts_loss, asr_loss] + ([d_sp_loss] if use_d else [])
        loss = _sum_and_backward(parts, accum_steps)
    losses['t_ae'].append(_log(t_ae_loss))
    losses['s_ae'].append(_log(s_ae_loss))
    if use_d:
        losses['d_ae'].append(_log(d_ae_loss))
    losses['asr_'].append(_log(asr_loss))
    losses['tts_'].append(_log(tts_loss))
    if use_d:
        losses['sp_d'].append(_log(d_sp_loss))
    return loss


@on_stream("disc")
def _discriminator_pair_losses(model, d_buf, d_a, d_s):
    """discriminator_hidden_to_loss of two discriminator batches whose hidden states sit side by side in `d_buf`: one discriminator
    call over both (the two batches' rows keep their own dropout decisions: masks are drawn per row), one loss per batch."""
    n = d_a[0].shape[0]
    d_hid = _JoinRows.apply(d_buf, d_a[0], d_s[0])
    d_out = model.discriminator(d_hid, torch.cat([d_a[1], d_s[1]]))
    return discriminator_loss(d_out[:n], d_a[2]), discriminator_loss(d_out[n:], d_s[2])


def joint_generator_phase(args, ae_batch, sp_batch):
    """Whether the generator phase of one outer step can run as train_gen_joint_step: one auto-encoder and one supervised sub-step, no
    cross-model sub-step between them, and the two batches in one shape."""
    from . import config
    if not config.JOINT_GEN or args.ae_steps != 1 or args.sp_steps != 1 or getattr(args, "cm_steps", 0) != 0:
        return False
    return all(tuple(a.shape) == tuple(b.shape) for a, b in zip(ae_batch[:2], sp_batch[:2]))


def train_cm_step(losses, model, batch, step, accum_steps, args):
    """src/train.py:418-444."""
    batch = process_batch(batch)
    if args.use_discriminator:
        t_cm_loss, s_cm_loss, d_cm_loss = crossmodel_step(model, batch, args, args.use_discriminator)
        loss = s_cm_loss + t_cm_loss + d_cm_loss
    else:
        t_cm_loss, s_cm_loss = crossmodel_step(model, batch, args)
        loss = s_cm_loss + t_cm_loss
    loss = loss / accum_steps
    loss.backward()
    losses['s_cm'].append(_log(s_cm_loss))
    losses['t_cm'].append(_log(t_cm_loss))
    if args.use_discriminator:
        losses['d_cm'].append(_log(d_cm_loss))
    return loss


def train_discriminator_step(losses, model, batch, step, accum_steps, args, log_out_to_tb=False, defer=False):
    """src/train.py:446-463.  `defer` (used by the training loop, not part of the reference signature): leave the D phase on
    the discriminator's stream instead of joining it into the caller's stream."""
    batch = process_batch(batch)
    with side_streams() as ctx:
        d_loss, d_output = discriminator_step(model, batch, args)
        ds = stream_of("disc") if (ctx.active and defer) else None
        if ds is None:
            join_streams()
            loss = _sum_and_backward([d_loss], accum_steps)
        else:
            # The whole D phase (forward, backward, and the optimizer step that follows) stays on the discriminator's stream:
            # with that stream ambient during backward the caller's stream never waits for it, so the next step's generator
            # forward (which does not read D's weights until its own D call, issued on this same stream) overlaps it.
            with torch.cuda.stream(ds):
                loss = _sum_and_backward([d_loss], accum_steps)
            ctx.leave_open = True
    losses['d'].append(_log(d_loss))
    return loss


def freeze_model_parameters(model):
    """src/train.py:465-467."""
    for param in model.parameters():
        param.requires_grad = False


def unfreeze_model_parameters(model):
    """src/train.py:469-471."""
    for param in model.parameters():
        param.requires_grad = True


def train_step(losses, model, optimizer, scheduler, batches, step, args, defer_d_phase=False):
    """One iteration of the hot loop of train() (src/train.py:602-655).
    `batches` = dict(unsup=[...ae_steps batches], cm=[...cm_steps], sup=[...sp_steps], disc=[...d_steps]).
    defer_d_phase=True leaves the discriminator phase (backward, clip + AdamW) on its own HIP stream so that the NEXT call's
    generator forward overlaps it; the caller then has to `join_streams()` (or synchronise the device) before reading
    parameters, gradients or losses from its own stream.  train() does that at every epoch end."""
    if not model.training:
        model.train()
    if args.use_discriminator:
        freeze_model_parameters(model.discriminator)
    accum_steps = args.ae_steps + getattr(args, "cm_steps", 0) + args.sp_steps
    if args.ae_steps == 1 and args.sp_steps == 1 and joint_generator_phase(args, batches["unsup"][0], batches["sup"][0]):
        ddp.arm()              # one backward for the whole generator phase: every bucket is final when its users have run theirs
        train_gen_joint_step(losses, model, batches["unsup"][0], batches["sup"][0], step, accum_steps, args)
    else:
        subs = [(train_ae_step, batches["unsup"][si]) for si in range(args.ae_steps)]
        subs += [(train_cm_step, batches["cm"][si]) for si in range(getattr(args, "cm_steps", 0))]
        subs += [(train_sp_step, batches["sup"][si]) for si in range(args.sp_steps)]
        for i, (fn, b) in enumerate(subs):
            if i == len(subs) - 1:
                ddp.arm()          # gradients become final in this sub-step: buckets travel as soon as their backward is enqueued
            fn(losses, model, b, step, accum_steps, args)
    optimizer_step(model, optimizer, args)
    if args.use_discriminator:
        unfreeze_model_parameters(model.discriminator)
        for si in range(args.d_steps):
            train_discriminator_step(losses, model, batches["disc"][si], step, args.d_steps, args, defer=defer_d_phase)
        optimizer_step(model, optimizer, args, defer=defer_d_phase)
    if scheduler is not None:
        scheduler.step()


class SyntheticBatchGetter:
    """Stand-in for BatchGetter (src/train.py:32-78): the reference iterates three DataLoaders over LJSpeech; the data
    pipeline is out of scope, so batches here are synthetic with the collate contract
    (text int64 [B,Tt] EOS-terminated and zero padded, mel float32 [B,Tm,80], lengths, sorted by text length)."""

    def __init__(self, args, t_text=180, t_mel=800, ragged=True, seed=0):
        self.B, self.Tt, self.Tm, self.ragged, self.seed, self.n = args.train_batch_size, t_text, t_mel, ragged, seed, 0

    def _next(self):
        from .portable import synth_batch
        self.n += 1
        return tuple(torch.from_numpy(x) for x in synth_batch(self.B, self.Tt, self.Tm, seed=self.seed + self.n, ragged=self.ragged))

    get_supervised_batch = get_unsupervised_batch = get_discriminator_batch = _next


def log_loss_metrics(losses, epoch, eval=False):
    """src/train.py:756-764 (one host read per logged key, at the epoch boundary)."""
    kind = "Eval_" if eval else "Train"
    out = {k: float(torch.stack([torch.as_tensor(x, dtype=torch.float32).cpu() for x in v]).mean()) for k, v in losses.items() if len(v)}
    print("{} epoch {:-3d} ".format(kind, epoch) + " ".join("%s %.4f" % kv for kv in sorted(out.items())))
    return out


def train(args, batch_getter=None, on_epoch_end=None, valid_dataloader=None):
    """The hot loop of the reference's train() (src/train.py:567-696) without TensorBoard / dataset code (evaluation and
    best-PER checkpointing run at each epoch end when `valid_dataloader` is given, src/train.py:668-679):
    per outer step ae_steps x AE, sp_steps x SP (accumulated, scaled by 1/accum_steps), optimizer_step, then d_steps x D,
    optimizer_step, scheduler.step().  Returns (model, per-epoch loss means)."""
    set_seed(args.seed)
    batch_getter = batch_getter or SyntheticBatchGetter(args)
    s_epoch, best, model, optimizer, scheduler = initialize_model(args)
    cm_steps = getattr(args, "cm_steps", 0)
    max_obj_steps = max(args.ae_steps, cm_steps, args.sp_steps, args.d_steps if args.use_discriminator else 0)
    accum_steps = args.ae_steps + cm_steps + args.sp_steps
    history = []
    # args.use_hip_graphs (not a reference key; default off): the body of the hot loop as a captured graph (unast_amd.graphed), one
    # capture per pair of input shapes met twice, kept in an LRU -- for loaders whose batches repeat shapes (length-bucketed samplers)
    # on host-bound configurations.  Batches must come as they are: padding them up to a bucket changes the result.
    stepper = None
    if getattr(args, "use_hip_graphs", False) and cm_steps == 0:
        from .graphed import GraphedTrainStep
        stepper = GraphedTrainStep(model, optimizer, scheduler, args)
        model.__dict__["_graph_stepper"] = stepper         # (its cache_report() is how a caller sees captures / replays / evictions)
    for epoch in range(s_epoch, args.epochs):
        losses = defaultdict(list)
        for s in range(args.epoch_steps):
            model.train()
            base = epoch * args.epoch_steps * max_obj_steps + s * max_obj_steps
            if stepper is not None:
                batches = dict(unsup=[batch_getter.get_unsupervised_batch() for _ in range(args.ae_steps)],
                               sup=[batch_getter.get_supervised_batch() for _ in range(args.sp_steps)],
                               disc=[batch_getter.get_discriminator_batch() for _ in range(args.d_steps if args.use_discriminator else 0)])
                stepper(losses, batches, base)             # includes scheduler.step()
                continue
            if args.use_discriminator:
                freeze_model_parameters(model.discriminator)
            subs = [(train_ae_step, batch_getter.get_unsupervised_batch, si) for si in range(args.ae_steps)]
            subs += [(train_cm_step, batch_getter.get_unsupervised_batch, si) for si in range(cm_steps)]
            subs += [(train_sp_step, batch_getter.get_supervised_batch, si) for si in range(args.sp_steps)]
            for i, (fn, get, si) in enumerate(subs):
                if i == len(subs) - 1:
                    ddp.arm()                      # last generator sub-step: gradient buckets travel during its backward
                fn(losses, model, get(), base + si, accum_steps, args)
            optimizer_step(model, optimizer, args)
            if args.use_discriminator:
                unfreeze_model_parameters(model.discriminator)
                for si in range(args.d_steps):
                    step = epoch * args.epoch_steps * max_obj_steps + s * max_obj_steps + si
                    train_discriminator_step(losses, model, batch_getter.get_discriminator_batch(), step, args.d_steps, args, defer=True)
                optimizer_step(model, optimizer, args, defer=True)
            if scheduler is not None:
                scheduler.step()
        if stepper is not None:
            stepper.flush(losses)                  # the last step's discriminator phase, before losses / checkpoints / evaluation
        join_streams()                             # the last D phase may still be running on its own stream
        history.append(log_loss_metrics(losses, epoch))
        if not all(v == v and abs(v) < float("inf") for v in history[-1].values()):
            raise RuntimeError("Loss is NaN")               # the reference's check_nan_loss, once per epoch instead of per sub-step
        if getattr(args, "checkpoint_path", None):
            from .checkpoint import save_ckp
            save_ckp(epoch, 300.0, model, optimizer, False, args.checkpoint_path)
        if valid_dataloader is not None:
            step = (epoch + 1) * args.epoch_steps * max_obj_steps - 1
            per, eval_losses = evaluate(model, valid_dataloader, step, args)
            history[-1].update({"eval/" + k: v for k, v in log_loss_metrics(eval_losses, epoch, eval=True).items()})
            history[-1]["eval/per"] = per
            if getattr(args, "checkpoint_path", None):
                save_ckp(epoch, per, model, optimizer, per < best, args.checkpoint_path)
            print("Eval_ epoch {:-3d} PER {:0.3f}%".format(epoch, per * 100))
            best = min(best, per)
        if on_epoch_end is not None:
            on_epoch_end(epoch, model, optimizer, history[-1])
    model.eval()
    return model, history


####---- Evaluation (src/train.py:474-565, 978-983) ----####
def compute_d_score(outputs, targets):
    """src/train.py:978-983: number of discriminator outputs on the right side of 0.5."""
    return torch.sum(torch.round(torch.sigmoid(outputs)) == torch.round(targets))


def evaluate(model, valid_dataloader, step, args, is_test=False):
    """src/train.py:474-565.  Expects paired speech & text batches.  Returns (per, losses[, d_score]).
    model.eval(): BatchNorm uses running statistics, dropout is off; noise_fn / specaugment stay active as in the
    reference (they do not look at model.training)."""
    import json
    import numpy as np
    from .utils import compute_per, compare_outputs
    join_streams()
    if is_test:
        os.makedirs(os.path.join(args.out_test_dir, 'mels'), exist_ok=True)
    model.eval()
    with torch.no_grad():
        losses = defaultdict(list)
        per, n_iters, d_score = 0, 0, 0
        text_pred_dict = {}
        text = None
        for batch in valid_dataloader:
            if is_test:
                batch, fnames = batch
            batch = process_batch(batch)
            x, _ = batch
            text, mel, text_len, mel_len = x
            out = autoencoder_step(model, batch, args, args.use_discriminator)
            losses['t_ae'].append(out[0].item())
            losses['s_ae'].append(out[1].item())
            if args.use_discriminator:
                losses['d_ae'].append(out[2].item())
            out = supervised_step(model, batch, args, args.use_discriminator)
            losses['asr'].append(out[0].item())
            losses['tts'].append(out[1].item())
            if args.use_discriminator:
                losses['d_sp'].append(out[2].item())
            out = crossmodel_step(model, batch, args, args.use_discriminator)
            losses['s_cm'].append(out[1].item())
            losses['t_cm'].append(out[0].item())
            if args.use_discriminator:
                losses['d_cm'].append(out[2].item())
            if args.use_discriminator:
                d_loss, d_output = discriminator_step(model, batch, args)
                losses['dis'].append(d_loss.item())
                if is_test:
                    d_score += compute_d_score(d_output[0], d_output[1]).item() / args.eval_batch_size / 2
            text_pred, text_pred_len = model.asr(None, None, mel, mel_len, infer=True)
            per += compute_per(text, text_pred, text_len, text_pred_len)
            n_iters += 1
            if is_test:
                tp, tpl = text_pred.cpu(), text_pred_len.cpu()
                for gt, gt_len, pred, pred_len, fname in zip(text.cpu(), text_len.cpu(), tp, tpl, fnames):
                    text_pred_dict[fname] = {'gt': gt.tolist()[:gt_len.item()], 'pred': pred.tolist()[:pred_len.item()]}
                _, post_pred, _, stop_lens = model.tts(text, text_len, None, None, infer=True)
                for pred, stop_len, fname in zip(post_pred.cpu(), stop_lens.cpu(), fnames):
                    np.save(os.path.join(args.out_test_dir, 'mels', fname + '.pt'), pred.numpy()[:stop_len.item()])
        if n_iters == 0:
            raise ValueError("evaluate: empty dataloader")
        compare_outputs(text[-1], text_pred[-1], text_len[-1], text_pred_len[-1])
    if is_test:
        json.dump(text_pred_dict, open(os.path.join(args.out_test_dir, 'text_preds.json'), 'w'))
        return per / n_iters, losses, d_score / n_iters
    return per / n_iters, losses


def evaluate_main(args, test_dataloader):
    """src/train.py:985-999 with the dataloader passed in (the LJSpeech dataset code is outside this path)."""
    set_seed(args.seed)
    s_epoch, _, model, _, _ = initialize_model(args)
    per, eval_losses, d_score = evaluate(model, test_dataloader, s_epoch, args, is_test=True)
    log_loss_metrics(eval_losses, s_epoch, eval=True)
    print("per : {}".format(per))
    print("d_score : {}".format(d_score))
    return per, eval_losses, d_score


#####----- Model, optimizer, scheduler initializations -----#####
def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """src/train.py:859-884."""
    def lr_lambda(current_step: int):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)


def get_transformer_paper_schedule(optimizer, num_warmup_steps, last_epoch=-1):
    """src/train.py:886-907 (note: lr = 0 at step 0)."""
    def lr_lambda(current_step: int):
        if current_step < num_warmup_steps:
            return float(current_step) / max(1.0, float(num_warmup_steps) ** 1.5)
        return 1.0 / max(1.0, float(current_step) ** 0.5)
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.Adam(W) semantics (src/train.py:929-932) over the model's flat buffers: one sum-of-squares launch per
    active range + one clip+AdamW launch per range.  Ranges without gradients in this phase (frozen discriminator in the
    generator phase, generator under no_grad in the D phase, never-used reduce_c_W) are skipped entirely — the
    `grad is None => skip` behaviour the reference relies on.  Works with torch LR schedulers (param_groups[0]['lr'])."""

    def __init__(self, model, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, decoupled=True):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        self.decoupled = decoupled
        self._m = self._v = None
        self._steps = defaultdict(int)
        self._ss = None
        self.last_grad_norm_sq = None
        self._slots = {}                 # gradient range -> index of its hyper-parameter triple in ops.step_state()
        self.captured_ranges = []        # ranges stepped while a HIP graph was being captured, in order

    def _buffers(self, st):
        if self._m is None or self._m.device != st.flat.device or self._m.numel() != st.total:
            self._m = torch.zeros_like(st.flat)
            self._v = torch.zeros_like(st.flat)
            self._ss_by_phase = {}
            self._ss = torch.zeros(1, dtype=torch.float64, device=st.flat.device)

    @torch.no_grad()
    def step(self, max_norm=0.0, closure=None, zero_grads=False):
        """zero_grads: the update pass also zeroes the gradients of the ranges it steps (the zero_grad() that follows it in
        optimizer_step, src/train.py:358-363, then has nothing left to fill)."""
        st = self.model._store()
        self._buffers(st)
        ranges = st.active_ranges()
        if not ranges:
            return
        ddp.finish(st, ranges)                 # waits for the buckets that travelled during the backward, reduces the rest
        g = self.param_groups[0]
        key = tuple(ranges)                    # one norm scalar per phase: the D phase may run on its own stream
        ss = self._ss_by_phase.get(key)
        if ss is None:
            ss = self._ss_by_phase[key] = torch.zeros(1, dtype=torch.float64, device=st.flat.device)
        self._ss = ss
        self._ss.zero_()
        for a, b in ranges:
            ops.sumsq(st.grad[a:b], self._ss)
        self.last_grad_norm_sq = self._ss
        capturing = torch.cuda.is_current_stream_capturing()
        for a, b in ranges:
            if capturing:
                # Recorded into a HIP graph (unast_amd.graphed): the learning rate and the bias corrections are read from device
                # memory at replay time; the host-side step count of this range advances per replay (replay_hyper).
                slot = self._slots.setdefault((a, b), len(self._slots))
                ops.adamw(st.flat[a:b], st.grad[a:b], self._m[a:b], self._v[a:b], self._ss, float(max_norm), 0.0,
                          g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], 0, split_out=st.flat_split[a:b],
                          decoupled=self.decoupled, dev_hyper=ops.hyper_slot(slot), zero_grad=zero_grads)
                self.captured_ranges.append((a, b))
                continue
            self._steps[(a, b)] += 1
            ops.adamw(st.flat[a:b], st.grad[a:b], self._m[a:b], self._v[a:b], self._ss, float(max_norm), float(g["lr"]),
                      g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._steps[(a, b)], split_out=st.flat_split[a:b],
                      decoupled=self.decoupled, zero_grad=zero_grads)
        upd = [r for r, ab in st.regions.items() if ab in ranges]
        if zero_grads:
            st.zeroed_by_step.update(upd)
        st.refresh_T(upd)                      # W^T copies of what was just updated
        st.refresh_planes(upd)                 # ... and its tiled bf16 planes (row-panel GEMM)

    def replay_hyper(self, rng, lr):
        """Host side of one replay of a captured step for the range `rng`: advances its step count and returns
        (slot, [lr, 1 - beta1^t, sqrt(1 - beta2^t)]) for the device block the captured AdamW launch reads."""
        self._steps[rng] += 1
        g = self.param_groups[0]
        return self._slots[rng], ops.adam_hyper(lr, g["betas"][0], g["betas"][1], self._steps[rng])

    def zero_grad(self, set_to_none=True):
        self.model._store().zero_grad()

    # ---- torch.optim.AdamW-compatible (de)serialisation (reference checkpoints: src/utils.py:139-195) ----------------
    def _param_ranges(self, st):
        """[(index, name, param, offset, numel, region_range)] in model.parameters() order (= torch's param indices)."""
        out = []
        names = {id(p): n for n, p in st.params.items()}
        for i, p in enumerate(self.model.parameters()):
            n = names[id(p)]
            o = st.offsets[n]
            rr = next((a, b) for (a, b) in st.regions.values() if a <= o < b)
            out.append((i, n, p, o, p.numel(), rr))
        return out

    def state_dict(self):
        st = self.model._store()
        self._buffers(st)
        state = {}
        for i, n, p, o, k, rr in self._param_ranges(st):
            steps = self._steps.get(rr, 0)
            if steps == 0:
                continue                                   # never updated (e.g. reduce_c_W): torch keeps no state either
            m, v = self._m[o:o + k], self._v[o:o + k]
            if n.endswith(".conv.weight"):                 # tap-major in HBM -> reference [Cout,Cin,5]
                co, ci, ks = p.shape
                m, v = m.view(co, ks, ci).permute(0, 2, 1), v.view(co, ks, ci).permute(0, 2, 1)
            else:
                m, v = m.view(p.shape), v.view(p.shape)
            state[i] = {"step": torch.tensor(float(steps)), "exp_avg": m.detach().cpu().contiguous(), "exp_avg_sq": v.detach().cpu().contiguous()}
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(list(self.model.parameters()))))
        for key, val in (("amsgrad", False), ("maximize", False), ("foreach", None), ("capturable", False), ("differentiable", False), ("fused", None)):
            g.setdefault(key, val)
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        st = self.model._store()
        self._buffers(st)
        self._m.zero_(); self._v.zero_()
        self._steps = defaultdict(int)
        for i, n, p, o, k, rr in self._param_ranges(st):
            ent = sd["state"].get(i, sd["state"].get(str(i)))
            if ent is None:
                continue
            m, v = ent["exp_avg"].to(self._m.device, torch.float32), ent["exp_avg_sq"].to(self._m.device, torch.float32)
            if n.endswith(".conv.weight"):
                m, v = m.permute(0, 2, 1), v.permute(0, 2, 1)
            self._m[o:o + k].copy_(m.reshape(-1))
            self._v[o:o + k].copy_(v.reshape(-1))
            self._steps[rr] = max(self._steps[rr], int(float(ent["step"])))
        g = sd["param_groups"][0]
        for key in ("lr", "weight_decay", "betas", "eps", "initial_lr"):
            if key in g:
                self.param_groups[0][key] = tuple(g[key]) if key == "betas" else g[key]

    def grad_norm(self):
        """Pre-clip global gradient norm of the last step (one host read; for logging/tests)."""
        return math.sqrt(float(self.last_grad_norm_sq.item()))


def allreduce_grads(st, ranges, scale_fn=None):
    """Data-parallel gradient exchange of the active gradient ranges on the caller's stream (new vs. the single-device
    reference): what unast_amd.ddp.finish does for everything that did not already travel during the
    backward pass.  Sum over ranks, then scaled by 1/world with a HIP kernel; no-op when torch.distributed is not
    initialised.  `scale_fn` exists so the exchange logic can be exercised by the world_size-2 gloo test on CPU tensors."""
    ddp._State.scale_fn = scale_fn
    try:
        return ddp.finish(st, ranges)
    finally:
        ddp._State.scale_fn = None


def initialize_model(args):
    """src/train.py:910-959 for model_type == 'transformer'."""
    if args.model_type != 'transformer':
        raise NotImplementedError("only model_type='transformer' is on the MI355X path (SURVEY.md section 2, row 11)")
    text_m, speech_m, discriminator, teacher = TextTransformer(args), SpeechTransformer(args), None, get_teacher_ratio(args)
    if args.use_discriminator:
        discriminator = LSTMDiscriminator(args.hidden, args.disc_hid, bidirectional=args.disc_bidirectional, num_layers=args.disc_num_layers)
    model = UNAST(text_m, speech_m, discriminator, teacher).to(_dev())
    if args.optim_type not in ('adam', 'adamw'):
        raise ValueError("optim_type must be adam or adamw")
    optimizer = FusedAdamW(model, lr=args.lr, weight_decay=args.weight_decay, decoupled=(args.optim_type == 'adamw'))
    s_epoch, best = 0, 300
    if getattr(args, "load_path", None) is not None and os.path.isfile(args.load_path):
        from .checkpoint import load_ckp
        s_epoch, best, model, optimizer = load_ckp(args.load_path, model, optimizer)
    scheduler = None
    if args.sched_type == 'multistep':
        milestones = [i * args.epoch_steps for i in args.lr_milestones]
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=args.lr_gamma, last_epoch=s_epoch * args.epoch_steps - 1)
    elif args.sched_type == 'linear':
        scheduler = get_linear_schedule_with_warmup(optimizer, args.warmup_steps, args.epochs * args.epoch_steps, s_epoch * args.epoch_steps - 1)
    elif args.sched_type == 'transformer':
        scheduler = get_transformer_paper_schedule(optimizer, args.warmup_steps, s_epoch * args.epoch_steps - 1)
    model.teacher.iter = s_epoch
    return s_epoch, best, model, optimizer, scheduler
