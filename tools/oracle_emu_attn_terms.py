"""What does dropping a LOW-part term of the attention core's split-bf16 products cost?  

The HIP kernels form every product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (csrc/attention.hip mma3).  The terms whose low part is the
kernel-made operand -- P_lo.V_hi (forward O), P_lo.dO_hi (dV), dS_lo.Q_hi (dK), dS_lo.K_hi (dQ) -- each cost one MFMA of three and the
vector instructions that produce the low part (v_cvt_pk, unpack, subtract, v_cvt_pk: 2.5 per score).  This script evaluates the oracle
in fp64 with the attention core replaced by a torch.autograd.Function that forms exactly the kernels' five products from split operands,
with a chosen set of those terms left out, every other contraction of the model with plain split-bf16 emulation in the forward
(oracle.unast_ref.MATMUL_EMU semantics), and prints per variant the error of the ten outputs the golden tests pin, of the seven losses
and of the gradients against the plain fp64 evaluation -- next to the tolerance the `-m gpu` tests hold (tests/test_gpu_parity.py).
CPU only (test infrastructure: imports oracle/)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unast_ref as R                              # noqa: E402
from unast_amd.portable import synth_batch, portable_tensor   # noqa: E402
from unast_amd.spec import state_dict_spec                     # noqa: E402

torch.Tensor.float = lambda self: self.double()
torch.set_default_dtype(torch.float64)
STATE = {"emu": False, "drop": frozenset(), "side": "both"}


def split64(x):
    hi = x.to(torch.bfloat16).double()
    lo = (x - hi).to(torch.bfloat16).double()
    return hi, lo


def mm3(a, b, drop_a_lo=False):
    ah, al = split64(a)
    bh, bl = split64(b)
    out = ah @ bh + ah @ bl
    if not drop_a_lo:
        out = out + al @ bh
    return out


def mm(a, b):
    return mm3(a, b) if STATE["emu"] else a @ b


class Core(torch.autograd.Function):
    """softmax(q k^T + mask) v for [B,H,T,64] operands, q pre-scaled; the five products as csrc/attention.hip forms them."""

    @staticmethod
    def forward(ctx, q, k, v, neg, drop):
        s = mm3(q, k.transpose(-1, -2)).masked_fill(neg, float("-inf"))
        lse = torch.logsumexp(s, dim=-1, keepdim=True)
        p = torch.exp(s - lse)
        o = mm3(p, v, drop_a_lo="P_lo.V" in drop)
        ctx.save_for_backward(q, k, v, neg, lse, o)
        ctx.drop = drop
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, neg, lse, o = ctx.saved_tensors
        drop = ctx.drop
        s = mm3(q, k.transpose(-1, -2)).masked_fill(neg, float("-inf"))
        p = torch.exp(s - lse)
        dp = mm3(do, v.transpose(-1, -2))
        delta = (do * o).sum(-1, keepdim=True)
        ds = p * (dp - delta)
        dv = mm3(p.transpose(-1, -2), do, drop_a_lo="P_lo.dO" in drop)
        dk = mm3(ds.transpose(-1, -2), q, drop_a_lo="dS_lo.Q" in drop)
        dq = mm3(ds, k, drop_a_lo="dS_lo.K" in drop)
        return dq, dk, dv, None, None


def mha(xq, xkv, P, pre, nhead, lens_k, causal):
    W, bias = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    E = xq.shape[-1]
    hd = E // nhead
    q = R.linear(xq, W[:E], bias[:E])
    k = R.linear(xkv, W[E:2 * E], bias[E:2 * E])
    v = R.linear(xkv, W[2 * E:], bias[2 * E:])
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    q = q.view(B, Tq, nhead, hd).transpose(1, 2) / math.sqrt(hd)
    k = k.view(B, Tk, nhead, hd).transpose(1, 2)
    v = v.view(B, Tk, nhead, hd).transpose(1, 2)
    neg = ~R.lens_mask(lens_k, Tk)[:, None, None, :]
    if causal:
        neg = neg | (torch.arange(Tk)[None, :] > torch.arange(Tq)[:, None])[None, None]
    neg = neg.expand(B, nhead, Tq, Tk)
    if STATE["emu"]:
        side = "text" if pre.startswith("text_m.") else "speech"
        drop = STATE["drop"] if STATE["side"] in ("both", side) else frozenset()
        o = Core.apply(q, k, v, neg, drop)
    else:
        s = (q @ k.transpose(-1, -2)).masked_fill(neg, float("-inf"))
        o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, Tq, E)
    return R.linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


R.mm, R.mha = mm, mha


def run(sd, L, batch, emu, drop=frozenset(), side="both"):
    STATE.update(emu=emu, drop=frozenset(drop), side=side)
    m = R.Model({k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, L)
    for n, p in m.P.items():
        if n.startswith("discriminator."):
            p.requires_grad_(False)
    b = (batch[0], batch[1].double(), batch[2], batch[3])
    ae = R.generator_losses(m, b)
    outs = dict(zip(("ae_logits", "ae_pre", "ae_post", "ae_stop", "ae_t_enc", "ae_s_enc"), [t.detach() for t in ae.pop("_ae_out")]))
    (sum(ae.values()) / 2).backward()
    text, mel, tl, ml = b
    with torch.no_grad():
        pre2, post2, stop2, _ = m.tts(text, tl, mel, ml)
        logits2, _ = m.asr(text, tl, mel, ml)
    outs.update(tts_pre=pre2, tts_post=post2, tts_stop=stop2, asr_logits=logits2)
    sp = R.supervised_losses(m, b)
    (sum(sp.values()) / 2).backward()
    STATE.update(emu=False)
    losses = {k: v.item() for k, v in list(ae.items()) + list(sp.items())}
    return outs, losses, {n: p.grad.clone() for n, p in m.P.items() if p.grad is not None}


def hot(n):
    return n.startswith("text_m.prenet.") or n.startswith("text_m.encoder.transformer_encoder.layers.0.self_attn.in_proj")


def report(tag, ref, got):
    (ro, rl, rg), (o, l, g) = ref, got
    oerr = max(float((o[k] - ro[k]).abs().max() / ro[k].abs().max()) for k in ro)
    lerr = max(abs(l[k] - rl[k]) / max(1.0, abs(rl[k])) for k in rl)
    tot = math.sqrt(sum(float((x ** 2).sum()) for x in rg.values()))
    e_cold, e_hot, n_cold, n_hot = [], [], [], []
    for n, r in rg.items():
        if r.norm().item() < 1e-5 * tot:
            continue
        e = (g[n] - r).norm().item() / r.norm().item()
        ne = abs(g[n].norm().item() - r.norm().item()) / r.norm().item()
        (e_hot if hot(n) else e_cold).append(e)
        (n_hot if hot(n) else n_cold).append(ne)
    print("%-44s outputs %.2e | losses %.2e | grads: median %.2e, max %.2e (norm err %.2e), hot max %.2e (norm err %.2e)" % (
        tag, oerr, lerr, float(np.median(e_cold + e_hot)), max(e_cold), max(n_cold), max(e_hot), max(n_hot)), flush=True)
    return oerr, lerr, max(n_cold), max(n_hot)


TERMS = ("P_lo.V", "P_lo.dO", "dS_lo.Q", "dS_lo.K")

if __name__ == "__main__":
    cases = [("ragged B=4 Tt=24 Tm=64 L=2", 2, (4, 24, 64, 0)), ("ragged B=8 Tt=70 Tm=300 L=2", 2, (8, 70, 300, 3))]
    if "--big" in sys.argv:
        cases.append(("B=2 Tt=180 Tm=800 L=4 (config-3 lengths)", 4, (2, 180, 800, 1)))
    print("tolerances the -m gpu tests hold: outputs 1e-3, losses 2e-4, gradient norms 1e-3 (hot tensors 1e-2)")
    for name, L, (B, Tt, Tm, seed) in cases:
        sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()}
        batch = tuple(torch.from_numpy(x) for x in synth_batch(B, Tt, Tm, seed=seed, ragged=True))
        print("== %s" % name, flush=True)
        ref = run(sd, L, batch, emu=False)
        report("all three terms everywhere (today)", ref, run(sd, L, batch, emu=True))
        for t in TERMS:
            for side in ("both", "speech", "text"):
                report("without %s (%s side%s)" % (t, side, "s" if side == "both" else ""), ref, run(sd, L, batch, emu=True, drop=[t], side=side))
        report("without all four (both sides)", ref, run(sd, L, batch, emu=True, drop=TERMS))
        report("without all four (speech side)", ref, run(sd, L, batch, emu=True, drop=TERMS, side="speech"))
