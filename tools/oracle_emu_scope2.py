"""Finer than tools/oracle_emu_scope.py: which forward product INSIDE the first text-encoder self-attention carries the error?
Split-bf16 operand rounding emulated in exactly one of: q/k/v in-projection, scores QK^T, P.V, out-projection.  CPU only."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unast_ref as R                              # noqa: E402
from unast_amd.portable import synth_batch, portable_tensor   # noqa: E402
from unast_amd.spec import state_dict_spec                     # noqa: E402

L = 2
sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()}
batch = tuple(torch.from_numpy(x) for x in synth_batch(8, 70, 300, seed=3, ragged=True))
torch.Tensor.float = lambda self: self.double()
torch.set_default_dtype(torch.float64)
orig_mha = R.mha
TARGET = "text_m.encoder.transformer_encoder.layers.0.self_attn."


def emu(a, b, on):
    if not on:
        return a @ b
    ah = a.to(torch.bfloat16).double(); al = (a - ah).to(torch.bfloat16).double()
    bh = b.to(torch.bfloat16).double(); bl = (b - bh).to(torch.bfloat16).double()
    return ah @ bh + ah @ bl + al @ bh


def make_mha(which):
    def mha(xq, xkv, P, pre, nhead, lens_k, causal):
        if pre != TARGET:
            return orig_mha(xq, xkv, P, pre, nhead, lens_k, causal)
        W, bias = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
        E = xq.shape[-1]; hd = E // nhead
        q = emu(xq, W[:E].t(), which == "in_proj") + bias[:E]
        k = emu(xkv, W[E:2 * E].t(), which == "in_proj") + bias[E:2 * E]
        v = emu(xkv, W[2 * E:].t(), which == "in_proj") + bias[2 * E:]
        B, Tq, _ = q.shape; Tk = k.shape[1]
        q = q.view(B, Tq, nhead, hd).transpose(1, 2) / math.sqrt(hd)
        k = k.view(B, Tk, nhead, hd).transpose(1, 2)
        v = v.view(B, Tk, nhead, hd).transpose(1, 2)
        s = emu(q, k.transpose(-1, -2), which == "scores")
        neg = ~R.lens_mask(lens_k, Tk)[:, None, None, :]
        s = s.masked_fill(neg, float("-inf"))
        p = torch.softmax(s, dim=-1)
        o = emu(p, v, which == "pv").transpose(1, 2).reshape(B, Tq, E)
        return emu(o, P[pre + "out_proj.weight"].t(), which == "out_proj") + P[pre + "out_proj.bias"]
    return mha


def grads(which):
    R.mha = make_mha(which)
    m = R.Model({k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, L)
    for n, p in m.P.items():
        if n.startswith("discriminator."):
            p.requires_grad_(False)
    b = (batch[0], batch[1].double(), batch[2], batch[3])
    ae = R.generator_losses(m, b); ae.pop("_ae_out")
    (sum(ae.values()) / 2).backward()
    sp = R.supervised_losses(m, b)
    (sum(sp.values()) / 2).backward()
    return {n: p.grad.clone() for n, p in m.P.items() if p.grad is not None}


ref = grads("none")
watch = ["text_m.prenet.embed.weight", "text_m.prenet.batch_norm3.weight", "text_m.encoder.transformer_encoder.layers.0.self_attn.in_proj_weight"]
for which in ("in_proj", "scores", "pv", "out_proj"):
    g = grads(which)
    print("emulated: %-9s" % which, "  ".join("%.2e %s" % ((g[n] - ref[n]).norm().item() / ref[n].norm().item(), n.split(".")[-2] + "." + n.split(".")[-1]) for n in watch), flush=True)
