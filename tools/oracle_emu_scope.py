"""Which FORWARD contractions put the 1e-2 error into the gradients of the text prenet / first text-encoder layer?  The oracle is
evaluated in fp64 with split-bf16 operand rounding emulated (oracle.unast_ref.MATMUL_EMU) in a chosen scope only; the gradient
error of the sensitive tensors against the plain fp64 evaluation is printed per scope.  (Backward products are exact in every
scope: autograd differentiates through the emulation.)  CPU only (test infrastructure: imports oracle/)."""
import os, sys
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
orig_mm, orig_mha, orig_enc = R.mm, R.mha, R.Model.text_encode
state = {"on": False}


def split64(x):                       # the 16-17 bits a hi/lo bf16 pair keeps
    hi = x.to(torch.bfloat16).double()
    lo = (x - hi).to(torch.bfloat16).double()
    return hi, lo


def mm(a, b):
    if not state["on"]:
        return a @ b
    ah, al = split64(a)
    bh, bl = split64(b)
    return ah @ bh + ah @ bl + al @ bh


R.mm = mm


def grads(scope):
    def mha(xq, xkv, P, pre, nhead, lens_k, causal):
        hit = scope == "text-enc-l0-attn" and pre == "text_m.encoder.transformer_encoder.layers.0.self_attn."
        prev = state["on"]
        state["on"] = prev or hit
        try:
            return orig_mha(xq, xkv, P, pre, nhead, lens_k, causal)
        finally:
            state["on"] = prev

    def text_encode(self, text, text_len):
        prev = state["on"]
        if scope == "text-prenet":          # the three convolutions in front of the encoder stack only
            P = self.P
            x = self._embed(text)
            state["on"] = True
            for i in (1, 2, 3):
                x = R.conv1d_k5(x, P["text_m.prenet.conv%d.conv.weight" % i], P["text_m.prenet.conv%d.conv.bias" % i], 2)
                x = torch.relu(self._bn(x, "text_m.prenet.batch_norm%d." % i))
            state["on"] = prev
            x = R.pos_enc(x, self.buf["text_m.pos_emb.pe"])
            return R.encoder_stack(x, text_len, P, "text_m.encoder.transformer_encoder.layers.", self.L, self.nhead)
        state["on"] = prev or scope == "text-encoder"
        try:
            return orig_enc(self, text, text_len)
        finally:
            state["on"] = prev
    R.mha, R.Model.text_encode = mha, text_encode
    state["on"] = scope == "everything"
    m = R.Model({k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, L)
    for n, p in m.P.items():
        if n.startswith("discriminator."):
            p.requires_grad_(False)
    b = (batch[0], batch[1].double(), batch[2], batch[3])
    ae = R.generator_losses(m, b)
    ae.pop("_ae_out")
    (sum(ae.values()) / 2).backward()
    sp = R.supervised_losses(m, b)
    (sum(sp.values()) / 2).backward()
    state["on"] = False
    return {n: p.grad.clone() for n, p in m.P.items() if p.grad is not None}


ref = grads("none")
watch = ["text_m.prenet.embed.weight", "text_m.prenet.conv1.conv.weight", "text_m.prenet.batch_norm3.weight",
         "text_m.encoder.transformer_encoder.layers.0.self_attn.in_proj_weight", "text_m.encoder.transformer_encoder.layers.0.linear1.weight",
         "text_m.encoder.transformer_encoder.layers.1.self_attn.in_proj_weight", "speech_m.encoder.transformer_encoder.layers.0.self_attn.in_proj_weight"]
for scope in (sys.argv[1:] or ["text-prenet", "text-enc-l0-attn", "text-encoder", "everything"]):
    g = grads(scope)
    print("split-bf16 operands emulated in the forward of: %s" % scope)
    for n in watch:
        print("   %.2e  %s" % ((g[n] - ref[n]).norm().item() / ref[n].norm().item(), n))
