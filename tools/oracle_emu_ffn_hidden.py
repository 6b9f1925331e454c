"""What would storing the FFN's hidden activation h = relu(W1 x + b1) as ONE bf16 part cost?  (It is the largest tensor of a layer: 210 MB
written and read per speech-side FFN forward at config 3; as a hi/lo pair it has the bytes of fp32.)  The oracle in fp64 with split-bf16
emulation everywhere (as tools/oracle_emu_attn_terms.py) and, per variant, linear2 formed as h_hi W2_hi + h_hi W2_lo only; prints the error
of the ten pinned outputs, the losses and the gradients against the plain fp64 evaluation next to the tolerances of the -m gpu tests.
CPU only (test infrastructure: imports oracle/)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_emu_attn_terms as E           # patches oracle.unast_ref.mm / mha for the emulation (STATE["emu"])
from oracle import unast_ref as R
from unast_amd.portable import synth_batch, portable_tensor
from unast_amd.spec import state_dict_spec

FFN = {"hi_only": "none"}
orig_ffn = R.ffn


def ffn(x, P, pre):
    side = "text" if pre.startswith("text_m.") else "speech"
    if not (E.STATE["emu"] and FFN["hi_only"] in ("both", side)):
        return orig_ffn(x, P, pre)
    h = torch.relu(R.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"]))
    return E.mm3(h, P[pre + "linear2.weight"].t(), drop_a_lo=True) + P[pre + "linear2.bias"]


R.ffn = ffn

if __name__ == "__main__":
    cases = [("ragged B=4 Tt=24 Tm=64 L=2", 2, (4, 24, 64, 0)), ("ragged B=8 Tt=70 Tm=300 L=2", 2, (8, 70, 300, 3))]
    print("tolerances the -m gpu tests hold: outputs 1e-3, losses 2e-4, gradient norms 1e-3 (hot tensors 1e-2)")
    for name, L, (B, Tt, Tm, seed) in cases:
        sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()}
        batch = tuple(torch.from_numpy(x) for x in synth_batch(B, Tt, Tm, seed=seed, ragged=True))
        print("== %s" % name, flush=True)
        ref = E.run(sd, L, batch, emu=False)
        FFN["hi_only"] = "none"
        E.report("all three terms everywhere (today)", ref, E.run(sd, L, batch, emu=True))
        for side in ("speech", "text", "both"):
            FFN["hi_only"] = side
            E.report("FFN hidden as one bf16 part (%s)" % side, ref, E.run(sd, L, batch, emu=True))
