"""Times K/V-cached generation at config-3 size: text alone, speech alone, both in lock-step (unast_amd.inference.run_pair).
Usage: python tools/bench_decode.py [text_cap speech_cap]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import train, utils                       # noqa: E402
from unast_amd.configs import make_args                  # noqa: E402
from unast_amd.inference import run_pair                 # noqa: E402
from unast_amd.portable import synth_batch               # noqa: E402

tc, sc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (300, 815)
B, Tt, Tm, L = 32, 180, 800, 4
dev = torch.device("cuda:0")
train.DEVICE = dev
utils.set_seed(0)
args = make_args(num_layers=L)
_, _, model, opt, _ = train.initialize_model(args)
from unast_amd.portable import portable_tensor            # noqa: E402
from unast_amd.spec import state_dict_spec                # noqa: E402
model.load_state_dict({k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L).items()})   # bench.py's weights
model.train()
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(B, Tt, Tm, 0))
(text, mel, tl, ml), _ = train.process_batch(batch)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


with torch.no_grad():
    t_enc, t_masks = model.text_m.encode(text, tl)
    s_enc, s_masks = model.speech_m.encode(mel, ml)
    ms_t, (tok, tlens) = timed(lambda: model.text_m.infer_sequence(s_enc, s_masks, tc))
    ms_s, (pre, post, st, slens) = timed(lambda: model.speech_m.infer_sequence(t_enc, t_masks, sc))
    ms_p, _ = timed(lambda: run_pair(model.text_m.generation(s_enc, s_masks, tc), model.speech_m.generation(t_enc, t_masks, sc)))
    # do the two branches of the lock-step graph overlap?  two generations of the same kind, neither stops early
    ms_tt, _ = timed(lambda: run_pair(model.text_m.generation(s_enc, s_masks, tc), model.text_m.generation(s_enc, s_masks, tc)))
print("text + text in lock-step: %7.1f ms (one alone: %.1f)" % (ms_tt, ms_t))
print("text  : %7.1f ms for %d positions (%.0f us/position)" % (ms_t, tok.shape[1], ms_t * 1e3 / max(tok.shape[1], 1)))
print("speech: %7.1f ms for %d positions (%.0f us/position)" % (ms_s, pre.shape[1], ms_s * 1e3 / max(pre.shape[1], 1)))
print("stop lengths: text", sorted(set(tlens.tolist())), " speech", sorted(set(slens.tolist())))
print("pair  : %7.1f ms (sum of the two alone: %.1f)" % (ms_p, ms_t + ms_s))
