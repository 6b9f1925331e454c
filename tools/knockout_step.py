"""Timing-only knock-outs of whole op families in the replayed train step: `python tools/knockout_step.py` runs bench.py's config-3 step
once per family with that family's ops replaced by no-ops (RESULTS ARE GARBAGE; only the step time means something) and prints how much
of the step each family accounts for ON THE CRITICAL PATH -- which is not its kernel time: LayerNorm backward's 0.9 ms of kernels are
worth 1.4 ms of step, the discriminator's LSTMs almost nothing.  GPU box, repo root."""
import json, os, subprocess, sys

FAMILIES = {
    "nothing": [],
    "LSTM recurrences (discriminator)": ["lstm_fwd", "lstm_bwd"],
    "convolutions (prenet / postnet GEMMs)": ["conv_fwd", "conv_dgrad", "conv_wgrad"],
    "BatchNorm": ["bn_fwd", "bn_bwd"],
    "weight gradients (grouped + single)": ["wgrad_group", "linear_wgrad"],
    "attention forward": ["attn_fwd"],
    "attention backward (+ delta)": ["attn_bwd"],
    "LayerNorm backward (stand-alone launches)": ["layernorm_bwd"],
    "positional encodings + leaky dropout": ["posenc_fwd", "posenc_bwd", "leaky_dropout"],
}
if os.environ.get("UNAST_KO_OPS") is not None:          # child: patch ops, then run bench.py in this process
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from unast_amd import ops
    for n in [x for x in os.environ["UNAST_KO_OPS"].split(",") if x]:
        setattr(ops, n, lambda *a, **k: None)
    sys.argv = ["bench.py", "--launch", "graph", "--no-cpu-baseline", "--steps", "12", "--warmup", "3"]
    import runpy
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
    sys.exit(0)
base = None
for name, fam in FAMILIES.items():
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, UNAST_KO_OPS=",".join(fam)), capture_output=True, text=True, timeout=600)
    try:
        ms = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        print("%-46s failed: %s" % (name, (r.stderr or r.stdout)[-300:].replace("\n", " ")), flush=True)
        continue
    base = ms if base is None else base
    print("%-46s %7.3f ms/step   (%+.2f)" % ("without " + name if fam else "complete step", ms, ms - base), flush=True)
