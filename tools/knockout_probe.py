"""Timing-only probe: the replayed config-3 step with one op family knocked out (its launches skipped; results are garbage, the
other launches and their order are unchanged).  The drop against the complete step is an UPPER BOUND of what any speed-up of
that family can give the step.  usage: KNOCK=<name> python tools/knockout_probe.py   (names: see `SETS`)"""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from unast_amd import ops
nop = lambda *a, **k: None
ret0 = lambda *a, **k: (a[3] if len(a) > 3 else None)        # linear_fwd / conv_fwd return their output tensor
SETS = {
    "none": [],
    "lstm": ["lstm_fwd", "lstm_bwd"],
    "ln_fwd": ["layernorm_fwd"], "ln_bwd": ["layernorm_bwd"],
    "bn": ["bn_fwd", "bn_bwd"],
    "attn_fwd": ["attn_fwd"], "attn_bwd": ["attn_bwd"], "attn": ["attn_fwd", "attn_bwd"],
    "wgrad": ["linear_wgrad", "conv_wgrad"],
    "conv": ["conv_fwd", "conv_dgrad", "conv_wgrad"],
    "gemm_all": ["gemm", "wgrad_group"],
}
which = os.environ.get("KNOCK", "none")
for n in SETS[which]:
    setattr(ops, n, nop)
sys.argv = ["bench.py", "--no-cpu-baseline", "--launch", os.environ.get("KNOCK_LAUNCH", "graph"), "--iso-steps", "0"] + sys.argv[1:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
