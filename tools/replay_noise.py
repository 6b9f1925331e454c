"""How far may five replayed steps and five eager steps of the tiny DDP test model drift apart?  tests/test_gpu_ddp.py compares them to 2e-3;
this runs the same worker in several forms and prints the discriminator-loss trajectories and their differences from the eager run:
  eager again (bit-identical?), eager with every weight scaled by 1 + 1.2e-7 (one ulp: the natural growth of rounding noise over the steps),
  the replayed step, the replayed step with the round-3 stream layout (UNAST_REPLAY_KEEP_CHAINS=0), and hipGraphLaunch (UNAST_GRAPH_REPLAY=0).
usage (GPU box): python tools/replay_noise.py"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_ddp import _run

base = dict(TEST_BACKEND="nccl", UNAST_DDP_FORCE="1")
forms = [
    ("eager", dict(UNAST_NATIVE_COMM="0", TEST_STEPS="5")),
    ("eager again", dict(UNAST_NATIVE_COMM="0", TEST_STEPS="5")),
    ("eager +1ulp", dict(UNAST_NATIVE_COMM="0", TEST_STEPS="5", TEST_PERTURB="1.2e-7")),
    ("eager -1ulp", dict(UNAST_NATIVE_COMM="0", TEST_STEPS="5", TEST_PERTURB="-1.2e-7")),
    ("replay", dict(TEST_GRAPH="1")),
    ("replay again", dict(TEST_GRAPH="1")),
    ("replay r3 layout", dict(TEST_GRAPH="1", UNAST_REPLAY_KEEP_CHAINS="0")),
    ("replay 1 stream", dict(TEST_GRAPH="1", UNAST_SIDE_STREAMS="0")),
    ("replay +1ulp", dict(TEST_GRAPH="1", TEST_PERTURB="1.2e-7")),
]
tmp = pathlib.Path(tempfile.mkdtemp())
ref = None
for i, (name, env) in enumerate(forms):
    _run(tmp, 1, 29700 + 2 * i, TEST_SAVE=str(tmp / ("f%d" % i)), **base, **env)
    r = torch.load(str(tmp / ("f%d" % i)) + ".0")
    if ref is None:
        ref = r
    for k in ("d", "s_ae", "t_ae"):
        if k in r["losses"]:
            diffs = ["%.1e" % abs(x - y) for x, y in zip(r["losses"][k], ref["losses"][k])]
            print("%-18s %-5s %s   |diff to eager| %s" % (name, k, ["%.6f" % x for x in r["losses"][k]], diffs), flush=True)
