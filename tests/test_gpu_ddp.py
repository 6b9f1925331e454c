"""Multi-rank rehearsal on ONE GPU: two ranks share cuda:0 and exchange gradients over gloo, exercising the same code path
the driver launches with RCCL on N GPUs (bench.py under torch.distributed.run).  Checks that both ranks end with identical
parameters (identical all-reduced gradients -> identical AdamW updates) and that bench.py prints a well-formed line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from collections import defaultdict
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0"); train.DEVICE = dev
args = make_args(num_layers=1, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(7); utils.set_deterministic(True)
_, _, model, opt, sched = train.initialize_model(args)
opt.param_groups[0]["lr"] = 1e-3
batch = tuple(torch.from_numpy(x) for x in synth_batch(2, 16, 40, seed=rank, ragged=True))      # different data per rank
losses = defaultdict(list)
train.train_step(losses, model, opt, None, dict(unsup=[batch], sup=[batch], disc=[batch]), 0, args)
flat = model._store().flat.detach().cpu()
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
assert torch.equal(gathered[0], gathered[1]), "ranks diverged after a data-parallel step"
l = torch.tensor([float(losses["s_ae"][0])]); ls = [torch.empty(1) for _ in range(world)]; dist.all_gather(ls, l)
assert ls[0].item() != ls[1].item(), "ranks should have seen different batches"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_share_one_gpu_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]


def test_bench_two_ranks_torchrun_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "tiny", "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                       # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["losses_finite"]
    assert d["config"]["global_batch"] == 4 and "cpu_baseline" not in d
