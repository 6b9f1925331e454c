"""Multi-rank rehearsal on ONE GPU: two ranks share cuda:0 and exchange gradients over gloo, exercising the same code path
the driver launches with RCCL on N GPUs (bench.py under torch.distributed.run).  Checks that both ranks end with identical
parameters (identical all-reduced gradients -> identical AdamW updates), that the gradient buckets travel during the backward
of the last generator sub-step (unast_amd.ddp), that bench.py prints a well-formed line, and -- with a single rank -- that the
RCCL (`nccl`) path itself executes on this box."""
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
dist.init_process_group(os.environ.get("TEST_BACKEND", "gloo"), rank=rank, world_size=world)
dev = torch.device("cuda:0"); train.DEVICE = dev
torch.cuda.set_device(dev)
args = make_args(num_layers=1, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(7); utils.set_deterministic(True)
_, _, model, opt, sched = train.initialize_model(args)
opt.param_groups[0]["lr"] = 1e-3
batch = tuple(torch.from_numpy(x) for x in synth_batch(2, 16, 40, seed=rank, ragged=True))      # different data per rank
losses = defaultdict(list)
from unast_amd import ddp
gnorms = []
_step = opt.step
def spy(*a, **k):
    r = _step(*a, **k)
    gnorms.append(opt.grad_norm())            # global norm of the rank-averaged gradients of this phase (one host read; test only)
    return r
opt.step = spy
for it in range(2):
    train.train_step(losses, model, opt, None, dict(unsup=[batch], sup=[batch], disc=[batch]), it, args, defer_d_phase=bool(it))
from unast_amd.engine import join_streams
join_streams(); torch.cuda.synchronize()
labels = [l[0] for l in ddp._State.log]
if os.environ.get("UNAST_DDP_OVERLAP", "1") != "0":
    # per outer step: the four generator buckets in backward order (decoders before encoders), then the D phase's range
    assert len(labels) == 10 and labels[:5] == labels[5:], labels
    assert set(labels[:2]) == {"text_dec", "speech_dec"} and set(labels[2:4]) == {"speech_enc", "text_enc"} and labels[4] == "rest", labels
    if world > 1:                                  # collectives pair up by issue order: it has to be the same on every rank
        objs = [None] * world
        dist.all_gather_object(objs, labels)
        assert all(o == labels for o in objs), objs
else:
    assert labels == ["rest", "rest"] * 2, labels
a, b = model._store().regions["gen"]
covered = sorted((l[1], l[2]) for l in ddp._State.log[:4]) if labels[0] != "rest" else [(a, b)]
assert covered[0][0] == a and covered[-1][1] == b and all(x[1] == y[0] for x, y in zip(covered, covered[1:])), covered
flat = model._store().flat.detach().cpu()
if world > 1:
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged after a data-parallel step"
    l = torch.tensor([float(losses["s_ae"][0])]); ls = [torch.empty(1) for _ in range(world)]; dist.all_gather(ls, l)
    assert ls[0].item() != ls[1].item(), "ranks should have seen different batches"
assert torch.isfinite(flat).all()
if os.environ.get("TEST_SAVE"):
    torch.save(dict(flat=flat, gnorms=gnorms, losses={k: [float(x) for x in v] for k, v in losses.items()}), os.environ["TEST_SAVE"] + ".%%d" %% rank)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run(tmp_path, world, port, **extra):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("rank %d ok" % r in outs[r] for r in range(world)), outs


def _same(a, b):
    """Two runs of the same two outer steps agree: the global norms of the exchanged gradients (a bucket reduced twice, not at
    all, or unscaled would move them by tens of percent) and the losses to accumulation-order noise; the parameters to a few
    Adam steps of that noise (the first updates are +-lr whatever the gradient's size, so near-zero gradients may flip sign)."""
    assert len(a["gnorms"]) == 4 and len(b["gnorms"]) == 4
    for x, y in zip(a["gnorms"], b["gnorms"]):
        assert abs(x - y) < 2e-4 * abs(y), (a["gnorms"], b["gnorms"])
    for k in a["losses"]:
        for x, y in zip(a["losses"][k], b["losses"][k]):
            assert abs(x - y) < 1e-4 * max(1.0, abs(y)), (k, x, y)
    d = (a["flat"] - b["flat"]).abs()
    assert float(d.max()) <= 4.1e-3 and float((d > 1e-5).float().mean()) < 0.02, (float(d.max()), float((d > 1e-5).float().mean()))


def test_two_ranks_share_one_gpu_gloo(tmp_path):
    """Buckets pre-issued during the backward (overlap on) and the round-1 form (one blocking all-reduce per phase inside the
    optimizer step) end with bit-identical parameters."""
    import torch
    _run(tmp_path, 2, 29541, TEST_SAVE=str(tmp_path / "ov"))
    _run(tmp_path, 2, 29545, TEST_SAVE=str(tmp_path / "blk"), UNAST_DDP_OVERLAP="0")
    _same(torch.load(str(tmp_path / "ov") + ".0"), torch.load(str(tmp_path / "blk") + ".0"))


def test_single_rank_nccl_executes_the_rccl_path(tmp_path):
    """World size 1 over the `nccl` backend (= RCCL on ROCm) with UNAST_DDP_FORCE=1: the collectives, the communication
    stream and its joins run for real on this one-GPU box, and the result equals the non-distributed step."""
    import torch
    _run(tmp_path, 1, 29547, TEST_BACKEND="nccl", UNAST_DDP_FORCE="1", TEST_SAVE=str(tmp_path / "nccl"))
    _run(tmp_path, 1, 29549, TEST_BACKEND="gloo", UNAST_DDP_FORCE="1", UNAST_DDP_OVERLAP="0", TEST_SAVE=str(tmp_path / "ref"))
    _same(torch.load(str(tmp_path / "nccl") + ".0"), torch.load(str(tmp_path / "ref") + ".0"))


def test_bench_single_rank_torchrun_nccl():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, nccl), with one rank and the forced
    collective path: RCCL initialises, the overlapped exchange runs, one JSON line comes out."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29551",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, UNAST_DDP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["dist_backend"] == "nccl" and d["value"] > 0 and d["losses_finite"]


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
