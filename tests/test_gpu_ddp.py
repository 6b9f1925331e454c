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
CM = os.environ.get("TEST_CM", "0") == "1"            # cross-model sub-step LAST (sp_steps = 0): its backward runs an encoder before its modality's decoder
args = make_args(num_layers=1, ae_steps=1, sp_steps=0 if CM else 1, d_steps=1, cm_steps=1 if CM else 0)
utils.set_seed(7); utils.set_deterministic(True, fixed_sums=os.environ.get("TEST_FIXED_SUMS", "0") == "1")      # fixed_sums: every fp32 sum in a fixed order
_, _, model, opt, sched = train.initialize_model(args)
opt.param_groups[0]["lr"] = 1e-3
if os.environ.get("TEST_PERTURB"):                     # tools/replay_noise.py: how fast does one ulp in the weights grow over the steps?
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.0 + float(os.environ["TEST_PERTURB"]))
SHAPES = os.environ.get("TEST_RANK_SHAPES", "0") == "1"          # each rank pads to its own lengths, as ranks of a real job do
shape = (2, 16, 40) if (rank == 0 or not SHAPES) else (3, 12, 56)
batch = tuple(torch.from_numpy(x) for x in synth_batch(*shape, seed=rank, ragged=True))      # different data per rank
losses = defaultdict(list)
from unast_amd import ddp
gnorms = []
_step = opt.step
def spy(*a, **k):
    r = _step(*a, **k)
    gnorms.append(opt.grad_norm())            # global norm of the rank-averaged gradients of this phase (one host read; test only)
    return r
if os.environ.get("TEST_GRAPH", "0") != "1":           # (a captured step cannot read a scalar back)
    opt.step = spy
GRAPH = os.environ.get("TEST_GRAPH", "0") == "1"      # the captured step replayed by the stream executor, collectives issued from C++
batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[batch])
if CM:
    model.speech_m.infer_max_len = model.text_m.infer_max_len = 6
    so, to = model.speech_m.infer_sequence, model.text_m.infer_sequence
    model.speech_m.infer_sequence = lambda memory, masks, max_len=6: so(memory, masks, max_len)
    model.text_m.infer_sequence = lambda memory, masks, max_len=6: to(memory, masks, max_len)
EAGER_RANK = int(os.environ.get("TEST_EAGER_RANK", "-1"))          # this rank never captures: it runs the same shifted bodies kernel by kernel
if GRAPH:
    from unast_amd.graphed import GraphedTrainStep
    stepper = GraphedTrainStep(model, opt, None, args)
    assert stepper.capturable(), "the native communicator should make the distributed step capturable"
    if rank == EAGER_RANK:
        class _Never(dict):
            def __contains__(self, k): return False
        stepper.warmed = _Never()
    body_logs = []
    for it in range(5):
        n0 = len(ddp._State.log)
        stepper(losses, batches, it)
        body_logs.append([b - a for (_, a, b) in ddp._State.log[n0:]])
    stepper.flush(losses)
    if rank == EAGER_RANK:
        assert stepper.stats["replays"] == 0 and stepper.stats["eager_bodies"] == 4, stepper.stats
        counts = body_logs[-1]                       # what the last shifted body issued eagerly, in issue order
    else:
        rec = next(iter(stepper.graphs.values()))
        assert rec.plan and rec.plan_info["allreduces"] >= 5, rec.plan_info
        print("plan", rec.plan_info)
        import ctypes
        from unast_amd._lib import lib
        arr = (ctypes.c_longlong * 64)()
        n = lib().unast_graph_plan_allreduce_counts(rec.plan, ctypes.addressof(arr), 64)
        counts = [int(arr[i]) for i in range(n)]
        assert n == rec.plan_info["allreduces"] and stepper.stats["replays"] >= 3, (n, stepper.stats)
    if world > 1:                                    # collectives pair up by issue order: replayed plan == eager issue == the other rank's
        objs = [None] * world
        dist.all_gather_object(objs, counts)
        assert all(o == counts for o in objs), objs
        print("collective order", counts)
else:
    for it in range(int(os.environ.get("TEST_STEPS", "2"))):
        train.train_step(losses, model, opt, None, batches, it, args, defer_d_phase=bool(it))
from unast_amd.engine import join_streams
join_streams(); torch.cuda.synchronize()
if (os.environ.get("TEST_BACKEND", "gloo") == "nccl" or os.environ.get("UNAST_COMM_LIB")) and os.environ.get("UNAST_NATIVE_COMM", "1") != "0":
    assert ddp._Native.handle and ddp._Native.issued > 0, "the exchange should have gone through unast_allreduce"
labels = [l[0] for l in ddp._State.log]
if GRAPH or os.environ.get("TEST_STEPS"):
    pass
elif CM and os.environ.get("UNAST_DDP_OVERLAP", "1") != "0":
    assert sorted(labels[:4]) == ["speech_dec", "speech_enc", "text_dec", "text_enc"], labels
    assert labels.index("speech_dec") < labels.index("speech_enc") and labels.index("text_dec") < labels.index("text_enc"), labels
elif os.environ.get("UNAST_DDP_OVERLAP", "1") != "0":
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
if not GRAPH and not os.environ.get("TEST_STEPS"):
    covered = sorted((l[1], l[2]) for l in ddp._State.log[:4]) if labels[0] != "rest" else [(a, b)]
    assert covered[0][0] == a and covered[-1][1] == b and all(x[1] == y[0] for x, y in zip(covered, covered[1:])), covered
flat = model._store().flat.detach().cpu()
if world > 1:
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged after a data-parallel step"
    assert torch.isfinite(flat).all(), "non-finite parameters (the test communicator fills a mismatched collective with NaN)"
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


def _same(a, b, exact=False):
    """Two runs of the same two outer steps agree: the global norms of the exchanged gradients (a bucket reduced twice, not at
    all, or unscaled would move them by tens of percent) and the losses to accumulation-order noise; the parameters to a few
    Adam steps of that noise (the first updates are +-lr whatever the gradient's size, so near-zero gradients may flip sign)."""
    import torch
    assert len(a["gnorms"]) == len(b["gnorms"])
    if exact:            # runs made with TEST_FIXED_SUMS=1: every fp32 sum in a fixed order -> losses and parameters agree to the BIT (the norms
        for x, y in zip(a["gnorms"], b["gnorms"]):            # are fp64 atomic sums: to 1e-12)
            assert abs(x - y) <= 1e-12 * abs(y), (a["gnorms"], b["gnorms"])
        assert a["losses"] == b["losses"], (a["losses"], b["losses"])
        assert torch.equal(a["flat"], b["flat"]), float((a["flat"] - b["flat"]).abs().max())
        return
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
    _run(tmp_path, 2, 29533, TEST_SAVE=str(tmp_path / "ovf"), TEST_FIXED_SUMS="1")           # ... and to the bit with every fp32 sum in a fixed order
    _run(tmp_path, 2, 29535, TEST_SAVE=str(tmp_path / "blkf"), UNAST_DDP_OVERLAP="0", TEST_FIXED_SUMS="1")
    for r in (0, 1):
        _same(torch.load(str(tmp_path / "ovf") + ".%d" % r), torch.load(str(tmp_path / "blkf") + ".%d" % r), exact=True)


def test_single_rank_nccl_executes_the_rccl_path(tmp_path):
    """World size 1 over the `nccl` backend (= RCCL on ROCm) with UNAST_DDP_FORCE=1: the collectives, the communication
    stream and its joins run for real on this one-GPU box, and the result equals the non-distributed step."""
    import torch
    _run(tmp_path, 1, 29547, TEST_BACKEND="nccl", UNAST_DDP_FORCE="1", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "nccl"))
    _run(tmp_path, 1, 29549, TEST_BACKEND="gloo", UNAST_DDP_FORCE="1", UNAST_DDP_OVERLAP="0", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "ref"))
    _same(torch.load(str(tmp_path / "nccl") + ".0"), torch.load(str(tmp_path / "ref") + ".0"), exact=True)


def test_two_ranks_cross_model_last_substep_overlap_equals_blocking(tmp_path):
    """With a cross-model sub-step as the LAST generator sub-step autograd runs the speech encoder's backward before the speech
    decoder's, and the decoder still adds speech_m.prenet.* gradients into the encoder's bucket: the bucket may only travel after both
    (ddp._buckets_of).  Overlapped and blocking exchange must give the same gradients."""
    import torch
    _run(tmp_path, 2, 29553, TEST_CM="1", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "ov"))
    _run(tmp_path, 2, 29555, TEST_CM="1", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "blk"), UNAST_DDP_OVERLAP="0")
    _same(torch.load(str(tmp_path / "ov") + ".0"), torch.load(str(tmp_path / "blk") + ".0"), exact=True)


_STEP_BOUNDS = [1e-5, 1e-4, 5e-4, 2e-3, 2e-2]          # per step, see the comment in the test below


def test_single_rank_nccl_replayed_step_issues_the_collectives_from_cpp(tmp_path):
    """Under a process group the train step is captured with marker nodes where the gradient buckets are exchanged; the stream-replay
    executor issues unast_allreduce (RCCL through the C ABI) there.  Five replayed steps equal five eager steps of the torch.distributed path."""
    import torch
    _run(tmp_path, 1, 29557, TEST_BACKEND="nccl", UNAST_DDP_FORCE="1", TEST_GRAPH="1", TEST_SAVE=str(tmp_path / "g"))
    _run(tmp_path, 1, 29559, TEST_BACKEND="nccl", UNAST_DDP_FORCE="1", UNAST_NATIVE_COMM="0", TEST_STEPS="5", TEST_SAVE=str(tmp_path / "e"))
    # ... and with every fp32 sum in a fixed order (utils.set_deterministic(True, fixed_sums=True)) the two agree to the BIT, losses and parameters
    _run(tmp_path, 1, 29551, TEST_BACKEND="nccl", UNAST_DDP_FORCE="1", TEST_GRAPH="1", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "gf"))
    _run(tmp_path, 1, 29543, TEST_BACKEND="nccl", UNAST_DDP_FORCE="1", UNAST_NATIVE_COMM="0", TEST_STEPS="5", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "ef"))
    af, bf = torch.load(str(tmp_path / "gf") + ".0"), torch.load(str(tmp_path / "ef") + ".0")
    assert af["losses"] == bf["losses"] and torch.equal(af["flat"], bf["flat"]), (af["losses"], bf["losses"], float((af["flat"] - bf["flat"]).abs().max()))
    a, b = torch.load(str(tmp_path / "g") + ".0"), torch.load(str(tmp_path / "e") + ".0")
    # Bounds per step, not one for all five: two EAGER runs of this worker already differ by 6e-8 / 8e-6 / 4e-5 / 2e-4 / 3e-3 in the
    # discriminator loss (atomic accumulation order; lr 1e-3 on a 2-utterance batch amplifies it ~10x per step), one ulp in the weights
    # grows to 5e-3 by step five -- profiles/r04_replay_noise.txt, tools/replay_noise.py.  A collective issued twice, skipped or on
    # stale gradients shows in step two already, where the bound is 1e-4.
    bounds = _STEP_BOUNDS
    for k in a["losses"]:
        assert len(a["losses"][k]) == len(b["losses"][k]) == 5, (k, len(a["losses"][k]), len(b["losses"][k]))
        for x, y, tol in zip(a["losses"][k], b["losses"][k], bounds):
            assert abs(x - y) < tol * max(1.0, abs(y)), (k, a["losses"][k], b["losses"][k])


SHIM = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")


def _need_shim():
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", os.path.join(ROOT, "unast_amd", "csrc"), "../../tests/native/libfake_rccl.so"], check=True)


def _close(a, b, steps=5):
    for k in a["losses"]:
        assert len(a["losses"][k]) == len(b["losses"][k]) == steps, (k, len(a["losses"][k]), len(b["losses"][k]))
        for x, y, tol in zip(a["losses"][k], b["losses"][k], _STEP_BOUNDS):
            assert abs(x - y) < tol * max(1.0, abs(y)), (k, a["losses"][k], b["losses"][k])


@pytest.mark.parametrize("jitter", ["0", "200"])
def test_two_ranks_replayed_step_with_collectives_from_cpp(tmp_path, jitter):
    """The path `bench.py --gpus N` takes, with TWO ranks: unast_comm_init with world = 2 and a broadcast unique id, the step captured with
    marker nodes, replayed by csrc/graph_exec.cpp which issues unast_allreduce at the markers.  RCCL refuses two ranks on one device, so
    csrc/comm.cpp binds tests/native/fake_rccl.cpp (UNAST_COMM_LIB: host-staged all-reduce with RCCL's stream semantics, which reports a
    collective whose size differs between ranks).  Inside the workers: the same collective order on both ranks, parameters bit-equal
    across ranks after five steps; here: equal to the blocking gloo exchange of the eager step.  With STREAM_JITTER the streams of the
    two ranks are shifted against each other at random: no hang, same result."""
    import torch
    _need_shim()
    _run(tmp_path, 2, 29561 + int(jitter) % 7, TEST_GRAPH="1", UNAST_COMM_LIB=SHIM, UNAST_STREAM_JITTER=jitter, TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "g"))
    _run(tmp_path, 2, 29571 + int(jitter) % 7, UNAST_DDP_OVERLAP="0", TEST_STEPS="5", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "e"))
    # every fp32 sum in a fixed order (TEST_FIXED_SUMS) and a two-rank sum is commutative: the replayed bucketed exchange and the blocking
    # eager one agree to the BIT on both ranks -- a collective on stale or half-written gradients cannot hide under a noise bound
    for r in (0, 1):
        a, b = torch.load(str(tmp_path / "g") + ".%d" % r), torch.load(str(tmp_path / "e") + ".%d" % r)
        assert a["losses"] == b["losses"] and torch.equal(a["flat"], b["flat"]), (r, a["losses"], b["losses"], float((a["flat"] - b["flat"]).abs().max()))


def test_two_ranks_one_replays_one_runs_eagerly_with_other_shapes(tmp_path):
    """Ranks decide eager / capture / replay from their OWN batch shapes (unast_amd/graphed.py), so one rank may replay a plan while another
    issues the same step kernel by kernel from ddp._issue on the same communicator: both must produce the same sequence of collectives.
    Rank 1 here has other padded lengths and never captures; rank 0 replays.  The workers compare the order (plan order on rank 0, eager
    issue order on rank 1) and end with bit-equal parameters; the test communicator would fill a mismatched collective with NaN."""
    import torch
    _need_shim()
    _run(tmp_path, 2, 29581, TEST_GRAPH="1", UNAST_COMM_LIB=SHIM, TEST_RANK_SHAPES="1", TEST_EAGER_RANK="1", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "g"))
    _run(tmp_path, 2, 29583, UNAST_DDP_OVERLAP="0", TEST_STEPS="5", TEST_RANK_SHAPES="1", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "e"))
    a, b = torch.load(str(tmp_path / "g") + ".0"), torch.load(str(tmp_path / "e") + ".0")       # (fixed summation order: to the bit)
    assert a["losses"] == b["losses"] and torch.equal(a["flat"], b["flat"]), (a["losses"], b["losses"], float((a["flat"] - b["flat"]).abs().max()))


def test_native_eager_exchange_two_ranks(tmp_path):
    """The eager step with the C ABI's communicator (event-chained unast_allreduce calls from ddp._issue) on two ranks == the blocking gloo exchange."""
    import torch
    _need_shim()
    _run(tmp_path, 2, 29585, UNAST_COMM_LIB=SHIM, TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "n"))
    _run(tmp_path, 2, 29587, UNAST_DDP_OVERLAP="0", TEST_FIXED_SUMS="1", TEST_SAVE=str(tmp_path / "b"))
    _same(torch.load(str(tmp_path / "n") + ".0"), torch.load(str(tmp_path / "b") + ".0"), exact=True)


def test_bench_two_ranks_torchrun_replayed_collectives():
    """bench.py as the driver launches it for N = 2, ranks sharing the one GPU over gloo + the test communicator: the line says the step was
    replayed with its exchanges issued from C++."""
    _need_shim()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29589",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--backend", "gloo", "--share-gpu", "--launch", "graph"]
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, UNAST_COMM_LIB=SHIM, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["losses_finite"]
    assert "C ABI" in d["gradient_exchange"] and "replay" in d["launch_mode"], (d["gradient_exchange"], d["launch_mode"])
    assert d["graph_replay"]["allreduces"] >= 5, d["graph_replay"]


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
    assert "C ABI" in d["gradient_exchange"] and "auto:" in d["launch_mode"], (d["gradient_exchange"], d["launch_mode"])     # (the captured step was a candidate)


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
