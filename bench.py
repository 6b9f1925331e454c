#!/usr/bin/env python3
"""Benchmark of the UNAST adversarial train step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
      N > 1 is launched by the driver as torch.distributed.run (one rank per GPU, RCCL); per-GPU batch is fixed (weak scaling).

One "step" = one iteration of the reference's hot loop with ae_steps = sp_steps = d_steps = 1, cm_steps = 0
(src/train.py:602-655): freeze(D) -> AE fwd+bwd (+adversarial term) -> SP fwd+bwd (+adversarial term) -> clip+AdamW ->
unfreeze(D) -> D step fwd+bwd -> clip+AdamW -> scheduler.step(); training mode with every dropout/noise/SpecAugment site
active; synthetic LJSpeech-shaped full-length batch (B=32, T_text=180, T_mel=800, 80 mels), random-init weights of the
transformer_d_trans architecture (L=4, d=256, 4 heads, FFN 1024, 2-layer bi-LSTM discriminator).
value = B * T_mel * n_gpus / step-time  [mel-frames/s].

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant kernel, HIP events recorded
inside the timed region on the launch stream) and `cpu_baseline` (the pinned oracle timed on the host cores on a bounded
sample of the same workload; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import unast_amd  # noqa: E402,F401  (sets its HIP runtime defaults before the first device call)

# dense bf16 MFMA peak and HBM peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
PEAK_MFMA_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
SUSTAINED_MFMA_TFLOPS = 1800.0          # v_mfma_f32_16x16x32_bf16 on random data, all SIMDs busy: 9.2 ns per MFMA per SIMD (tools/mfma_peak.cpp)

# Algorithmic TFLOP of one train step per GPU (full gen+disc step for c3/c5, generator-only for c2).
STEP_TFLOP = {"c3": 4.875, "c2": 0.510, "c5": 15.940}
ATTN_BWD_PRODUCTS = 5          # MFMA products the one-pass backward executes per (query, key) tile (7 in the two-kernel form: S and dP recomputed)
ATTN_BWD_KERNEL = "attn_dkv_kernel<%(n)d,1> (dK, dV and dQ in one pass) + attn_delta_kernel"

WORKLOADS = {
    # name: (B, Tt, Tm, L, use_discriminator)
    "c3": (32, 180, 800, 4, True),      # BASELINE.json configs[2]: the configuration the metric is quoted on
    "c2": (8, 128, 512, 3, False),      # configs[1]: generator-only
    "c5": (32, 300, 2000, 4, True),     # configs[4]: long-form
    "tiny": (2, 24, 64, 2, True),
    "c3-nodisc": (32, 180, 800, 4, False),   # diagnostic: what the discriminator's streams cost the config-3 step
}


class OpTimer:
    """Wraps selected unast_amd.ops entry points with HIP event pairs on the launch stream (torch's current stream is the
    stream every kernel of this package is launched on)."""

    def __init__(self, ops_mod, names):
        self.ops, self.names = ops_mod, names
        self.orig, self.events, self.meta = {}, defaultdict(list), {}
        self.active = True          # event pairs cost ~15 us of host time per launch: the bench switches them on for a sample of the steps

    def __enter__(self):
        for n in self.names:
            f = getattr(self.ops, n)
            self.orig[n] = f

            def wrapped(*a, __f=f, __n=n, **k):
                if not self.active or torch.cuda.is_current_stream_capturing():   # launches recorded into a HIP graph (generation) are not timed
                    return __f(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = __f(*a, **k)
                e1.record()
                if __n != "linear_dgrad_lnbwd" or r:               # (returns False, having launched nothing, where the kernel does not serve the shape)
                    self.events[__n].append((e0, e1, self._key(__n, a, k)))
                return r
            setattr(self.ops, n, wrapped)
        return self

    def _key(self, n, a, k):
        if n == "gemm":
            # (a_mode, b_mode, M, N, K, conv channels of A / B, extra [M,N] operands read by the epilogue: R, G, C if beta)
            conv = k.get("conv", (0, 0, 0, 0))
            extra = int(k.get("R") is not None) + int(k.get("G") is not None) + int(bool(k.get("beta", 0)))
            return (a[0], a[1], a[8], a[9], a[10], conv[1], conv[2], extra)
        if n in ("attn_fwd", "attn_bwd"):
            idx = 6 if n == "attn_fwd" else 11
            return tuple(int(x) for x in a[idx:idx + 5])          # B, H, Tq, Tk, causal
        if n == "wgrad_group":
            return a[0]                                           # ((out, in, tokens, has_bias), ...)
        if n == "linear_dgrad_lnbwd":
            # the input-gradient GEMM with a LayerNorm backward in its epilogue (dy2d, W, R, z, mean, rstd, gamma, dz, dz_drop, ...): a row-panel
            # launch with N = 256 whose epilogue also reads R and z and writes dropout(dz)
            return (int(a[0].shape[0]), 256, int(a[0].shape[1]), int(a[2] is not None) + 1 + int(a[8] is not None))
        if n == "panel_gemm":
            # (M, N, K, extra [M,N] operands the epilogue reads or writes beyond C: residual, gate, the LayerNorm output)
            A, N = a[0], a[3]
            extra = int(k.get("R") is not None) + int(k.get("G") is not None) + int(k.get("ln") is not None)
            return (int(A.shape[0]), int(N), int(k.get("K") or A.shape[1]), extra)
        return ()

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def summary(self):
        """{op: {key: (calls, total_ms)}} after a device synchronize."""
        out = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for n, evs in self.events.items():
            for e0, e1, key in evs:
                rec = out[n][key]
                rec[0] += 1
                rec[1] += e0.elapsed_time(e1)
        return out


def gemm_flops(key):
    M, N, K = key[2:5]
    return 2.0 * M * N * K


def gemm_bytes(key):
    """Algorithmic HBM bytes of one contraction (every fp32 operand element read once, every output
    element written once): A + B + C, plus the [M,N] operands the epilogue reads (residual, gate, C when accumulating).
    Implicit-GEMM convolutions count the activation once (rows x channels), not once per tap."""
    a_mode, b_mode, M, N, K, ca, cb, extra = key
    a_el = M * ca if a_mode == 1 else M * K            # OP_KC_CONV: the [B*T, Cin] input
    b_el = K * cb if b_mode == 4 else N * K            # OP_RC_CONV_WGRAD: the [B*T, Cin] input
    return 4.0 * (a_el + b_el + M * N * (1 + extra))


def panel_flops(key):
    M, N, K, _ = key
    return 2.0 * M * N * K


def panel_bytes(key):
    """Row-panel GEMM: A once, the weight planes once (2 x 2 bytes per element), C once, plus the extra [M,N] operands (residual read, LayerNorm
    output written)."""
    M, N, K, extra = key
    return 4.0 * (M * K + N * K + M * N * (1 + extra))


def group_flops(key):
    return sum(2.0 * M * N * K for (M, N, K, _) in key)


def group_bytes(key):
    """A grouped weight-gradient launch: dY and X of every problem read once, dW read and written once (it accumulates)."""
    return sum(4.0 * (K * M + K * N + 2 * M * N) for (M, N, K, _) in key)


def attn_flops(name, key):
    B, H, Tq, Tk, causal = key
    pairs = Tq * (Tq + 1) / 2 if causal else Tq * Tk
    per = 4.0 * B * H * pairs * 64               # QK^T + PV
    return per if name == "attn_fwd" else per * 2.5      # backward: 5 products (S, dP, dV, dK, dQ)


def csrc_digest():
    """sha256 over the kernel sources: what tools/profile_manifest.py records next to the rocprofv3 summaries it files under profiles/."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "unast_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "unast_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(ROOT, "unast_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profiles_fresh():
    """True when the committed profile summaries this line quotes were taken from the present kernel sources (profiles/manifest.json), False
    when csrc/ has changed since (the quoted rocprof / PMC figures are then history, not this build), None without a manifest."""
    mp = os.path.join(ROOT, "profiles", "manifest.json")
    if not os.path.exists(mp):
        return None
    try:
        ok = json.load(open(mp)).get("csrc_sha16") == csrc_digest()
    except Exception:
        return None
    if not ok:
        print("bench.py: profiles/manifest.json was written for other kernel sources: roofline.traffic / rocprof_avg_launch_us quote an older build", file=sys.stderr)
    return ok


def make_batch(B, Tt, Tm, seed):
    from unast_amd.portable import synth_batch
    return tuple(torch.from_numpy(x) for x in synth_batch(B, Tt, Tm, seed=seed, ragged=False))


def cpu_baseline_worker(Tt, Tm, L, use_disc, Bs, budget_s):
    """Oracle (CPU restatement, pinned against the reference's golden vectors) timed on the host cores: full gen+disc steps
    (same step definition, same batch size as the GPU run): 1 warm-up + up to 3 timed steps, as many as fit the time budget."""
    from oracle import unast_ref as R
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))             # the 1-GPU box's CPU share is 16 cores; more threads only thrash
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L, use_discriminator=use_disc).items()}
    m = R.Model(sd, L)
    m.packed_lstm = True                       # torch's packed-sequence LSTM, as the reference (src/module.py:306,315-316)
    opt = R.AdamW(m.P, lr=1e-3, weight_decay=1e-6)
    batch = make_batch(Bs, Tt, Tm, 0)
    t_start = time.time()
    t0 = time.time()
    R.full_step(m, opt, batch, use_discriminator=use_disc)
    warm = time.time() - t0
    times = []
    while len(times) < 3 and (time.time() - t_start) + (times[-1] if times else warm) < budget_s:
        t0 = time.time()
        R.full_step(m, opt, batch, use_discriminator=use_disc)
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2] if times else warm
    return {"value": round(Bs * Tm / med, 2), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": "full gen+disc steps of the same workload and batch (B=%d, T_text=%d, T_mel=%d, L=%d), fp32 torch-CPU oracle (dropout off): "
                      "1 warm-up (%.1f s) + %d timed step(s), median %.1f s%s" % (Bs, Tt, Tm, L, warm, len(times), med,
                                                                              "" if times else " (only the warm-up fitted the time box: cold figure)")}


def cpu_baseline(B, Tt, Tm, L, use_disc, budget_s):
    """Runs the worker in a child process with a hard time box so the default bench always finishes within minutes."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "print('CPUBASE ' + json.dumps(bench.cpu_baseline_worker(%d, %d, %d, %r, %d, %d)))" % (ROOT, Tt, Tm, L, use_disc, B, budget_s))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s + 60, env=env)
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "mel-frames/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "B=%d steps of the same workload did not finish within the %d s time box" % (B, budget_s + 60)}
    for line in out.stdout.decode().splitlines():
        if line.startswith("CPUBASE "):
            return json.loads(line[len("CPUBASE "):])
    return {"error": out.stderr.decode()[-400:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default=os.environ.get("UNAST_PREC", "bf16x3"), choices=["bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=int, default=110, help="seconds of CPU-oracle stepping (1 warm-up + up to 3 timed steps of the same batch)")
    ap.add_argument("--launch", default=os.environ.get("UNAST_LAUNCH"), choices=["auto", "graph", "eager"],
                    help="graph = replay the captured step (unast_amd.graphed), eager = one Python launch per kernel, auto = time a few untimed "
                         "steps of each before the warm-up and keep the faster (the default: at config 3 the two tie on a box with a fast "
                         "host -- 31.0 vs 31.0 ms/step -- and the replay wins where eager enqueue, 26-30 ms/step, gets close to the GPU's "
                         "30 ms; at config 2 the replay is 9.3 against 17 ms/step)")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--iso-detail", action="store_true", help="print the per-shape table of the isolated steps to stderr")
    ap.add_argument("--iso-steps", type=int, default=2, help="single-stream eager steps after the timed region whose GEMM / attention launches are timed with HIP events")
    ap.add_argument("--cm-steps", type=int, default=0, help="add this many cross-model (back-translation) sub-steps per step; reported "
                    "separately from the headline metric, which is defined with cm_steps = 0 (SURVEY.md section 8d)")
    ap.add_argument("--cm-max-len", type=int, default=0, help="cap of the autoregressive generation inside the cm sub-step (0 = reference "
                    "defaults 815 mel frames / 300 tokens)")
    ap.add_argument("--time-every", type=int, default=0, help="eager form only: HIP-event pairs around the GEMM / attention launches on every N-th step of the timed region (0 = none)")
    ap.add_argument("--profile-ops", action="store_true", help="time every op family (adds event overhead; not for the headline number)")
    ap.add_argument("--backend", default=os.environ.get("UNAST_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) is the product path; gloo only rehearses the multi-rank logic on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or (os.environ.get("UNAST_DDP_FORCE", "0") == "1" and "RANK" in os.environ)     # forced: rehearsal of the RCCL path with one rank
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from unast_amd import config, ops, train, utils
    from unast_amd.configs import make_args
    if os.environ.get("UNAST_AUTOGRAD_ST", "0") == "1":        # experiment: backward closures on the calling thread
        torch.autograd.set_multithreading_enabled(False)
    config.set_precision(a.precision)
    B, Tt, Tm, L, use_disc = WORKLOADS[a.workload]
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=a.cm_steps, use_discriminator=use_disc)
    train.DEVICE = dev
    utils.set_seed(1234)                      # identical random-init weights on every rank
    utils.set_deterministic(False)
    _, _, model, opt, sched = train.initialize_model(args)
    utils.set_seed(1234 + rank)               # per-rank dropout / noise / permutation streams
    batch = make_batch(B, Tt, Tm, seed=rank)
    batch = tuple(t.to(dev) for t in batch)   # inputs resident in HBM before the timed region
    batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[batch] * a.cm_steps)
    if a.cm_steps and a.cm_max_len:
        orig_s, orig_t = model.speech_m.infer_sequence, model.text_m.infer_sequence
        model.speech_m.infer_sequence = lambda memory, masks, max_len=a.cm_max_len: orig_s(memory, masks, max_len)
        model.text_m.infer_sequence = lambda memory, masks, max_len=a.cm_max_len: orig_t(memory, masks, max_len)
        model.speech_m.infer_max_len = model.text_m.infer_max_len = a.cm_max_len
    losses = defaultdict(list)

    native = False
    if dist_on:
        from unast_amd import ddp
        native = bool(ddp.native_comm())       # RCCL through the C ABI: the captured step then carries its gradient exchanges (csrc/comm.cpp)
    can_graph = a.cm_steps == 0 and (not dist_on or native) and not a.profile_ops and a.time_every == 0
    launch = "eager" if (a.no_graph or not can_graph) else (a.launch or "auto")
    stepper = None
    auto_note = None
    if launch in ("graph", "auto"):
        from unast_amd.graphed import GraphedTrainStep
        stepper = GraphedTrainStep(model, opt, sched, args)

    def one_step(i):
        if stepper is not None:          # replay of the captured step (the first three calls run eagerly / capture it)
            stepper(losses, batches, i)
        else:
            train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True)      # as train() does; the timed region ends with a device synchronise

    def sync():
        torch.cuda.synchronize(dev)
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(dev)

    n_prime = 3 if stepper is not None else 0     # eager generator phase, eager shifted step, capture + first replay: before the W warm-up steps
    for i in range(n_prime):
        one_step(i)
    if launch == "auto":
        # untimed calibration: a few steps of each launch mode, keep the faster
        def probe(fn, n=6):
            fn(0); sync()
            t = time.perf_counter()
            for i in range(n):
                fn(i)
            sync()
            return (time.perf_counter() - t) / n * 1e3
        ms_graph = probe(lambda i: stepper(losses, batches, i))
        stepper.flush(losses)
        ms_eager = probe(lambda i: train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True))
        if world > 1:                           # one decision for all ranks: the slowest rank's figures
            import torch.distributed as dist
            mm = torch.tensor([ms_graph, ms_eager], dtype=torch.float64, device=dev)
            dist.all_reduce(mm, op=dist.ReduceOp.MAX)
            ms_graph, ms_eager = float(mm[0]), float(mm[1])
        from unast_amd.engine import join_streams
        join_streams(); sync()
        auto_note = "auto: graph replay %.2f ms/step vs eager %.2f ms/step in 6 untimed steps each" % (ms_graph, ms_eager)
        if ms_eager < ms_graph:
            stepper = None
    for i in range(a.warmup):
        one_step(n_prime + i)
    sync()
    # In-region HIP-event pairs exist only in the eager form (a replayed graph has no per-kernel host hooks) and only with --time-every:
    # they cost ~15 us of host time per launch, enough to make the host the bottleneck.  The roofline figures come from the isolated
    # single-stream steps after the timed region.
    timed = ["gemm", "wgrad_group", "panel_gemm", "linear_dgrad_lnbwd", "attn_fwd", "attn_bwd"]
    if a.profile_ops:
        timed += ["layernorm_fwd", "layernorm_bwd", "colsum", "bn_fwd", "bn_bwd", "embed_fwd", "embed_bwd", "posenc_fwd", "posenc_bwd", "rowmask",
                  "add_inplace", "add_strided", "specaugment", "disc_gather", "disc_scatter", "speech_loss_fwd", "speech_loss_bwd", "text_loss_fwd",
                  "text_loss_bwd", "bce_logits", "disc_targets", "lstm_fwd", "lstm_bwd", "leaky_dropout", "sumsq", "adamw", "scale_inplace"]
    in_region = (stepper is None) and (a.profile_ops or a.time_every > 0)
    every = 1 if a.profile_ops else max(a.time_every, 1)
    with OpTimer(ops, timed) as ot:
        ot.active = False
        t0 = time.perf_counter()
        for i in range(a.steps):
            ot.active = in_region and (i % every == 0)
            one_step(n_prime + a.warmup + i)
        t_host = time.perf_counter() - t0          # host-side enqueue time (kernels run asynchronously)
        sync()
        dt = time.perf_counter() - t0
    n_timed = len(range(0, a.steps, every)) if in_region else 0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / a.steps * 1e3
    frames = B * Tm * world
    value = frames / (dt / a.steps)
    if stepper is not None:
        stepper.flush(losses)                      # the last step's discriminator phase (every timed call ran one D phase and one generator phase)
    sync()
    last = {k: float(v[-1]) for k, v in losses.items()}
    finite = all(v == v and abs(v) < 1e30 for v in last.values())
    # Isolated launch durations: extra steps AFTER the timed region, launched kernel by kernel with everything on ONE stream and a
    # HIP-event pair around every GEMM / attention launch on that stream (every rank runs them: the step contains the collectives).
    # These are what `roofline` reports and what the rocprofv3 summaries in profiles/ show (average duration per kernel).
    side = config.SIDE_STREAMS
    config.SIDE_STREAMS = False
    try:
        with OpTimer(ops, ["gemm", "wgrad_group", "panel_gemm", "linear_dgrad_lnbwd", "attn_fwd", "attn_bwd"]) as ot_iso:
            for i in range(a.iso_steps):
                train.train_step(losses, model, opt, sched, batches, n_prime + a.warmup + a.steps + i, args)
            sync()
        iso = ot_iso.summary()
    finally:
        config.SIDE_STREAMS = side
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank != 0:
        return
    summ = ot.summary()

    def family(su, n):
        calls = sum(v[0] for v in su[n].values())
        tot = sum(v[1] for v in su[n].values())
        fl = sum((gemm_flops(k) if n == "gemm" else attn_flops(n, k)) * v[0] for k, v in su[n].items())
        by = sum(gemm_bytes(k) * v[0] for k, v in su[n].items()) if n == "gemm" else 0.0
        if n == "gemm" and "wgrad_group" in su:      # the grouped weight-gradient launches belong to the same family (same kernel body)
            gsu = su["wgrad_group"]
            calls += sum(v[0] for v in gsu.values()); tot += sum(v[1] for v in gsu.values())
            fl += sum(group_flops(k) * v[0] for k, v in gsu.items()); by += sum(group_bytes(k) * v[0] for k, v in gsu.items())
        if n == "gemm" and "panel_gemm" in su:       # ... and so do the row-panel launches (the same contractions on the other kernel)
            psu = su["panel_gemm"]
            calls += sum(v[0] for v in psu.values()); tot += sum(v[1] for v in psu.values())
            fl += sum(panel_flops(k) * v[0] for k, v in psu.items()); by += sum(panel_bytes(k) * v[0] for k, v in psu.items())
        if n == "gemm" and "linear_dgrad_lnbwd" in su:       # ... and the input-gradient launches that carry a LayerNorm backward
            lsu = su["linear_dgrad_lnbwd"]
            calls += sum(v[0] for v in lsu.values()); tot += sum(v[1] for v in lsu.values())
            fl += sum(panel_flops(k) * v[0] for k, v in lsu.items()); by += sum(panel_bytes(k) * v[0] for k, v in lsu.items())
        return dict(calls=calls, ms=tot, flops=fl, bytes=by)
    global ATTN_BWD_PRODUCTS, ATTN_BWD_KERNEL
    if not config.ATTN_FUSED_BWD:
        ATTN_BWD_PRODUCTS, ATTN_BWD_KERNEL = 7, "attn_q_kernel<%(n)d,1> + attn_dkv_kernel<%(n)d,0> + attn_delta_kernel"
    fam = {n: family(iso, n) for n in ("gemm", "attn_fwd", "attn_bwd")}
    steps_iso = max(a.iso_steps, 1)

    def mfma_entry(n, kernel):
        d = fam[n]
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        return {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_MFMA_BF16_TFLOPS, 4), "traffic": None, "launches_per_step": d["calls"] / steps_iso,
                "avg_launch_us": round(d["ms"] * 1e3 / max(d["calls"], 1), 2), "ms_per_step": round(d["ms"] / steps_iso, 3),
                "mfma_issue_frac_of_peak": round(tf * (config.NSPLIT if n == "attn_fwd" else config.NSPLIT * ATTN_BWD_PRODUCTS / 5.0) / PEAK_MFMA_BF16_TFLOPS, 4)}
    g = fam["gemm"]
    gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0
    gtf = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    traffic = None
    for tp in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        tp = os.path.join(ROOT, "profiles", tp)
        if a.workload == "c3" and config.NSPLIT == 3 and os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                gk = [v for k, v in tj["kernels"].items() if "gemm_kernel" in k or "gemm_group" in k or "panel_kernel" in k]
                traffic = round(sum(v["hbm_MB_per_launch"] * v["launches"] for v in gk) / sum(v["launches"] for v in gk) * 1e6, 0)
                break
            except Exception:
                traffic = None
    rocprof_avg = None           # the same family's average launch duration in the committed single-stream rocprofv3 summary
    for cp in ("r04_b_bench_c3_single_stream_kernel_stats.csv", "r03_b_bench_c3_single_stream_kernel_stats.csv", "r02_f_bench_c3_single_stream_kernel_stats.csv"):
        cp = os.path.join(ROOT, "profiles", cp)
        if a.workload == "c3" and config.NSPLIT == 3 and os.path.exists(cp):
            try:
                import csv
                rows = [r for r in csv.DictReader(open(cp)) if r["Name"].startswith("void gemm_kernel") or "gemm_group_kernel" in r["Name"] or "panel_kernel" in r["Name"]]
                rocprof_avg = round(sum(int(r["TotalDurationNs"]) for r in rows) / sum(int(r["Calls"]) for r in rows) / 1e3, 2)
                break
            except Exception:
                rocprof_avg = None
    roofline = {"kernel": "gemm_kernel<*,*,%d> + panel_kernel<*> (all linear / conv contractions: forward, dgrad, grouped wgrad; the K <= 256 ones over >= 16 384 rows on the row-panel kernel)" % config.NSPLIT,
                "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": round(g["bytes"] / max(g["calls"], 1), 0),
                "launches_per_step": g["calls"] / steps_iso, "avg_launch_us": round(g["ms"] * 1e3 / max(g["calls"], 1), 2),
                "ms_per_step": round(g["ms"] / steps_iso, 3), "measured_over": "%d single-stream eager step(s) after the timed region" % a.iso_steps,
                "rocprof_avg_launch_us": rocprof_avg, "profiles_match_csrc": profiles_fresh(),
                "mfma_view": {"achieved_tflops": round(gtf, 2), "frac_of_2500_dense_bf16": round(gtf / PEAK_MFMA_BF16_TFLOPS, 4),
                              "mfma_issue_tflops": round(gtf * config.NSPLIT, 1), "sustained_mfma_peak_measured_tflops": SUSTAINED_MFMA_TFLOPS},
                "attn_fwd": mfma_entry("attn_fwd", "attn_fwd32_kernel<%d> (32x32x16 MFMA tiles)" % config.NSPLIT),
                "attn_bwd": mfma_entry("attn_bwd", ATTN_BWD_KERNEL % {"n": config.NSPLIT}),
                "note": "dominant family = GEMM.  achieved = algorithmic bytes (fp32 A + B + C and epilogue operands, each once; conv inputs once, not "
                        "per tap) / HIP-event time of these launches, taken on the launch stream in the isolated single-stream steps after the "
                        "timed region (inside the timed region a launch shares the chip with the kernels of the other three streams, and a replayed "
                        "capture has no per-kernel host hooks); traffic = PMC FETCH_SIZE(x2 on gfx950)+WRITE_SIZE per launch "
                        "from profiles/ (separate rocprofv3 passes of this command), null if absent; rocprof_avg_launch_us = the family's average kernel "
                        "duration in the committed single-stream rocprofv3 summary (profiles/r04_b_* or the newest older one; profiles_match_csrc says whether csrc/ still is what that summary was taken from): avg_launch_us brackets each launch with a "
                        "HIP-event pair and so carries ~4 us of dispatch per launch on top of it; mfma_view / attn_*: 2MNK FLOPs per "
                        "contraction, 4*B*H*Tq*Tk*64 per attention forward (x2.5 backward, causal at T(T+1)/2); each product costs %d bf16 MFMAs in "
                        "%s mode; sustained MFMA peak = tools/mfma_peak.cpp on this chip" % (config.NSPLIT, a.precision)}
    if in_region and n_timed:
        fr = {n: family(summ, n) for n in ("gemm", "attn_fwd", "attn_bwd")}
        roofline["overlapped_streams"] = {"families_ms_per_step": {n: round(fr[n]["ms"] / n_timed, 3) for n in fr}, "timed_steps": n_timed,
                                          "gemm_avg_launch_us": round(fr["gemm"]["ms"] * 1e3 / max(fr["gemm"]["calls"], 1), 2),
                                          "note": "event pairs inside the timed region (eager form only): durations include time shared with other streams"}
    out = {"metric": "mel-frames/sec/node (train step, gen+disc) at B=32,T_mel=800; 1/2/4/8-GPU scaling",
           "value": round(value, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16x3" if config.NSPLIT == 3 else "bf16", "data": "synthetic",
           "dist_backend": (a.backend if dist_on else None),
           "gradient_exchange": (None if not dist_on else ("RCCL through the C ABI (unast_comm_*), issued by the stream-replay executor / ddp.py" if native else "torch.distributed all_reduce")),
           "launch_mode": (("hip-graph replay of the captured step (unast_amd.graphed), %d untimed priming calls" % n_prime) if stepper is not None
                           else "eager (one Python launch per kernel)" + ("; gradient buckets all-reduced during the backward (unast_amd.ddp)" if dist_on else ""))
                          + ((" [" + auto_note + "]") if auto_note else ""),
           "config": {"workload": "%s: %s train step (AE+SP+clip/AdamW%s), per-GPU B=%d, T_text=%d, T_mel=%d, "
                                  "num_layers=%d, d=256, 4 heads, FFN 1024%s, dropout/noise/SpecAugment active%s" % (
                                      a.workload, "full adversarial gen+disc" if use_disc else "generator-only", ", D step+clip/AdamW" if use_disc else "", B, Tt, Tm, L,
                                      ", 2x bi-LSTM(64) discriminator" if use_disc else "",
                                      (" + %d cross-model sub-step(s) with K/V-cached generation (NOT the headline configuration)" % a.cm_steps) if a.cm_steps else ""),
                      "global_batch": B * world, "parallelism": "dp%d" % world,
                      "precision": "split-bf16 (hi/lo) MFMA operands, fp32 accumulate and fp32 activations" if config.NSPLIT == 3 else "bf16 MFMA operands, fp32 accumulate"},
           "graph_replay": (dict(next(iter(stepper.graphs.values())).plan_info,
                                 **{k: v for k, v in stepper.cache_report()["per_capture"][0].items() if k in ("capture_ms", "replays", "replay_host_ms", "eager_host_ms")})
                            if (stepper is not None and stepper.graphs) else None),
           "host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 3),
           "losses_finite": finite, "last_losses": {k: round(v, 5) for k, v in last.items()},
           "roofline": roofline}
    if a.workload in STEP_TFLOP and not a.cm_steps:
        tf = STEP_TFLOP[a.workload] * world / (ms * 1e-3)
        out["whole_step"] = {"algorithmic_tflop_per_step_per_gpu": STEP_TFLOP[a.workload], "achieved_tflops": round(tf, 1),
                             "frac_of_2500_dense_bf16_per_gpu": round(tf / world / PEAK_MFMA_BF16_TFLOPS, 4),
                             "note": "SURVEY.md section 8(d) table: multiply-add = 2, backward = 2x forward, causal self-attention at T(T+1)/2"}
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(B, Tt, Tm, L, use_disc, a.cpu_budget)
        except Exception as e:  # the checker must never take the bench down
            out["cpu_baseline"] = {"error": repr(e)}
    if a.profile_ops and n_timed:
        prof = {n: round(sum(v[1] for v in summ[n].values()) / n_timed, 3) for n in summ}
        sys.stderr.write("per-op ms/step: " + json.dumps(dict(sorted(prof.items(), key=lambda kv: -kv[1]))) + "\n")
        for n in ("gemm", "attn_fwd", "attn_bwd"):
            rows = sorted(summ[n].items(), key=lambda kv: -kv[1][1])[:12]
            for k, v in rows:
                fl = (gemm_flops(k) if n == "gemm" else attn_flops(n, k))
                sys.stderr.write("  %-9s %-28s calls/step %5.1f  avg %8.1f us  %7.1f TF/s\n" % (n, k, v[0] / n_timed, v[1] * 1e3 / v[0], fl * v[0] / (v[1] * 1e-3) / 1e12))
    if a.iso_detail:
        for n in ("gemm", "wgrad_group", "attn_fwd", "attn_bwd"):
            rows = sorted(iso[n].items(), key=lambda kv: -kv[1][1])[:16]
            for k, v in rows:
                fl = (gemm_flops(k) if n == "gemm" else group_flops(k) if n == "wgrad_group" else attn_flops(n, k))
                sys.stderr.write("  iso %-9s %-44s calls/step %5.1f  avg %8.1f us  total %7.3f ms/step  %7.1f TF/s\n" % (
                    n, k, v[0] / steps_iso, v[1] * 1e3 / v[0], v[1] / steps_iso, fl * v[0] / (v[1] * 1e-3) / 1e12))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
