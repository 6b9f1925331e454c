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

WORKLOADS = {
    # name: (B, Tt, Tm, L, use_discriminator)
    "c3": (32, 180, 800, 4, True),      # BASELINE.json configs[2]: the configuration the metric is quoted on
    "c2": (8, 128, 512, 3, False),      # configs[1]: generator-only
    "c5": (32, 300, 2000, 4, True),     # configs[4]: long-form
    "tiny": (2, 24, 64, 2, True),
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


def attn_flops(name, key):
    B, H, Tq, Tk, causal = key
    pairs = Tq * (Tq + 1) / 2 if causal else Tq * Tk
    per = 4.0 * B * H * pairs * 64               # QK^T + PV
    return per if name == "attn_fwd" else per * 2.5      # backward: 5 products (S, dP, dV, dK, dQ)


def make_batch(B, Tt, Tm, seed):
    from unast_amd.portable import synth_batch
    return tuple(torch.from_numpy(x) for x in synth_batch(B, Tt, Tm, seed=seed, ragged=False))


def cpu_baseline_worker(Tt, Tm, L, use_disc, Bs):
    """Oracle (CPU restatement, pinned against the reference's golden vectors) timed on the host cores: one full
    gen+disc step (same step definition as the GPU run) on a bounded sample of the workload (Bs utterances)."""
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
    t0 = time.time()
    R.full_step(m, opt, batch, use_discriminator=use_disc)
    dt = time.time() - t0
    return {"value": round(Bs * Tm / dt, 2), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": "1 full gen+disc step of the same workload on B=%d utterance(s) (T_text=%d, T_mel=%d, L=%d), fp32 torch-CPU oracle "
                      "(dropout off), %.1f s" % (Bs, Tt, Tm, L, dt)}


def cpu_baseline(Tt, Tm, L, use_disc, budget_s):
    """Runs the worker in a child process with a hard time box so the default bench always finishes within minutes."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "print('CPUBASE ' + json.dumps(bench.cpu_baseline_worker(%d, %d, %d, %r, 16)))" % (ROOT, Tt, Tm, L, use_disc))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s, env=env)
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "mel-frames/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "B=16 step of the same workload did not finish within the %d s time box" % budget_s}
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
    ap.add_argument("--cm-steps", type=int, default=0, help="add this many cross-model (back-translation) sub-steps per step; reported "
                    "separately from the headline metric, which is defined with cm_steps = 0 (SURVEY.md section 8d)")
    ap.add_argument("--cm-max-len", type=int, default=0, help="cap of the autoregressive generation inside the cm sub-step (0 = reference "
                    "defaults 815 mel frames / 300 tokens)")
    ap.add_argument("--time-every", type=int, default=10, help="HIP-event pairs around the GEMM / attention launches on every N-th step of the timed region")
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

    def one_step(i):
        train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True)      # as train() does; the timed region ends with a device synchronise

    def sync():
        torch.cuda.synchronize(dev)
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        one_step(i)
    timed = ["gemm", "attn_fwd", "attn_bwd"]
    if a.profile_ops:
        timed += ["layernorm_fwd", "layernorm_bwd", "colsum", "bn_fwd", "bn_bwd", "embed_fwd", "embed_bwd", "posenc_fwd", "posenc_bwd", "rowmask",
                  "add_inplace", "add_strided", "specaugment", "disc_gather", "disc_scatter", "speech_loss_fwd", "speech_loss_bwd", "text_loss_fwd",
                  "text_loss_bwd", "bce_logits", "disc_targets", "lstm_fwd", "lstm_bwd", "leaky_dropout", "sumsq", "adamw", "scale_inplace"]
    sync()
    # HIP-event pairs around every GEMM / attention launch of a step cost ~15 ms of host time per step -- enough to make the host
    # the bottleneck (40 ms of enqueueing against 36 ms of GPU work) -- so they are on for every `every`-th step of the timed region.
    every = 1 if (a.steps <= 4 or a.profile_ops) else a.time_every
    n_timed = len(range(0, a.steps, every))
    with OpTimer(ops, timed) as ot:
        t0 = time.perf_counter()
        for i in range(a.steps):
            ot.active = (i % every == 0)
            one_step(a.warmup + i)
        t_host = time.perf_counter() - t0          # host-side enqueue time (kernels run asynchronously)
        sync()
        dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / a.steps * 1e3
    frames = B * Tm * world
    value = frames / (dt / a.steps)

    last = {k: float(v[-1]) for k, v in losses.items()}
    finite = all(v == v and abs(v) < 1e30 for v in last.values())
    # Isolated launch durations of the same kernels: two extra steps AFTER the timed region with everything on one stream (every
    # rank runs them: the step contains the collectives).  With the side streams on, a launch in the timed region shares the chip
    # with launches of other streams, so its duration says less about the kernel itself; the rocprofv3 summaries in profiles/ are
    # isolated durations too (under the profiler the host is the bottleneck and the streams hardly overlap).
    iso = None
    if config.SIDE_STREAMS:
        config.SIDE_STREAMS = False
        try:
            with OpTimer(ops, ["gemm"]) as ot_iso:
                for i in range(2):
                    one_step(a.warmup + a.steps + i)
                sync()
            iso = ot_iso.summary()["gemm"]
        finally:
            config.SIDE_STREAMS = True
    if rank != 0:
        return
    summ = ot.summary()
    # ---- dominant kernel family + roofline ------------------------------------------------------------------
    fam = {}
    for n in ("gemm", "attn_fwd", "attn_bwd"):
        calls = sum(v[0] for v in summ[n].values())
        tot = sum(v[1] for v in summ[n].values())
        fl = sum((gemm_flops(k) if n == "gemm" else attn_flops(n, k)) * v[0] for k, v in summ[n].items())
        fam[n] = dict(calls=calls, ms=tot, flops=fl, bytes=(sum(gemm_bytes(k) * v[0] for k, v in summ[n].items()) if n == "gemm" else 0.0))
    dom = max(fam, key=lambda n: fam[n]["ms"])
    d = fam[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    kernel_name = {"gemm": "gemm_kernel<*,*,%d> (all linear/conv contractions)" % config.NSPLIT,
                   "attn_fwd": "attn_q_kernel<%d,0>" % config.NSPLIT, "attn_bwd": "attn_q_kernel<%d,1> + attn_dkv_kernel<%d>" % (config.NSPLIT, config.NSPLIT)}[dom]
    if dom == "gemm":
        # The d=256 contractions are priced against HBM: with fp32 activations and 3 MFMAs per product they sit below the ridge
        # point, the MFMA pipe is 40 % busy (rocprofv3 PMC) and a load-only build of the kernel already takes 70 % of its time
        # (csrc/gemm.hip).  The MFMA view is kept beside it.
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        traffic = None
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.json")
        if a.workload == "c3" and config.NSPLIT == 3 and os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                g = [v for k, v in tj["kernels"].items() if "gemm_kernel" in k]
                traffic = round(sum(v["hbm_MB_per_launch"] * v["launches"] for v in g) / sum(v["launches"] for v in g) * 1e6, 0)
            except Exception:
                traffic = None
        roofline = {"kernel": kernel_name, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(d["bytes"] / max(d["calls"], 1), 0),
                    "launches_per_step": d["calls"] / n_timed, "timed_steps": n_timed, "avg_launch_us": round(d["ms"] * 1e3 / max(d["calls"], 1), 2),
                    "concurrent_streams": (4 if config.WGRAD_STREAMS else 3) if config.SIDE_STREAMS else 1,
                    "isolated": (None if not iso else (lambda c, ms, by: {"avg_launch_us": round(ms * 1e3 / max(c, 1), 2), "achieved": round(by / (ms * 1e-3) / 1e9, 1),
                                                                          "frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})(
                        sum(v[0] for v in iso.values()), sum(v[1] for v in iso.values()), sum(gemm_bytes(k) * v[0] for k, v in iso.items()))),
                    "mfma_view": {"achieved_tflops": round(ach, 2), "frac_of_2500_dense_bf16": round(ach / PEAK_MFMA_BF16_TFLOPS, 4),
                                  "mfma_issue_tflops": round(ach * config.NSPLIT, 1), "sustained_mfma_peak_measured_tflops": SUSTAINED_MFMA_TFLOPS},
                    "note": "achieved = algorithmic bytes (fp32 A + B + C and epilogue operands, each once; conv inputs once, not per tap) / HIP-event "
                            "time of these launches inside the timed region (event pairs on every --time-every-th step when steps > 4, see timed_steps; text side, speech side and discriminator run on three HIP streams and the "
                            "speech side's weight gradients on a fourth, so a launch's duration includes time it shares the chip with kernels of the "
                            "others; `isolated` = the same launches in two extra single-stream steps after the timed region, comparable with profiles/); traffic = PMC FETCH_SIZE(x2 on gfx950)+WRITE_SIZE per launch from "
                            "profiles/r01_pmc_hbm_traffic.json (separate rocprofv3 passes of this command), null if that file is absent; mfma_view: 2MNK "
                            "FLOPs per contraction, each product costs %d bf16 MFMAs in %s mode; sustained peak = tools/mfma_peak.cpp on this chip" % (config.NSPLIT, a.precision),
                    "families_ms_per_step": {n: round(fam[n]["ms"] / n_timed, 3) for n in fam}}
    else:
        roofline = {"kernel": kernel_name, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_MFMA_BF16_TFLOPS, 4), "traffic": None,
                    "launches_per_step": d["calls"] / n_timed, "timed_steps": n_timed, "avg_launch_us": round(d["ms"] * 1e3 / max(d["calls"], 1), 2),
                    "mfma_issue_frac": round(ach * config.NSPLIT / PEAK_MFMA_BF16_TFLOPS, 4),
                    "note": "achieved = algorithmic FLOPs (4*B*H*Tq*Tk*64 per attention forward, x2.5 backward) / HIP-event time of these launches "
                            "inside the timed region; each product costs %d bf16 MFMAs in %s mode" % (config.NSPLIT, a.precision),
                    "families_ms_per_step": {n: round(fam[n]["ms"] / n_timed, 3) for n in fam}}
    out = {"metric": "mel-frames/sec/node (train step, gen+disc) at B=32,T_mel=800; 1/2/4/8-GPU scaling",
           "value": round(value, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16x3" if config.NSPLIT == 3 else "bf16", "data": "synthetic",
           "dist_backend": (a.backend if dist_on else None),
           "config": {"workload": "%s: full adversarial gen+disc train step (AE+SP+clip/AdamW, D step+clip/AdamW), per-GPU B=%d, T_text=%d, T_mel=%d, "
                                  "num_layers=%d, d=256, 4 heads, FFN 1024, 2x bi-LSTM(64) discriminator, dropout/noise/SpecAugment active%s" % (
                                      a.workload, B, Tt, Tm, L, (" + %d cross-model sub-step(s) with K/V-cached generation (NOT the headline configuration)" % a.cm_steps) if a.cm_steps else ""),
                      "global_batch": B * world, "parallelism": "dp%d" % world,
                      "precision": "split-bf16 (hi/lo) MFMA operands, fp32 accumulate and fp32 activations" if config.NSPLIT == 3 else "bf16 MFMA operands, fp32 accumulate"},
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
            out["cpu_baseline"] = cpu_baseline(Tt, Tm, L, use_disc, 150)
        except Exception as e:  # the checker must never take the bench down
            out["cpu_baseline"] = {"error": repr(e)}
    if a.profile_ops:
        prof = {n: round(sum(v[1] for v in summ[n].values()) / n_timed, 3) for n in summ}
        sys.stderr.write("per-op ms/step: " + json.dumps(dict(sorted(prof.items(), key=lambda kv: -kv[1]))) + "\n")
        for n in ("gemm", "attn_fwd", "attn_bwd"):
            rows = sorted(summ[n].items(), key=lambda kv: -kv[1][1])[:12]
            for k, v in rows:
                fl = (gemm_flops(k) if n == "gemm" else attn_flops(n, k))
                sys.stderr.write("  %-9s %-28s calls/step %5.1f  avg %8.1f us  %7.1f TF/s\n" % (n, k, v[0] / n_timed, v[1] * 1e3 / v[0], fl * v[0] / (v[1] * 1e-3) / 1e12))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
