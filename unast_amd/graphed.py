"""The train step as HIP-graph replays.

One outer step of the reference's hot loop (/root/reference/src/train.py:602-655) is ~2 500 kernel launches; issued one by one
from Python the host needs 18-30 ms per step, which bounds small configurations outright (BASELINE config 2) and leaves the
large one a slow host away from it.  `GraphedTrainStep` captures the step once per input shape -- four HIP streams with their
fork / join events included -- and replays it; per replay the host does three things: copies the batch into the static input
buffers, writes 16 words of step state (RNG epoch, learning rate, Adam bias corrections: ops.set_step_state) and launches the
graph.

What makes the step capturable:
  * dropout / noise / SpecAugment / permutation masks are functions of (seed, stream, row, col, EPOCH) with the epoch read from
    device memory (csrc/common.h rng_epoch), so a replay draws fresh masks while forward and backward of one replay agree;
  * clip + AdamW read lr and the bias corrections from device memory (unast_adamw dev_hyper);
  * the discriminator's row permutation comes from the same counter RNG (unast_randperm) instead of torch.randperm;
  * losses are device scalars, gathered into one vector inside the graph and snapshotted after the replay (no host sync).

Schedule: as train_step(defer_d_phase=True) -- the discriminator phase of step k-1 (its encoders' forward, the LSTM forward /
backward, its clip + AdamW) shares the chip with the generator forward of step k.  In eager mode that overlap spans two calls;
a graph cannot leave work behind, so the captured unit is the SHIFTED step  [ D phase of step k-1  ||  generator phase of
step k ]  with the same dependencies (D update before the generator's own D call, all on the discriminator's stream; generator
update after everything).  The first call runs its generator phase eagerly, `flush()` runs the last pending D phase.

Variable input shapes: the captured unit reads the discriminator batch of step k-1 and the generator batches of step k, so a capture
is keyed by BOTH shapes (and the mode flags); static input buffers are kept per shape and shared by the captures that read them.
Captures live in an LRU (MAX_GRAPHS entries, UNAST_MAX_GRAPHS) and allocate from ONE private memory pool -- replays never overlap
and leave nothing alive but their loss vector, so the pool holds the largest step's working set once, not once per capture.  A
(shape, shape) pair is run eagerly the first time it is met and captured the second time (a pair met once costs nothing extra);
`cache_report()` gives captures / replays / evictions and the measured capture cost against the host time a replay saves.
Shapes must be the batch's own padded shapes: padding batches up to a bucket changes the result (BatchNorm statistics and the stop
loss see the pad rows).

Not capturable, falls back to train.train_step: cm_steps > 0 (data-dependent autoregressive generation); under a torch.distributed
group only with the C ABI's own RCCL communicator (ddp.native_comm), else the eager step with unast_amd.ddp.
"""
import os
import time
from collections import OrderedDict, defaultdict

import torch

from . import config, ddp, ops
from . import train as T
from .engine import join_streams
from .utils import is_deterministic

MAX_GRAPHS = int(os.environ.get("UNAST_MAX_GRAPHS", "32"))
MAX_WARMED = 4096            # (shape, shape) pairs remembered as "met once"


class _Captured:
    __slots__ = ("graph", "loss_keys", "loss_vec", "ranges", "plan", "plan_info", "capture_ms", "replays", "replay_host_ms")


# Replay of a captured step: "streams" = the captured nodes re-issued on ordinary HIP streams by csrc/graph_exec.cpp,
# "hipgraph" = hipGraphLaunch of the instantiated graph, "auto" = by the shape of the graph: hipGraphLaunch is the cheaper
# launch for a nearly linear graph (config 2, no discriminator: 16 cross-stream edges, 9.7 vs 10.5 ms/step) and the slower
# executor for a heavily forked one (config 3: 190 cross-stream edges, 35-39 vs 32 ms/step).  MI355X, ROCm 7.2.
REPLAY = os.environ.get("UNAST_GRAPH_REPLAY", "auto")
REPLAY_STREAMS = int(os.environ.get("UNAST_GRAPH_REPLAY_STREAMS", "4"))       # 32.2 / 34.1 / 35.1 / 35.6 ms/step with 4 / 5 / 6 / 8 at config 3
AUTO_MIN_CROSS_EDGES = 64


class GraphedTrainStep:
    """Drop-in for train.train_step(losses, model, optimizer, scheduler, batches, step, args, defer_d_phase=True)."""

    def __init__(self, model, optimizer, scheduler, args):
        if not isinstance(optimizer, T.FusedAdamW):
            raise TypeError("GraphedTrainStep needs the FusedAdamW built by train.initialize_model")
        self.model, self.opt, self.sched, self.args = model, optimizer, scheduler, args
        self.static = {}              # {"unsup": [...], "sup": [...], "disc": [...]} of 4-tuples of device tensors: what the next body reads
        self.static_gen = {}          # generator-side shapes -> {"unsup": [...], "sup": [...]}
        self.static_disc = {}         # discriminator batch shapes -> [...]
        self.gen_sig = self.disc_sig = None     # shapes of the loaded generator batches / of the batch of the PENDING discriminator phase
        self.flags = None
        self.graphs = OrderedDict()   # (disc shapes, generator shapes, flags) -> _Captured, least recently used first
        self.pool = None              # the captures' shared memory pool
        self.pending_lr = None        # learning rate of the D phase that has not run yet (None: nothing pending)
        self.warmed = OrderedDict()   # signatures whose shifted body has run eagerly once -> host ms of that run
        self.stats = dict(captures=0, replays=0, eager_bodies=0, evictions=0)
        self.epoch = 0
        ops.step_state()              # device block + RNG-epoch pointer exist before any capture

    # ---- plumbing ------------------------------------------------------------------------------------------------
    def capturable(self):
        # under a process group the step is capturable when the gradient exchange runs through the C ABI's own RCCL communicator: the
        # capture then holds marker nodes and the stream-replay executor issues the collectives (ddp.native_comm, csrc/comm.cpp)
        return getattr(self.args, "cm_steps", 0) == 0 and (not ddp.active() or bool(ddp.native_comm()))

    def _signatures(self, batches):
        a = self.args
        shp = lambda k, n: tuple((k, i, tuple(tuple(t.shape) for t in batches[k][i])) for i in range(n))
        gen = shp("unsup", a.ae_steps) + shp("sup", a.sp_steps)
        disc = shp("disc", a.d_steps) if a.use_discriminator else ()
        return gen, disc, (is_deterministic(), config.NSPLIT, self.model.training, config.JOINT_GEN, config.JOINT_DECODERS)

    @staticmethod
    def _static_like(batch, dev):
        """Static buffers of one (text, mel, text_len, mel_len) batch; the lengths as int32, the form every kernel takes them in: the copy
        into the buffer converts, and process_batch's own cast becomes a no-op inside the capture (6 launches per step)."""
        return tuple(torch.empty(t.shape, dtype=(torch.int32 if (i >= 2 and not t.dtype.is_floating_point) else t.dtype), device=dev) for i, t in enumerate(batch))

    def _copy_in(self, dsts, srcs):
        for dst, src in zip(dsts, srcs):
            for d, s in zip(dst, src):
                if not (s.is_cuda and s.data_ptr() == d.data_ptr()):
                    d.copy_(s, non_blocking=True)

    def _load_gen(self, batches, gen_sig):
        """Copies the generator batches into the static buffers of their shape (device-to-device or host-to-device on the current stream)."""
        dev = T._dev()
        st = self.static_gen.get(gen_sig)
        if st is None:
            a = self.args
            st = self.static_gen[gen_sig] = {k: [self._static_like(batches[k][i], dev) for i in range(n)] for k, n in (("unsup", a.ae_steps), ("sup", a.sp_steps))}
        self.gen_sig = gen_sig
        self.static["unsup"], self.static["sup"] = st["unsup"], st["sup"]
        for k in ("unsup", "sup"):
            self._copy_in(st[k], batches[k])

    def _load_disc(self, batches, disc_sig):
        dev = T._dev()
        st = self.static_disc.get(disc_sig)
        if st is None:
            n = self.args.d_steps if self.args.use_discriminator else 0
            st = self.static_disc[disc_sig] = [self._static_like(batches["disc"][i], dev) for i in range(n)]
        self.disc_sig = disc_sig
        self.static["disc"] = st
        self._copy_in(st, batches["disc"])

    def _gen_phase(self, losses):
        a, model = self.args, self.model
        if a.use_discriminator:
            T.freeze_model_parameters(model.discriminator)
        accum = a.ae_steps + a.sp_steps
        if a.ae_steps == 1 and a.sp_steps == 1 and T.joint_generator_phase(a, self.static["unsup"][0], self.static["sup"][0]):
            ddp.arm()
            T.train_gen_joint_step(losses, model, self.static["unsup"][0], self.static["sup"][0], 0, accum, a)
        else:
            subs = [(T.train_ae_step, b) for b in self.static["unsup"]] + [(T.train_sp_step, b) for b in self.static["sup"]]
            for i, (fn, b) in enumerate(subs):
                if i == len(subs) - 1:
                    ddp.arm()                      # last generator sub-step: gradient buckets travel during its backward (no-op when not distributed)
                fn(losses, model, b, 0, accum, a)
        T.optimizer_step(model, self.opt, a)

    def _d_phase(self, losses, defer):
        a, model = self.args, self.model
        T.unfreeze_model_parameters(model.discriminator)
        for b in self.static["disc"]:
            T.train_discriminator_step(losses, model, b, 0, a.d_steps, a, defer=defer)
        T.optimizer_step(model, self.opt, a, defer=defer)

    def _body(self, losses):
        """[ D phase of the previous step || generator phase of this step ], all streams joined at the end."""
        if self.args.use_discriminator:
            self._d_phase(losses, defer=True)
        self._gen_phase(losses)
        join_streams()

    # ---- the step ------------------------------------------------------------------------------------------------
    def __call__(self, losses, batches, step=0):
        a, model, opt = self.args, self.model, self.opt
        if not self.capturable():
            T.train_step(losses, model, opt, self.sched, batches, step, a, defer_d_phase=True)
            return
        if not model.training:
            model.train()
        model._store().sync_split()                       # parameters written through torch since the last step?
        gen_sig, disc_sig, flags = self._signatures(batches)
        if self.pending_lr is not None and flags != self.flags:
            self.flush(losses)                            # mode change (deterministic / eval): the pending D phase belongs to the old mode
        self.flags = flags
        lr_now = float(opt.param_groups[0]["lr"])
        if a.use_discriminator and self.pending_lr is None:
            # first step (or first after flush()): generator phase only, eagerly; its D phase runs inside the next call
            self._load_gen(batches, gen_sig)
            self._load_disc(batches, disc_sig)
            self._gen_phase(losses)
            join_streams()
        else:
            self._load_gen(batches, gen_sig)               # static["disc"] still holds the previous step's batch: the body reads it first
            sig = (self.disc_sig, gen_sig, flags)
            rec = self.graphs.get(sig)
            if rec is None and sig not in self.warmed:
                # once eagerly with exactly the body's call sequence (lazy initialisations, allocator warm-up, stream creation); a pair of
                # shapes that never comes back is never captured
                t0 = time.perf_counter()
                self._with_lrs(lambda: self._body(losses), lr_now)
                self.warmed[sig] = (time.perf_counter() - t0) * 1e3
                while len(self.warmed) > MAX_WARMED:
                    self.warmed.popitem(last=False)
                self.stats["eager_bodies"] += 1
            else:
                if rec is None:
                    rec = self._capture(sig)
                else:
                    self.graphs.move_to_end(sig)           # most recently used
                self._replay(rec, losses, lr_now)
            if a.use_discriminator:
                self._load_disc(batches, disc_sig)         # ordered behind the replay that read the previous one
        if a.use_discriminator:
            self.pending_lr = lr_now
        if self.sched is not None:
            self.sched.step()

    def _with_lrs(self, fn, lr_now):
        """Eager run of the shifted body: the D phase of the previous step uses the previous step's learning rate."""
        g = self.opt.param_groups[0]
        if not self.args.use_discriminator:
            return fn()
        # the body is D phase then generator phase; switch lr between them by wrapping the optimizer's step
        seen = {"n": 0}
        orig = self.opt.step

        def step(*aa, **kk):
            g["lr"] = self.pending_lr if seen["n"] == 0 else lr_now
            seen["n"] += 1
            return orig(*aa, **kk)
        self.opt.step = step
        try:
            return fn()
        finally:
            self.opt.step = orig
            g["lr"] = lr_now

    def _capture(self, sig):
        from .inference import _capture
        join_streams()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while len(self.graphs) >= max(MAX_GRAPHS, 1):     # evict the least recently used capture (nothing of it is in flight after the synchronize)
            self._drop(next(iter(self.graphs)))
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        rec = _Captured()
        rec.replays, rec.replay_host_ms = 0, 0.0
        cap_losses = defaultdict(list)
        self.opt.captured_ranges = []
        sync_flag, T.SYNC_LOSSES = T.SYNC_LOSSES, False

        def fn():
            self._body(cap_losses)
            flat = [(k, v) for k, vs in cap_losses.items() for v in vs]
            rec.loss_keys = [k for k, _ in flat]
            rec.loss_vec = [v.reshape(()) for _, v in flat] if flat else None       # scalars in the capture's memory: gathered after each replay
        try:
            rec.graph = _capture(fn, pool=self.pool, keep_graph=True)
        finally:
            T.SYNC_LOSSES = sync_flag
        rec.plan, rec.plan_info = 0, None
        distributed = ddp.active()
        if REPLAY in ("streams", "auto") or distributed:
            from ._lib import lib
            import ctypes
            rec.plan = lib().unast_graph_plan_create(rec.graph.raw_cuda_graph(), REPLAY_STREAMS)
            if rec.plan:
                info = (ctypes.c_int * 4)()
                lib().unast_graph_plan_info(rec.plan, ctypes.addressof(info))
                rec.plan_info = dict(mode="streams", kernels=info[0], memsets=info[1], memcpys=info[2], cross_stream_edges=info[3], streams=lib().unast_graph_plan_streams(rec.plan))
                nar = lib().unast_graph_plan_allreduces(rec.plan)
                rec.plan_info["allreduces"] = nar
                if nar > 0:
                    lib().unast_graph_plan_set_comm(rec.plan, ddp.native_comm())
                if REPLAY == "auto" and info[3] < AUTO_MIN_CROSS_EDGES and nar == 0 and not distributed:
                    lib().unast_graph_plan_destroy(rec.plan)
                    rec.plan = 0
                    rec.plan_info["mode"] = "hipGraphLaunch (nearly linear graph)"
            else:
                rec.plan_info = dict(mode="hipGraphLaunch (fallback: %s)" % lib().unast_last_error().decode())
        if not rec.plan:
            if distributed:
                raise RuntimeError("GraphedTrainStep: a distributed step needs the stream-replay executor (its gradient exchanges are marker "
                                   "nodes); plan creation failed: %s" % rec.plan_info)
            rec.graph.instantiate()
        rec.ranges = list(self.opt.captured_ranges)
        self.graphs[sig] = rec
        torch.cuda.synchronize()
        rec.capture_ms = (time.perf_counter() - t0) * 1e3
        self.stats["captures"] += 1
        return rec

    def _drop(self, sig):
        rec = self.graphs.pop(sig)
        if rec.plan:
            from ._lib import lib
            lib().unast_graph_plan_destroy(rec.plan)
            rec.plan = 0
        rec.graph = None
        self.stats["evictions"] += 1
        # static input buffers no capture reads any more (and that are not the loaded ones) go with it
        used_d = {k[0] for k in self.graphs} | {self.disc_sig}
        used_g = {k[1] for k in self.graphs} | {self.gen_sig}
        for k in [k for k in self.static_disc if k not in used_d]:
            del self.static_disc[k]
        for k in [k for k in self.static_gen if k not in used_g]:
            del self.static_gen[k]

    def cache_report(self):
        """Counters of the capture cache and, per cached capture, what it cost and what it saves: capture_ms (one-off), the host time of
        the eager body it replaced (eager_host_ms, from the run that preceded the capture) and of its replays (replay_host_ms, mean)."""
        per = []
        for sig, rec in self.graphs.items():
            per.append(dict(disc_shapes=sig[0], gen_shapes=sig[1], capture_ms=round(rec.capture_ms, 2), replays=rec.replays,
                            replay_host_ms=round(rec.replay_host_ms / max(rec.replays, 1), 3), eager_host_ms=round(self.warmed.get(sig, float("nan")), 3)))
        return dict(self.stats, cached=len(self.graphs), max_graphs=MAX_GRAPHS, shapes_gen=len(self.static_gen), shapes_disc=len(self.static_disc), per_capture=per)

    def _replay(self, rec, losses, lr_now):
        t0 = time.perf_counter()
        hyper = {}
        dr = self.model._store().regions.get("disc")
        for rng in rec.ranges:
            lr = self.pending_lr if (self.args.use_discriminator and rng == dr) else lr_now
            slot, vals = self.opt.replay_hyper(rng, lr)
            hyper[slot] = vals
        self.epoch += 1
        ops.set_step_state(self.epoch, hyper)
        if rec.plan:
            from ._lib import lib, check
            check(lib().unast_graph_plan_replay(rec.plan, ops._stream()), "unast_graph_plan_replay")
        else:
            rec.graph.replay()
        self.opt._step_count = getattr(self.opt, "_step_count", 0) + 1       # what torch's LR schedulers look at
        if rec.loss_vec is not None:
            snap = torch.stack(rec.loss_vec)              # one launch: gathers and snapshots (the scalars are overwritten by the next replay)
            for i, k in enumerate(rec.loss_keys):
                losses[k].append(snap[i])
        rec.replays += 1
        rec.replay_host_ms += (time.perf_counter() - t0) * 1e3
        self.stats["replays"] += 1

    def flush(self, losses=None):
        """Runs the discriminator phase that is still pending (eagerly, joined): before evaluation, checkpoints, or reading
        parameters."""
        if self.pending_lr is None:
            join_streams()
            return
        losses = losses if losses is not None else defaultdict(list)
        g = self.opt.param_groups[0]
        lr_now = g["lr"]
        g["lr"] = self.pending_lr
        try:
            self._d_phase(losses, defer=False)
        finally:
            g["lr"] = lr_now
        T.freeze_model_parameters(self.model.discriminator)
        join_streams()
        self.pending_lr = None
