"""Data-parallel gradient exchange, overlapped with the backward pass (new vs. the single-device reference; SURVEY.md
section 8e, north_star: "RCCL all-reduce of grads over xGMI overlapped with the discriminator backward").

One process per GPU, parameters replicated, per-GPU batch fixed.  The flat gradient buffer of the generator (engine.FlatStore)
is cut into four contiguous BUCKETS in the order the backward pass finishes them:

    text decoder + text postnet | speech decoder + speech postnet/heads | speech prenet + encoder | text prenet + encoder

Gradients accumulate over the generator sub-steps of one outer step (ae / cm / sp, /root/reference/src/train.py:608-628), so a
bucket is final only in the LAST of them.  train_step arms the exchange before that sub-step; when the backward of a public
model call (`decode_sequence`, `encode`) has been enqueued and it was the last user of its bucket, the bucket's all-reduce is
issued asynchronously behind the streams that produced it (the side stream of that call and the weight-gradient companion
streams; RCCL runs it on the process group's own stream) -- so the two decoder buckets (55 % of the bytes) travel while the
backward continues through the frozen LSTM discriminator and the encoders.  The optimizer step waits for those collectives,
reduces whatever was not pre-issued (the encoder buckets' tail, the discriminator phase's 0.28 M floats, any call pattern the hooks did not see), and
only then computes the global norm: the reference's order (generator update before the D-phase forward,
/root/reference/src/train.py:628-637) is kept.

Every rank runs the same Python and the same autograd graph, so buckets become ready -- and collectives are issued -- in the
same order on all ranks.
"""
import os

import torch

from . import ops

BUCKETS = ("text_dec", "speech_dec", "speech_enc", "text_enc")
_PREFIXES = {
    "text_enc": ("text_m.prenet.", "text_m.pos_emb.", "text_m.encoder."),
    "text_dec": ("text_m.decoder.", "text_m.postnet."),
    "speech_enc": ("speech_m.prenet.", "speech_m.pos_emb.", "speech_m.encoder."),
    "speech_dec": ("speech_m.decoder.", "speech_m.postnet."),
}
# Exercise the collective path at world size 1 too (tests on the one-GPU box; RCCL executes with a single rank).
FORCE = os.environ.get("UNAST_DDP_FORCE", "0") == "1"
# 0 = one blocking all-reduce per active range inside the optimizer step (round-1 behaviour; for A/B timing).
OVERLAP = os.environ.get("UNAST_DDP_OVERLAP", "1") != "0"


class _State:
    armed = False
    fwd_count = {}          # bucket -> forward segments recorded (with a tape) since arm()
    bwd_count = {}
    issued = []             # [(a, b)] ranges whose all-reduce has been issued (this optimizer phase)
    pending = []            # work handles of the pre-issued collectives
    log = []                # (bucket or "rest", a, b) in issue order -- read by tests
    scale_fn = None         # test hook: CPU tensors have no HIP scale kernel


class _Native:
    handle = 0              # unast_comm_init communicator of this process (0: none)
    tried = False
    issued = 0              # collectives issued through it (tests)
    last = None             # event behind the newest collective: the next one waits for it (see _issue)


def native_comm():
    """The RCCL communicator behind the C ABI for the current process group (nccl backend, CUDA tensors), created on first use: rank 0
    draws the unique id (unast_comm_unique_id), the launcher's process group broadcasts it, every rank calls unast_comm_init, and one
    small all-reduce checks the result against the group's size.  0 when disabled (config.NATIVE_COMM), not distributed, another
    backend, or when the check fails -- the exchange then stays on torch.distributed."""
    if _Native.tried:
        return _Native.handle
    dist = _dist()
    if dist is None:
        return 0
    _Native.tried = True
    from . import config
    import os
    shim = bool(os.environ.get("UNAST_COMM_LIB"))       # tests: csrc/comm.cpp binds another library with RCCL's entry points (tests/native/fake_rccl.cpp)
    if not config.NATIVE_COMM or not torch.cuda.is_available() or (dist.get_backend() != "nccl" and not shim):
        return 0
    import ctypes
    from ._lib import lib
    L = lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    ids = [None]
    if rank == 0:
        buf = (ctypes.c_char * 128)()
        if L.unast_comm_unique_id(ctypes.addressof(buf)) != 0:
            buf = None
        ids[0] = bytes(buf) if buf is not None else b""
    dist.broadcast_object_list(ids, src=0)
    if len(ids[0]) != 128:
        return 0                                         # (rank 0's verdict, broadcast: every rank returns here together)
    # From here on a rank that fails locally must STILL take part in the agreement below: a rank that returned early would leave the
    # others blocked in the probe collective / the MIN all-reduce for good.  Two rounds, both over the launcher's process group:
    # (1) did every rank get a communicator?  Only then (2) the probe collective on it, and the verdict on its result.
    b = ctypes.create_string_buffer(ids[0], 128)
    h = L.unast_comm_init(ctypes.addressof(b), rank, world)
    flag = torch.tensor([1.0 if h else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if float(flag) < 1.0:
        if h:
            L.unast_comm_destroy(h)
        return 0
    probe = torch.ones(256, dtype=torch.float32, device="cuda")
    ok = L.unast_allreduce(h, probe.data_ptr(), probe.numel(), ops._stream()) == 0
    torch.cuda.synchronize()
    ok = ok and bool((probe == float(world)).all())
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)              # every rank takes the same decision
    if float(flag) < 1.0:
        L.unast_comm_destroy(h)
        return 0
    _Native.handle = h
    return h


def _dist():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() == 1 and not FORCE:
        return None
    return dist


def active():
    return _dist() is not None


def bucket_ranges(store):
    """{bucket: (a, b)} over the flat buffer: each bucket runs from its first parameter to the next bucket's first parameter
    (alignment gaps included; they hold zeros), and together they tile the generator region exactly."""
    starts = {}
    for n, off in store.offsets.items():
        for bk, pres in _PREFIXES.items():
            if n.startswith(pres):
                starts[bk] = min(starts.get(bk, off), off)
    ga, gb = store.regions["gen"]
    order = sorted(starts.items(), key=lambda kv: kv[1])
    out = {}
    for i, (bk, a) in enumerate(order):
        b = order[i + 1][1] if i + 1 < len(order) else gb
        out[bk] = (a, b)
    if order:
        assert order[0][1] == ga, "generator region does not start with a bucket"
        for n, off in store.offsets.items():        # every generator parameter lies inside the bucket its name selects
            for bk, pres in _PREFIXES.items():
                if n.startswith(pres):
                    a, b = out[bk]
                    assert a <= off and off + store.params[n].numel() <= b, "parameter %s is outside bucket %s" % (n, bk)
    return out


def _issue_stream():
    """The stream a pre-issued bucket's all-reduce is enqueued behind: the weight-gradient companion of the current side stream
    when there is one (after it has joined the current stream), else the current stream itself.  No stream of its own: with
    the four of the train step's schedule plus RCCL's internal one the hardware queues are already shared, and a fifth user
    stream cost 2.6 ms/step on the one-GPU box (36.1 vs 33.5 ms, forced single-rank nccl)."""
    from . import engine
    cur = torch.cuda.current_stream()
    w = engine._wgrad_stream_of_current() if engine._Streams.enabled or engine._Streams.used else None
    if w is not None and w != cur:
        engine.wait(w, cur)
        return w
    return cur


def arm():
    """Called by train_step right before the LAST generator sub-step of an outer step: from here on a bucket whose users have
    all run their backward is final and may travel."""
    if not (OVERLAP and active()):
        return
    _State.armed = True
    _State.fwd_count, _State.bwd_count = {}, {}


def disarm():
    _State.armed = False


def _buckets_of(bucket):
    """A public call names the bucket its own parameters form; the decoders ALSO accumulate into parameters that live in their
    modality's encoder bucket (speech_decode runs speech_m.prenet.*, text_decode the embedding table text_m.prenet.embed), so
    they count as users of that bucket too: it is final only when the encoder's AND the decoder's backward have been enqueued,
    whatever order autograd runs them in (a cross-model sub-step runs an encoder's backward before its modality's decoder's)."""
    if bucket is None:
        return ()
    if isinstance(bucket, (tuple, list)):
        return tuple(bucket)
    return (bucket,) + _ALSO_WRITES.get(bucket, ())


_ALSO_WRITES = {"text_dec": ("text_enc",), "speech_dec": ("speech_enc",)}


def segment_forward(bucket):
    if _State.armed:
        for bk in _buckets_of(bucket):
            _State.fwd_count[bk] = _State.fwd_count.get(bk, 0) + 1


def segment_backward(bucket, store):
    """The backward of one public call that owns `bucket` has been enqueued on the current stream (its weight gradients on the
    companion streams)."""
    if not _State.armed or bucket is None:
        return
    for bk in _buckets_of(bucket):
        _State.bwd_count[bk] = _State.bwd_count.get(bk, 0) + 1
        if _State.bwd_count[bk] < _State.fwd_count.get(bk, 0):
            continue
        rng = bucket_ranges(store).get(bk)
        if rng is None or "gen" not in store.touched or any(r == rng for r in _State.issued):
            continue
        _issue(store, bk, rng, overlap=True)


def _issue(store, label, rng, overlap):
    dist = _dist()
    a, b = rng
    buf = store.grad[a:b]
    h = native_comm() if buf.is_cuda else 0
    if h:
        # RCCL through the C ABI: stream-ordered behind `s`; while a HIP graph is being captured a marker node stands in for the call and
        # the stream-replay executor issues the collective there (csrc/graph_exec.cpp)
        from ._lib import lib, check
        from . import engine
        cur = torch.cuda.current_stream()
        s = _issue_stream() if overlap else cur
        for name in list(engine._Streams.used):          # weight / LayerNorm-parameter gradients live on the companion streams
            if name.endswith("_w"):
                c = engine._side(name)
                if c != s:
                    engine.wait(s, c)
        with torch.cuda.stream(s):
            # One communicator, one collective at a time: the buckets are issued from several streams (the companions of the text and of
            # the speech side, the discriminator's), and two collectives of one RCCL communicator that run concurrently -- or in a
            # different order on another rank -- deadlock or corrupt its channels.  Each one therefore waits for the event behind the
            # previous one; the host-side issue order is the same on every rank.  Under capture only markers are left (no events: extra
            # side-to-side waits and events that die inside a capture are what takes hipStreamEndCapture down on ROCm 7.2, engine._ViaOrigin);
            # the stream-replay executor chains the collectives of a plan itself (csrc/graph_exec.cpp).
            if torch.cuda.is_current_stream_capturing():
                check(lib().unast_allreduce_marker(buf.data_ptr(), buf.numel(), ops._stream()), "unast_allreduce_marker")
                if s != cur:
                    _State.pending.append(s.record_event())
            else:
                if _Native.last is not None:
                    s.wait_event(_Native.last)
                check(lib().unast_allreduce(h, buf.data_ptr(), buf.numel(), ops._stream()), "unast_allreduce")
                _Native.last = s.record_event()
                if s != cur:
                    _State.pending.append(_Native.last)
        _Native.issued += 1
        _State.issued.append(rng)
        _State.log.append((label, a, b))
        return
    if buf.is_cuda and overlap:
        from . import engine
        s = _issue_stream()
        for name in list(engine._Streams.used):          # weight / LayerNorm-parameter gradients live on the companion streams
            if name.endswith("_w"):
                c = engine._side(name)
                if c != s:
                    engine.wait(s, c)
        with torch.cuda.stream(s):
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)   # RCCL: runs behind `s` on the group's own stream
        _State.pending.append(work)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    _State.issued.append(rng)
    _State.log.append((label, a, b))


def _subtract(rng, done):
    """Parts of [a,b) not covered by the ranges in `done`."""
    parts = [rng]
    for (c, d) in done:
        nxt = []
        for (a, b) in parts:
            if d <= a or c >= b:
                nxt.append((a, b))
            else:
                if a < c:
                    nxt.append((a, c))
                if d < b:
                    nxt.append((d, b))
        parts = nxt
    return [p for p in parts if p[1] > p[0]]


def finish(store, ranges):
    """Optimizer-step side: the current stream waits for the pre-issued buckets, reduces the rest of the active gradient
    ranges and scales everything by 1/world.  Returns the number of collectives issued here (0 when not distributed)."""
    disarm()
    if not active():
        _State.issued, _State.pending = [], []
        return 0
    dist = _dist()
    n = 0
    for rng in ranges:
        for part in _subtract(rng, _State.issued):
            _issue(store, "rest", part, overlap=False)
            n += 1
    for work in _State.pending:
        if isinstance(work, torch.cuda.Event):
            torch.cuda.current_stream().wait_event(work)   # native path: the collective is stream-ordered before this event
        else:
            work.wait()                                    # the current stream waits for that collective (no host block with RCCL)
    scale = _State.scale_fn or ops.scale_inplace
    ws = dist.get_world_size()
    for a, b in ranges:
        scale(store.grad[a:b], 1.0 / ws)
    _State.issued, _State.pending = [], []
    return n
