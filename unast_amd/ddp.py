"""Data-parallel gradient exchange, overlapped with the backward pass (new vs. the single-device reference; SURVEY.md
section 8e, north_star: "RCCL all-reduce of grads over xGMI overlapped with the discriminator backward").

One process per GPU, parameters replicated, per-GPU batch fixed.  The flat gradient buffer of the generator (engine.FlatStore)
is cut into four contiguous BUCKETS in the order the backward pass finishes them:

    text decoder + text postnet | speech decoder + speech postnet/heads | speech prenet + encoder | text prenet + encoder

Gradients accumulate over the generator sub-steps of one outer step (ae / cm / sp, /root/reference/src/train.py:608-628), so a
bucket is final only in the LAST of them.  train_step arms the exchange before that sub-step; when the backward of a public
model call (`decode_sequence`, `encode`) has been enqueued and it was the last user of its bucket, the bucket's all-reduce is
issued on a communication stream that waits for the producing streams (the side stream of that call and the weight-gradient
companion streams) -- so the two decoder buckets (55 % of the bytes) travel while the backward continues through the frozen
LSTM discriminator and the encoders.  The optimizer step waits for the communication stream, reduces whatever was not
pre-issued (the encoder buckets' tail, the discriminator phase's 0.28 M floats, any call pattern the hooks did not see), and
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
    issued = []             # [(a, b)] ranges whose all-reduce is already on the communication stream (this optimizer phase)
    comm = {}               # device index -> torch.cuda.Stream
    log = []                # (bucket or "rest", a, b) in issue order -- read by tests
    scale_fn = None         # test hook: CPU tensors have no HIP scale kernel


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


def _comm_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _State.comm.get(idx)
    if s is None:
        s = _State.comm[idx] = torch.cuda.Stream(device=device)
    return s


def arm():
    """Called by train_step right before the LAST generator sub-step of an outer step: from here on a bucket whose users have
    all run their backward is final and may travel."""
    if not (OVERLAP and active()):
        return
    _State.armed = True
    _State.fwd_count, _State.bwd_count = {}, {}


def disarm():
    _State.armed = False


def segment_forward(bucket):
    if _State.armed and bucket is not None:
        _State.fwd_count[bucket] = _State.fwd_count.get(bucket, 0) + 1


def segment_backward(bucket, store):
    """The backward of one public call that owns `bucket` has been enqueued on the current stream (its weight gradients on the
    companion streams)."""
    if not _State.armed or bucket is None:
        return
    _State.bwd_count[bucket] = _State.bwd_count.get(bucket, 0) + 1
    if _State.bwd_count[bucket] < _State.fwd_count.get(bucket, 0):
        return
    rng = bucket_ranges(store).get(bucket)
    if rng is None or "gen" not in store.touched or any(r == rng for r in _State.issued):
        return
    _issue(store, bucket, rng, overlap=True)


def _issue(store, label, rng, overlap):
    dist = _dist()
    a, b = rng
    buf = store.grad[a:b]
    ws = dist.get_world_size()
    scale = _State.scale_fn or ops.scale_inplace
    if buf.is_cuda and overlap:
        from . import engine
        comm = _comm_stream(buf.device)
        cur = torch.cuda.current_stream()
        comm.wait_stream(cur)
        for name in list(engine._Streams.used):          # weight / LayerNorm-parameter gradients live on the companion streams
            s = engine._side(name)
            if s != cur:
                comm.wait_stream(s)
        with torch.cuda.stream(comm):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)    # RCCL: enqueued behind `comm`, which then waits for its completion
            scale(buf, 1.0 / ws)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        scale(buf, 1.0 / ws)
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
    """Optimizer-step side: the current stream waits for the pre-issued buckets and reduces the rest of the active gradient
    ranges.  Returns the number of collectives issued here (0 when not distributed)."""
    disarm()
    if not active():
        _State.issued = []
        return 0
    n = 0
    for rng in ranges:
        for part in _subtract(rng, _State.issued):
            _issue(store, "rest", part, overlap=False)
            n += 1
    if store.grad.is_cuda:
        comm = _State.comm.get(store.grad.device.index if store.grad.device.index is not None else torch.cuda.current_device())
        if comm is not None:
            torch.cuda.current_stream().wait_stream(comm)
    _State.issued = []
    return n
