"""Host-side execution engine: flat parameter/gradient storage, a minimal backward tape, and the bridge that makes
each public model call one node of torch's autograd graph.

Design (MI355X-first, not a port of the reference's module-by-module autograd):
  * all parameters of a model live in ONE flat fp32 HBM buffer (and one flat gradient buffer of the same layout), so
    clip+AdamW is two kernel launches and the data-parallel gradient exchange is a single bucket per phase;
  * conv weights are stored tap-major [Cout,5,Cin] (the implicit-GEMM K order); the nn.Parameter the user sees is a
    permuted view with the reference's [Cout,Cin,5] shape, so state_dict round-trips unchanged;
  * weight-gradient kernels ACCUMULATE into the flat gradient buffer (split-K atomics / +=), the buffer is zeroed
    once per optimizer phase, and `p.grad` are views into it;
  * forward code records closures on a Tape; one torch.autograd.Function per public call (`encode`, `decode_sequence`,
    discriminator forward, each loss) runs the tape in reverse, so `loss.backward()` works as in the reference
    while every arithmetic kernel is ours.
"""
import torch

from . import ops

ALIGN = 64  # floats (256 B) between parameter starts


# ---------------------------------------------------------------------------------------------------------------
# Side streams: text side / speech side / discriminator as three HIP streams inside one train step
# ---------------------------------------------------------------------------------------------------------------
class _Streams:
    enabled = False      # inside `with side_streams():`
    pool = {}            # (device index, name) -> torch.cuda.Stream
    producer = {}        # storage address -> (stream name, event recorded after the producing call); views share the storage
    used = set()
    trace = None         # list of (stream name, call, start event, end event) when tracing


def _stream_group(name):
    """Logical stream name -> the real stream it runs on (several logical streams may share one)."""
    from . import config
    return config.STREAM_GROUPS.get(name, name)


def _side(name):
    key = (torch.cuda.current_device(), _stream_group(name))
    s = _Streams.pool.get(key)
    if s is None:
        from . import config
        s = _Streams.pool[key] = torch.cuda.Stream(priority=config.STREAM_PRIORITY.get(key[1], 0))
    return s


def _tensors(obj, out):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            out.append(obj)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _tensors(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors(o, out)
    return out


class side_streams:
    """Context of one train sub-step: while it is active, calls decorated with `on_stream(name)` run on their own HIP stream
    and only wait for the streams that produced their inputs.  Leaving it joins every side stream into the current one."""

    def __enter__(self):
        from . import config
        self.active = config.SIDE_STREAMS and not _Streams.enabled and torch.cuda.is_available()
        if self.active:
            _Streams.enabled = True
            _Streams.producer.clear()
            if config.WGRAD_STREAMS:
                ops.WGRAD_SIDE = _wgrad_stream_of_current
        return self

    def __exit__(self, *exc):
        if self.active:
            ops.WGRAD_SIDE = None
            if not getattr(self, "leave_open", False):
                join_streams()
            _Streams.enabled = False
            _Streams.producer.clear()


def _wgrad_stream_of_current():
    """Companion stream "<name>_w" of the side stream that is current (None on any other stream)."""
    cur = torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    from . import config
    for (d, name), s in _Streams.pool.items():
        if d == dev and s == cur and name in config.WGRAD_COMPANION_OF:
            _Streams.used.add(name + "_w")
            return _side(name + "_w")
    return None


def join_wgrad_streams():
    """The current stream waits for the weight-gradient companion streams (before an optimizer step that runs on a side
    stream itself)."""
    cur = torch.cuda.current_stream()
    for name in list(_Streams.used):
        if name.endswith("_w"):
            wait(cur, _side(name))


def stream_of(name):
    """The side stream `name` as a context manager target, or None when side streams are off (callers then stay on the
    current stream)."""
    from . import config
    if not (config.SIDE_STREAMS and torch.cuda.is_available()):
        return None
    _Streams.used.add(name)
    return _side(name)


def join_streams():
    """The current stream waits for everything enqueued on the side streams and their weight-gradient companions (before
    torch ops that read their results, at sub-step boundaries, before the optimizer).  Host-side cost only: no device
    synchronisation.  (Leaving the companions out of the sub-step joins was measured: no gain, 39.0 vs 39.2 ms/step.)"""
    if not _Streams.used:
        return
    cur = torch.cuda.current_stream()
    for name in _Streams.used:
        s = _side(name)
        if s != cur:
            wait(cur, s)
    if not _Streams.enabled:
        _Streams.used.clear()


class _CaptureWaits:
    """Waits between streams placed by this package while a HIP graph is being captured.  ROCm 7.2's hipStreamEndCapture dies (a
    segmentation fault inside the runtime, tools/debug_capture.py T1 / T6 / T10) when two streams that are both forked from the capture's
    origin wait on EACH OTHER -- A waits for B after B has waited for A; one-way waits (T9) and waits through the origin (T3) are fine.
    The cause inside the runtime is not known; the package therefore never builds that topology (on_stream hands tensors side -> origin
    -> side, _ViaOrigin makes autograd do the same, ddp._issue leaves marker nodes without events), and every explicit wait goes through
    `wait()` below, which REFUSES the second half of such a pair with a Python error instead of letting the process die in EndCapture."""
    origin = None        # the capture's origin stream (inference._capture)
    edges = set()        # (waiter stream, source stream) pairs of the open capture


def capture_begins(origin):
    _CaptureWaits.origin = origin
    _CaptureWaits.edges = set()
    # every library call from here on tells the library which stream its launch went to, so that the stream replay of the captured
    # graph can keep this module's stream structure (include/unast_hip.h: unast_capture_note)
    from . import _lib
    l = _lib.lib()
    l.unast_capture_reset()
    _GradTail.clear()
    note, cur = l.unast_capture_note, ops._stream
    _lib.CAPTURE_NOTE = lambda: note(cur())


def capture_ends():
    _CaptureWaits.origin = None
    _CaptureWaits.edges = set()
    from . import _lib
    _lib.CAPTURE_NOTE = None


def wait(waiter, source_stream, event=None):
    """`waiter` waits for `event` (recorded on `source_stream`) or, without one, for everything enqueued on `source_stream`."""
    o = _CaptureWaits.origin
    if o is not None and waiter != o and source_stream != o and waiter != source_stream:
        a, b = waiter.cuda_stream, source_stream.cuda_stream
        if (b, a) in _CaptureWaits.edges:
            raise RuntimeError("inside a HIP-graph capture two side streams may not wait on each other (stream %#x already waited for %#x): "
                               "hipStreamEndCapture of ROCm 7.2 crashes on that topology -- hand over through the capture's origin stream" % (b, a))
        _CaptureWaits.edges.add((a, b))
    if event is not None:
        waiter.wait_event(event)
    else:
        waiter.wait_stream(source_stream)


import os as _os
_FORCE_VIA_ORIGIN = _os.environ.get("UNAST_VIA_ORIGIN", "0") == "1"      # experiment: the capture-time hand-off discipline in eager mode


class _ViaOrigin(torch.autograd.Function):
    """Identity placed -- on the CALLER's stream -- on an edge between two side streams while a HIP graph is being captured.
    ROCm 7.2's hipStreamEndCapture crashes when a side stream waits on an event of another side stream that has itself waited
    on the first one (A -> B -> A; tools/debug_capture.py T1/T6 crash, T3 "via origin" and T9 "one way" do not), and the train
    step has such pairs: text encoder -> speech decoder in the forward, speech decoder -> text encoder in the backward.  With
    this node in between, torch's autograd hands the gradient side -> origin -> side, and on_stream does the same for the
    forward value, so side streams only ever wait on the origin stream."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g


PRUNED = [0]            # inherited dependencies dropped by _Segment._backward while capturing (tests look at it)
_GradTail = {}          # while capturing: storage address of a gradient -> the graph nodes that produced it (tail of its stream then)


def _prune_on():
    from . import config
    return config.CAPTURE_PRUNE


def _capture_deps(stream):
    import ctypes
    from . import _lib
    buf = (ctypes.c_void_p * 64)()
    n = _lib.lib().unast_capture_get_deps(stream.cuda_stream, buf, 64)
    return None if n < 0 else tuple(buf[i] for i in range(n))


def _prune(stream, origin, keep):
    import ctypes
    from . import _lib
    buf = (ctypes.c_void_p * max(len(keep), 1))(*keep)
    n = _lib.lib().unast_capture_prune(stream.cuda_stream, origin.cuda_stream, buf, len(keep))
    if n < 0:
        raise RuntimeError("unast_capture_prune failed: %s" % _lib.lib().unast_last_error().decode())
    PRUNED[0] += n


def on_stream(name):
    """Decorator: inside `side_streams()` the call runs on the side stream `name` after waiting for (a) everything enqueued on
    the caller's stream so far and (b) the side streams that produced any tensor argument (tracked by storage, so views and
    permutes keep their producer).  Cross-stream arguments are handed to the caching allocator with record_stream."""
    def deco(fn):
        def wrapper(*args, **kw):
            if not _Streams.enabled:
                return fn(*args, **kw)
            cur = torch.cuda.current_stream()
            s = _side(name)
            if cur == s:
                return fn(*args, **kw)
            via_origin = torch.cuda.is_current_stream_capturing() or _FORCE_VIA_ORIGIN
            waited = set()
            inputs = set()
            cross = set()
            for t in _tensors((args, kw), []):
                inputs.add(t.untyped_storage().data_ptr())
                src = _Streams.producer.get(t.untyped_storage().data_ptr())
                if src is not None and src[0] != name:
                    cross.add(id(t))
                    if id(src[1]) not in waited:
                        # only the call that produced this input, not its whole stream; while capturing, through the caller's stream
                        wait(cur if via_origin else s, _side(src[0]), src[1])
                        waited.add(id(src[1]))
                t.record_stream(s)
            if via_origin and cross:
                def route(o):
                    if isinstance(o, torch.Tensor) and id(o) in cross and o.requires_grad and torch.is_grad_enabled():
                        r = _ViaOrigin.apply(o)
                        r.record_stream(s)
                        return r
                    return o
                args = tuple(route(a) for a in args)
                kw = {k: route(v) for k, v in kw.items()}
            wait(s, cur)
            with torch.cuda.stream(s):
                ops.jitter()                            # (test infrastructure, off by default: config.STREAM_JITTER)
                if _Streams.trace is not None:          # diagnostic only (tools/stream_timeline.py): event pair around the call
                    import time
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(); h0 = time.perf_counter()
                out = fn(*args, **kw)
                if _Streams.trace is not None:
                    h1 = time.perf_counter()
                    e1 = torch.cuda.Event(enable_timing=True); e1.record()
                    _Streams.trace.append((name, "%s [host enqueue %.2f ms]" % (getattr(fn, "__qualname__", str(fn)), (h1 - h0) * 1e3), e0, e1))
                ev = s.record_event()
            for t in _tensors(out, []):
                ptr = t.untyped_storage().data_ptr()
                if ptr not in inputs:                   # an argument handed back (decode_sequence returns tgt_lens) is not a product:
                    _Streams.producer[ptr] = (name, ev)  # re-tagging it made later users of the batch's lengths wait for this call
            _Streams.used.add(name)
            return out
        wrapper.__name__ = getattr(fn, "__name__", "wrapped")
        wrapper.__doc__ = fn.__doc__
        return wrapper
    return deco


class Var:
    """An activation and its gradient slot.  ln / pre (functional.py, the post-LN stacks): the LayerNorm that produced `v` and whose
    only consumer is the next sub-layer -- that sub-layer's backward may then run the LayerNorm's backward in the epilogue of its last
    GEMM and leave the result in `pre` = (dz, dz_dropped) instead of a gradient in `g`."""
    __slots__ = ("v", "g", "ln", "pre")

    def __init__(self, v, g=None):
        self.v = v
        self.g = g
        self.ln = None
        self.pre = None


def acc(var, grad):
    """Accumulate `grad` into var.g (first writer owns the buffer)."""
    if grad is None:
        return
    if var.g is None:
        var.g = grad
    else:
        ops.add_inplace(var.g, grad)


class Tape:
    def __init__(self):
        self.fns = []

    def record(self, fn):
        self.fns.append(fn)

    def backward(self):
        for fn in reversed(self.fns):
            fn()
        self.fns = []


class _Segment(torch.autograd.Function):
    """One public model call = one autograd node; inputs/outputs are torch tensors, the inside is kernels + Tape."""

    @staticmethod
    def forward(ctx, run, hook, *inputs):
        tape = Tape()
        in_vars = [Var(t) if isinstance(t, torch.Tensor) and t.is_floating_point() else t for t in inputs]
        outs = run(tape, *in_vars)
        ctx.tape, ctx.in_vars, ctx.out_vars = tape, in_vars, outs
        ctx.hook = hook
        if hook is not None:
            hook[0](hook[1])                           # ddp.segment_forward(bucket)
        ctx.label = getattr(run, "__qualname__", "segment").split(".<locals>")[0]
        res = tuple(o.v for o in outs)
        return res if len(res) > 1 else res[0]

    @staticmethod
    def backward(ctx, *gouts):
        ops.jitter()                                    # (test infrastructure, off by default: config.STREAM_JITTER)
        if _Streams.trace is not None:                  # diagnostic only (tools/stream_timeline.py)
            import time
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            h0 = time.perf_counter()
            try:
                return _Segment._backward(ctx, *gouts)
            finally:
                h1 = time.perf_counter()
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                cur = torch.cuda.current_stream()
                nm = next((n for (d, n), st in _Streams.pool.items() if st == cur), "main")
                _Streams.trace.append((nm, "bwd %s [host enqueue %.2f ms]" % (ctx.label, (h1 - h0) * 1e3), e0, e1))
        return _Segment._backward(ctx, *gouts)

    @staticmethod
    def _backward(ctx, *gouts):
        cur = torch.cuda.current_stream() if _Streams.used else None
        origin = _CaptureWaits.origin
        prune = origin is not None and cur is not None and cur != origin and _prune_on()
        if prune:
            # Inside a capture this stream has just synced with the origin for its incoming gradients and inherited the producers of every
            # hand-off the origin ever relayed; keep those of ITS gradients only (include/unast_hip.h unast_capture_prune).
            keep = []
            for g in gouts:
                if g is not None:
                    keep += _GradTail.get(g.untyped_storage().data_ptr(), ())
            _prune(cur, origin, keep)
        for o, g in zip(ctx.out_vars, gouts):
            if g is not None and g.dtype != torch.float32:
                g = g.float()
            if g is not None and cur is not None:
                g.record_stream(cur)                  # may have been produced (and be freed) on another stream
            o.g = g
        ctx.tape.backward()
        if ctx.hook is not None:
            ctx.hook[2](ctx.hook[1], ctx.hook[3])      # ddp.segment_backward(bucket, store): this call's gradients are enqueued
        grads = tuple((v.g if isinstance(v, Var) else None) for v in ctx.in_vars)
        if origin is not None and cur is not None and _prune_on():
            tail = _capture_deps(cur)                  # what produced these gradients, for the segment that receives them
            if tail:
                for gr in grads:
                    if gr is not None:                  # (a union: two producers may fill row blocks of one buffer, and the allocator reuses addresses)
                        ptr = gr.untyped_storage().data_ptr()
                        old = _GradTail.get(ptr)
                        _GradTail[ptr] = tail if not old else tuple(set(old) | set(tail))
        ctx.tape = ctx.in_vars = ctx.out_vars = None
        return (None, None) + grads


def ddp_hook(bucket, store):
    """`hook` argument of run_segment for a public call whose parameters form the gradient bucket `bucket` (unast_amd.ddp)."""
    from . import ddp
    return (ddp.segment_forward, bucket, ddp.segment_backward, store)


def run_segment(run, hook, *inputs):
    """Executes `run(tape, *vars) -> [Var,...]`.  With grad mode off no tape is kept.  `hook` (see ddp_hook) tells the
    data-parallel layer when this call's backward has been enqueued."""
    if not torch.is_grad_enabled():
        in_vars = [Var(t) if isinstance(t, torch.Tensor) and t.is_floating_point() else t for t in inputs]
        outs = run(None, *in_vars)
        res = tuple(o.v for o in outs)
        return res if len(res) > 1 else res[0]
    return _Segment.apply(run, hook, *inputs)


# ---------------------------------------------------------------------------------------------------------------
# Flat parameter store
# ---------------------------------------------------------------------------------------------------------------
def _region_of(name):
    if name.startswith("discriminator."):
        return "disc_unused" if ".reduce_c_W." in name else "disc"
    return "gen"


def _layout_order(names):
    """Physical order: generator params, then discriminator (with the adjacency the kernels rely on), then the
    never-used reduce_c_W.  Adjacent pairs form single GEMM operands:
      [linear_project.weight | stop_linear.weight] -> [81,256]; LSTM (fwd | reverse) weights and biases per layer."""
    gen = [n for n in names if _region_of(n) == "gen"]
    disc = [n for n in names if _region_of(n) == "disc"]
    unused = [n for n in names if _region_of(n) == "disc_unused"]

    def move_after(lst, first, second):
        if first in lst and second in lst:
            lst.remove(second)
            lst.insert(lst.index(first) + 1, second)
    move_after(gen, "speech_m.postnet.linear_project.weight", "speech_m.postnet.stop_linear.weight")
    move_after(gen, "speech_m.postnet.linear_project.bias", "speech_m.postnet.stop_linear.bias")
    for n in list(disc):
        if n.endswith("_reverse"):
            move_after(disc, n[: -len("_reverse")], n)
    return gen, disc, unused


class FlatStore:
    """Owns the flat parameter / gradient buffers of one module tree and re-points its nn.Parameters into them."""

    def __init__(self, module, prefix=""):
        """`prefix` maps a stand-alone sub-model onto canonical names (e.g. "text_m." for a bare TextTransformer)."""
        named = [(prefix + n, p) for n, p in module.named_parameters()]
        if not named:
            raise ValueError("module has no parameters")
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("unast_amd runs on the GPU only: move the model to a CUDA (ROCm) device first; there is no CPU path")
        self.device = dev
        self.params = dict(named)
        gen, disc, unused = _layout_order([n for n, _ in named])
        self.offsets, self.regions = {}, {}
        off = 0
        packed_after = {"speech_m.postnet.linear_project.weight", "speech_m.postnet.linear_project.bias"}
        for rname, lst in (("gen", gen), ("disc", disc), ("disc_unused", unused)):
            start = off
            prev = None
            for n in lst:
                tight = prev is not None and (prev in packed_after or (n.endswith("_reverse") and prev == n[: -len("_reverse")]))
                if not tight:
                    off = (off + ALIGN - 1) // ALIGN * ALIGN
                self.offsets[n] = off
                off += self.params[n].numel()
                prev = n
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.regions[rname] = (start, off)
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.phys, self.gphys = {}, {}
        for n, p in self.params.items():
            o, k = self.offsets[n], p.numel()
            if n.endswith(".conv.weight"):        # tap-major physical layout [Cout,5,Cin]
                co, ci, ks = p.shape
                ph = self.flat[o:o + k].view(co, ks, ci)
                ph.copy_(p.data.permute(0, 2, 1))
                p.data = ph.permute(0, 2, 1)
                self.gphys[n] = self.grad[o:o + k].view(co, ks, ci)
            else:
                ph = self.flat[o:o + k].view(p.shape)
                ph.copy_(p.data)
                p.data = ph
                self.gphys[n] = self.grad[o:o + k].view(p.shape)
            self.phys[n] = ph
        # the weights once more in the GEMM's pre-split operand format (same byte offsets; refreshed by the AdamW kernel)
        self.flat_split = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self._split_ver = None
        self._build_transposed()
        self._build_planes()
        ops.register_weight_span(self.flat.data_ptr(), self.total * 4, self.flat_split.data_ptr(), self.dgrad_T, self.planes_of)
        self.dummy = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)   # forces autograd to call our backward
        self.touched = set()           # regions that received gradients since the last zero_grad
        self.zeroed_by_step = set()    # regions whose gradients the optimizer's update pass has already zeroed (FusedAdamW.step(zero_grads=True))
        self.grads_exposed = False

    def __del__(self):
        try:
            ops.unregister_weight_span(self.flat.data_ptr())
        except Exception:
            pass

    # transposed pre-split weights (B operand of the input-gradient GEMMs) -------------------------------------------
    def _build_transposed(self):
        """Every 2-D weight W [rows, K] (and the combined operands that are used as one matrix) gets a slot for W^T in the
        pre-split format, [K][ceil4(rows)]; `refresh_T` fills the slots of a region with one launch (csrc/optim.hip
        transpose_split_kernel) after every optimizer step and whenever sync_split() re-splits the weights."""
        from . import config
        self._tmats, self._tcache, self.flat_T, self._ttiles = [], {}, None, {}
        if not config.DGRAD_TRANSPOSED:
            return
        cands = []
        for n, p in self.params.items():
            if p.dim() == 2 and not n.endswith(".conv.weight"):
                cands.append((self.offsets[n], p.shape[0], p.shape[1], _region_of(n)))
        P = self.params
        if "speech_m.postnet.linear_project.weight" in P:                      # [linear_project | stop_linear] -> one [81, 256] head
            a, b = P["speech_m.postnet.linear_project.weight"], P["speech_m.postnet.stop_linear.weight"]
            cands.append((self.offsets["speech_m.postnet.linear_project.weight"], a.shape[0] + b.shape[0], a.shape[1], "gen"))
        for n, p in P.items():                                                    # LSTM [weight_ih | weight_ih_reverse] -> [2 * 4H, Din]
            if n.startswith("discriminator.rnn.rnn.weight_ih_l") and not n.endswith("_reverse") and (n + "_reverse") in P:
                cands.append((self.offsets[n], 2 * p.shape[0], p.shape[1], "disc"))
        t_off = 0
        for off, rows, K, region in cands:
            if K % 4 or off % 4:
                continue
            ldT = (rows + 3) // 4 * 4
            self._tmats.append((off, rows, K, t_off, ldT, region))
            t_off += (K * ldT + ALIGN - 1) // ALIGN * ALIGN
        self.flat_T = torch.zeros(max(t_off, 4), dtype=torch.float32, device=self.device)
        for region in ("gen", "disc", "disc_unused"):
            tiles = []
            for off, rows, K, to, ldT, rg in self._tmats:
                if rg != region:
                    continue
                for r0 in range(0, ldT, 64):
                    for c0 in range(0, K, 64):
                        tiles += [off, to, rows, K, r0, c0]
            if tiles:
                self._ttiles[region] = (torch.tensor(tiles, dtype=torch.int32, device=self.device), len(tiles) // 6)

    def refresh_T(self, regions=None):
        for region, (tiles, n) in self._ttiles.items():
            if regions is None or region in regions:
                ops.transpose_split(self.flat, self.flat_T, tiles, n)

    def dgrad_T(self, W):
        """(W^T view pointer, its row stride) for a weight view W [N, K] inside the flat store, or None."""
        if self.flat_T is None or W.dim() != 2 or W.stride(1) != 1:
            return None
        key = (W.data_ptr(), W.shape[0], W.shape[1])
        hit = self._tcache.get(key, 0)
        if hit != 0:
            return hit
        res = None
        pos = (W.data_ptr() - self.flat.data_ptr()) // 4
        N, K = W.shape
        if W.stride(0) == K:
            for off, rows, Km, to, ldT, rg in self._tmats:
                if Km == K and off <= pos < off + rows * K and (pos - off) % K == 0:
                    row0 = (pos - off) // K
                    if row0 % 4 == 0 and row0 + N <= rows and (row0 + N == rows or N % 4 == 0):
                        # the slice's last chunk may only run into the zero padding, never into a neighbour's rows
                        res = (self.flat_T.data_ptr() + 4 * (to + row0), ldT)
                        if row0 + N == rows:
                            break
        self._tcache[key] = res
        return res

    # tiled bf16 planes (B operand of the row-panel GEMM, csrc/panel.hip) ------------------------------------------------
    def _build_planes(self):
        """Every 2-D weight W [rows, K] that planes.eligible accepts gets tiled hi / lo planes of itself (forward operand) and
        of W^T (the operand of its input-gradient GEMM dX = dY W); combined operands (the [81, 256] head, the LSTM's [2 * 4H, Din]
        input projections) and the q rows of the cross-attention in-projections likewise.  `refresh_planes` re-tiles a region with
        one launch (unast_retile_weights) after every optimizer step and whenever sync_split() sees parameters written through torch."""
        from . import config, planes
        self._pmats, self._pcache, self.planes_buf, self._pdescs = [], {}, None, {}
        if not config.PANEL_GEMM:
            return
        cands = []                                   # (flat offset, source row stride, rows, cols, region)
        for n, p in self.params.items():
            if p.dim() == 2 and not n.endswith(".conv.weight"):
                cands.append((self.offsets[n], p.shape[1], p.shape[0], p.shape[1], _region_of(n)))
                if n.endswith("multihead_attn.in_proj_weight") and p.shape[0] == 3 * p.shape[1]:      # q rows and k/v rows: their own W^T
                    cands.append((self.offsets[n], p.shape[1], p.shape[1], p.shape[1], _region_of(n)))
                    cands.append((self.offsets[n] + p.shape[1] * p.shape[1], p.shape[1], 2 * p.shape[1], p.shape[1], _region_of(n)))
        P = self.params
        if "speech_m.postnet.linear_project.weight" in P:
            a, b = P["speech_m.postnet.linear_project.weight"], P["speech_m.postnet.stop_linear.weight"]
            cands.append((self.offsets["speech_m.postnet.linear_project.weight"], a.shape[1], a.shape[0] + b.shape[0], a.shape[1], "gen"))
        for n, p in P.items():
            if n.startswith("discriminator.rnn.rnn.weight_ih_l") and not n.endswith("_reverse") and (n + "_reverse") in P:
                cands.append((self.offsets[n], p.shape[1], 2 * p.shape[0], p.shape[1], "disc"))
        mats, seen = [], set()
        for off, ld, rows, cols, region in cands:
            if planes.eligible(rows, cols) and (off, rows, cols, 0) not in seen:        # Wd[n = rows][k = cols]
                seen.add((off, rows, cols, 0)); mats.append((off, ld, 0, rows, cols, region))
            if planes.eligible(cols, rows) and (off, rows, cols, 1) not in seen:        # Wd[n = cols][k = rows] = W^T
                seen.add((off, rows, cols, 1)); mats.append((off, ld, 1, cols, rows, region))
        if not mats:
            return
        descs, placed, total = planes.plan([m[:5] for m in mats])
        self.planes_buf = torch.zeros(total, dtype=torch.uint8, device=self.device)
        self._pmats = [(m[0], m[1], m[2], m[3], m[4], m[5]) + pl for m, pl in zip(mats, placed)]      # + (dst_off, plane_bytes, ksteps)
        import numpy as np
        regions = np.repeat([m[5] for m in mats], [((m[3] + 63) // 64) * ((((m[4] + 31) // 32) * 32 + 63) // 64) for m in mats])
        for region in ("gen", "disc", "disc_unused"):
            sel = descs[regions == region]
            if len(sel):
                self._pdescs[region] = (planes.descs_to_device(sel, self.device), len(sel))

    def refresh_planes(self, regions=None):
        for region, (descs, n) in self._pdescs.items():
            if regions is None or region in regions:
                ops.retile_weights(self.flat, self.planes_buf, descs, n)

    def planes_of(self, W, transposed=False):
        """(hi-plane pointer, plane bytes) of the tiled planes of the stored weight view W [N, K] (transposed: of W^T), or None."""
        if self.planes_buf is None or W.dim() != 2 or W.stride(1) != 1:
            return None
        key = (W.data_ptr(), W.shape[0], W.shape[1], W.stride(0), transposed)
        hit = self._pcache.get(key, 0)
        if hit != 0:
            return hit
        res = None
        pos = (W.data_ptr() - self.flat.data_ptr()) // 4
        rows, cols = W.shape
        for off, ld, tr, N, K, region, dst, pb, ksteps in self._pmats:
            if W.stride(0) != ld or tr != int(transposed):
                continue
            if not transposed:
                # rows [row0, row0 + rows) of a stored [N, K] operand: the planes are cut in 16-row tiles, a 64-row boundary keeps groups whole
                if cols == K and off <= pos < off + N * K and (pos - off) % K == 0:
                    row0 = (pos - off) // K
                    if row0 % 64 == 0 and row0 + rows <= N:
                        res = (self.planes_buf.data_ptr() + dst + (row0 // 16) * ksteps * 1024, pb)
                        break
            elif pos == off and cols == N and rows == K:
                res = (self.planes_buf.data_ptr() + dst, pb)
                break
        self._pcache[key] = res
        return res

    def sync_split(self):
        """Re-split the weights if any parameter was written through torch since the last look (load_state_dict, manual
        edits: they bump the parameter's version counter; the AdamW kernel updates both copies itself)."""
        ver = 0
        for p in self.params.values():
            ver += p._version
        if ver != self._split_ver:
            ops.split_f32(self.flat, self.flat_split)
            self.refresh_T()
            self.refresh_planes()
            self._split_ver = ver
            # The copies were remade on whatever stream the first public call after the write runs on (inside train_*_step: a side stream),
            # and the next call -- on ANOTHER side stream -- reads them a few microseconds later: the joint generator step, which issues
            # the text and the speech encoder back to back, read stale copies in one of three runs right after load_ckp
            # (tests/test_gpu_loop_ckpt.py).  Rare path (a torch-side write), so: the host waits here and every later launch is behind it.
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()

    # combined operands ---------------------------------------------------------------------------------------
    def span(self, first, last, shape, grad=False):
        """View over consecutive parameters first..last as one tensor of `shape`."""
        buf = self.grad if grad else self.flat
        o = self.offsets[first]
        n = 1
        for s in shape:
            n *= s
        assert o + n <= self.offsets[last] + self.params[last].numel() + 0 and self.offsets[last] + self.params[last].numel() == o + n, \
            "parameters %s..%s are not contiguous" % (first, last)
        return buf[o:o + n].view(shape)

    def g(self, name):
        """Gradient view for `name`, or None when the parameter is frozen (requires_grad=False)."""
        p = self.params[name]
        if not p.requires_grad:
            return None
        self.touched.add(_region_of(name))
        return self.gphys[name]

    def gspan(self, first, last, shape):
        if not self.params[first].requires_grad:
            return None
        self.touched.add(_region_of(first))
        return self.span(first, last, shape, grad=True)

    def expose_grads(self):
        """Make p.grad views of the flat gradient buffer for every parameter of a touched region (reference
        semantics: parameters that took no part in the backward keep grad None and are skipped by AdamW)."""
        join_streams()
        for n, p in self.params.items():
            if _region_of(n) in self.touched and p.requires_grad:
                gv = self.gphys[n]
                p.grad = gv.permute(0, 2, 1) if n.endswith(".conv.weight") else gv
            else:
                p.grad = None

    def zero_grad(self):
        ops.reset_wgrad_choices()
        for r in self.touched:
            if r in self.zeroed_by_step:
                continue
            a, b = self.regions[r]
            self.grad[a:b].zero_()
        self.touched = set()
        self.zeroed_by_step = set()
        for p in self.params.values():
            p.grad = None

    def active_ranges(self):
        return [self.regions[r] for r in ("gen", "disc") if r in self.touched]
