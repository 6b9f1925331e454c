"""Autoregressive inference with a per-layer K/V cache.

The reference's infer_sequence (src/network.py:219-252 speech, 455-481 text) re-runs the whole decoder over the growing
prefix at every step (O(T^2) layer evaluations, `torch.cat` growth, one host sync per step for the stop test).  Here each
step pushes ONE new position through the decoder: the self-attention keys/values of earlier positions stay in an HBM
cache, the cross-attention K/V of the memory are projected once per layer, and the stop test is read back every
`SYNC_EVERY` steps (tokens generated after the true exit step are discarded, so results are identical).
Positions generated after a sequence has stopped are masked as padded keys exactly as the reference's `dec_mask` does:
they form a suffix, so the mask is a per-sequence valid length min(i+1, stop_len+1).

One decoded position is ~35 small launches on B rows each (csrc/decode.hip: contractions and single-query attention built for
latency; embedding, positional encoding and the LayerNorms are produced inside the contraction that consumes them, the K|V of
the new position go straight from the in-projection into the cache), i.e. bound by launch overhead.  The step therefore keeps
its position in DEVICE memory, so that it can be captured once as a HIP graph and replayed per position
(`config.DECODE_GRAPH`); dropout streams of replayed launches are varied by the device-side epoch counter of
`ops.rng_epoch_counter()`.  Two generations that do not depend on each other (the two directions of a cross-model step) are
decoded in lock-step from one graph with two branches (`run_pair`).

Semantics note: with dropout active (model.train() under no_grad, as in cm_text_in / cm_speech_in) the reference draws
fresh masks for every prefix position at every step; the cached form draws them once per position.  With the RNG sites
off (parity tests) and in eval mode the two are identical.
"""
import math

import torch

from . import config, ops
from .utils import SOS_IDX, EOS_IDX, PAD_IDX

SYNC_EVERY = 8


def _empty(*shape, dev):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class _LayerStep:
    """One decoder layer applied to a single new position per sequence, with cached self-attention K/V.  Eight launches of the
    latency-built kernels of csrc/decode.hip: the three LayerNorms are folded into the contraction that consumes them (norm3 into
    the NEXT layer's first contraction), the K|V of the new position go straight from the in-projection into the cache."""

    def __init__(self, cx, lp, mem2d, lens_mem, B, Tk, Tcap, H, drop):
        self.cx, self.lp, self.B, self.Tk, self.Tcap, self.H = cx, lp, B, Tk, Tcap, H
        self.p = cx.p(drop)
        self.lens_mem = lens_mem
        P = cx.P
        E = mem2d.shape[1]
        self.E = E
        Wc, bc = P[lp + "multihead_attn.in_proj_weight"], P[lp + "multihead_attn.in_proj_bias"]
        self.memkv = _empty(B * Tk, 2 * E, dev=mem2d.device)
        ops.linear_fwd(mem2d, Wc[E:], bc[E:], self.memkv)                      # projected once for all steps
        self.cache = torch.zeros(B, Tcap, 2 * E, dtype=torch.float32, device=mem2d.device)

    def norm_out(self):
        return (self.cx.P[self.lp + "norm3.weight"], self.cx.P[self.lp + "norm3.bias"])

    def __call__(self, x, pro, pos_t, stop_lens):
        """x: the layer input [B,E]; pro: how the first contraction produces its input rows from x (kwargs of
        ops.decode_linear: {} = x as it is, ln=(gamma, beta) of the previous layer's norm3 on its pre-norm sum, embed=/posenc= for
        the first layer).  Returns this layer's pre-norm3 sum; the consumer applies norm_out()."""
        cx, lp, B, E, H = self.cx, self.lp, self.B, self.E, self.H
        P, p, dev = cx.P, self.p, self.cache.device
        # --- self-attention over the cache (positions 0..pos; stopped sequences keep their frozen valid length)
        q = _empty(B, E, dev=dev)
        xn = _empty(B, E, dev=dev) if pro else None
        ops.decode_linear(x, P[lp + "self_attn.in_proj_weight"], P[lp + "self_attn.in_proj_bias"], q, xn_out=xn, seed=cx.seed,
                          cache=self.cache, split_col=E, pos=pos_t, **pro)       # q -> q, K|V -> cache[:, pos]
        if pro:
            x = xn
        kv2d = self.cache.view(B * self.Tcap, 2 * E)
        O = _empty(B, E, dev=dev)
        ops.decode_attn(q, kv2d[:, :E], kv2d[:, E:], self.Tcap, O, H, stop_lens=stop_lens, pos=pos_t, drop_p=p, seed=cx.seed, stream_id=cx.stream())
        z1 = _empty(B, E, dev=dev)
        ops.decode_linear(O, P[lp + "self_attn.out_proj.weight"], P[lp + "self_attn.out_proj.bias"], z1, drop_p=p, seed=cx.seed, stream_id=cx.stream(), R=x)
        # --- cross-attention over the memory: q = W_q norm1(z1)
        Wc, bc = P[lp + "multihead_attn.in_proj_weight"], P[lp + "multihead_attn.in_proj_bias"]
        q2, x1 = _empty(B, E, dev=dev), _empty(B, E, dev=dev)
        ops.decode_linear(z1, Wc[:E], bc[:E], q2, ln=(P[lp + "norm1.weight"], P[lp + "norm1.bias"]), xn_out=x1)
        O2 = _empty(B, E, dev=dev)
        ops.decode_attn(q2, self.memkv[:, :E], self.memkv[:, E:], self.Tk, O2, H, lens=self.lens_mem, drop_p=p, seed=cx.seed, stream_id=cx.stream())
        z2 = _empty(B, E, dev=dev)
        ops.decode_linear(O2, P[lp + "multihead_attn.out_proj.weight"], P[lp + "multihead_attn.out_proj.bias"], z2, drop_p=p, seed=cx.seed, stream_id=cx.stream(), R=x1)
        # --- feed-forward on norm2(z2)
        W1 = P[lp + "linear1.weight"]
        h, x2 = _empty(B, W1.shape[0], dev=dev), _empty(B, E, dev=dev)
        ops.decode_linear(z2, W1, P[lp + "linear1.bias"], h, act=1, drop_p=p, seed=cx.seed, stream_id=cx.stream(),
                          ln=(P[lp + "norm2.weight"], P[lp + "norm2.bias"]), xn_out=x2)
        z3 = _empty(B, E, dev=dev)
        ops.decode_linear(h, P[lp + "linear2.weight"], P[lp + "linear2.bias"], z3, drop_p=p, seed=cx.seed, stream_id=cx.stream(), R=x2)
        return z3


def _run_layers(layers, x, pro, pos_t, stop_lens):
    """The decoder stack on one position; returns (pre-norm sum of the last layer, that norm's (gamma, beta))."""
    for L in layers:
        x = L(x, pro, pos_t, stop_lens)
        pro = {"ln": L.norm_out()}
    return x, pro["ln"]


_CAPTURE_STREAM = {}


def _capture(fn, pool=None, keep_graph=False):
    """Captures fn() into a HIP graph on a dedicated stream.  torch.cuda.graph() is not used: on entry it collects garbage and
    EMPTIES the caching allocator (12 ms per capture here, and the train step that follows has to hipMalloc its whole working
    set again)."""
    dev = torch.cuda.current_device()
    cs = _CAPTURE_STREAM.get(dev)
    if cs is None:
        cs = _CAPTURE_STREAM[dev] = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    graph = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()      # keep_graph: the hipGraph_t stays readable
    from . import engine
    cs.wait_stream(cur)
    with torch.cuda.stream(cs):
        graph.capture_begin(**({"pool": pool} if pool is not None else {}))
        engine.capture_begins(cs)            # explicit waits between forked streams are checked from here on (engine._CaptureWaits)
        try:
            fn()
        except BaseException:
            engine.capture_ends()
            try:
                graph.capture_end()            # close the capture so that the stream is usable again; the error below is the one to report
            except Exception:
                pass
            raise
        engine.capture_ends()
        graph.capture_end()
    cur.wait_stream(cs)
    return graph


class _Generation:
    """One autoregressive generation in flight: `step(epoch)` issues the launches of one position (state in device memory),
    `reset()` rewinds it, `finish(steps)` turns the buffers into the reference's return values."""

    def __init__(self, step, reset, finish, pos_t, stop_lens, max_len):
        self.step, self.reset, self.finish = step, reset, finish
        self.pos_t, self.stop_lens, self.max_len = pos_t, stop_lens, max_len
        self.steps = 0

    def run(self):
        """Decodes until every sequence has stopped or max_len positions.  With config.DECODE_GRAPH the step is captured once and
        replayed."""
        stop_lens, max_len = self.stop_lens, self.max_len

        def all_stopped():
            return not bool((stop_lens == max_len).any())                        # the only host read-back of the loop
        ctr = ops.rng_epoch_counter()
        one = None
        if config.DECODE_GRAPH and max_len >= 2 * SYNC_EVERY:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):                                        # warm-up outside capture (lazy initialisations)
                self.step(None)
            cur.wait_stream(side)
            self.reset()
            one = _capture(lambda: self.step(ctr)).replay                        # ctr: fresh dropout streams at the next position
        steps = 0
        try:
            for i in range(max_len):
                if one is not None:
                    one()
                else:
                    self.step(None)
                steps = i + 1
                if steps % SYNC_EVERY == 0 and all_stopped():
                    break
        finally:
            if one is not None:
                ctr.zero_()                                                      # forward/backward pairs must see epoch 0
        self.steps = steps
        return self.finish(steps)


_PAIR_STREAM = {}


def run_pair(ga, gb):
    """Runs two independent generations in lock-step: one captured graph per position with the two steps on two branches
    (a step is a chain of ~35 dependent, chip-under-filling launches, i.e. bound by latency, so the two chains overlap almost
    perfectly).  When one of them is done the other continues with a graph of its own.  The dropout epoch of both is the
    position, exactly as when they run one after the other.  Returns (ga.finish(..), gb.finish(..))."""
    if not (config.DECODE_GRAPH and min(ga.max_len, gb.max_len) >= 2 * SYNC_EVERY):
        return ga.run(), gb.run()
    ctr = ops.rng_epoch_counter()
    cur = torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    side = _PAIR_STREAM.get(dev)
    if side is None:
        side = _PAIR_STREAM[dev] = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):                                                # warm-up outside capture (lazy initialisations)
        ga.step(None)
        gb.step(None)
    cur.wait_stream(side)
    ga.reset()
    gb.reset()

    def capture(active):
        def both():
            cap = torch.cuda.current_stream()
            side.wait_stream(cap)                                                # fork
            with torch.cuda.stream(side):
                active[1].step(None)
            active[0].step(None)
            cap.wait_stream(side)                                                # join: both branches have read this position's epoch
            ctr.add_(1)
        return _capture(both if len(active) == 2 else (lambda: active[0].step(ctr)))

    active = [ga, gb]
    graph = capture(active)
    i = 0
    try:
        while active:
            graph.replay()
            i += 1
            done = [g for g in active if i >= g.max_len]
            if i % SYNC_EVERY == 0:
                live = torch.stack([(g.stop_lens == g.max_len).any() for g in active]).tolist()   # one host read-back
                done += [g for g, l in zip(active, live) if not l and g not in done]
            if done:
                for g in done:
                    g.steps = i
                active = [g for g in active if g not in done]
                if active:
                    graph = capture(active)
    finally:
        ctr.zero_()                                                              # forward/backward pairs must see epoch 0
    return ga.finish(ga.steps), gb.finish(gb.steps)


def _exit_step(stop_lens, max_len):
    """Iteration count at which the reference's loop exits: the step at which the last sequence stopped, else max_len."""
    sl = stop_lens.tolist()
    return max(sl) if all(s != max_len for s in sl) else max_len


@torch.no_grad()
def infer_text(m, cx, memory, lens_mem, max_len):
    """TextTransformer.infer_sequence (src/network.py:455-481)."""
    return text_generation(m, cx, memory, lens_mem, max_len).run()


@torch.no_grad()
def text_generation(m, cx, memory, lens_mem, max_len):
    """The state of TextTransformer.infer_sequence as a _Generation (cross-attention K/V of the memory are projected here)."""
    B, Tk, E = memory.shape
    dev = memory.device
    a = m.args
    P = cx.P
    mem2d = memory.contiguous().view(B * Tk, E)
    layers = [_LayerStep(cx, "text_m.decoder.transformer_decoder.layers.%d." % l, mem2d, lens_mem, B, Tk, max_len, a.nhead, a.d_drop)
              for l in range(a.num_layers)]
    tokens = torch.full((B, max_len + 1), PAD_IDX, dtype=torch.int64, device=dev)
    tokens[:, 0] = SOS_IDX
    stop_lens = torch.full((B,), max_len, dtype=torch.int64, device=dev)
    Emb = P["text_m.prenet.embed.weight"]
    V = P["text_m.postnet.fc1.weight"].shape[0]
    ldl = (V + 3) // 4 * 4
    pos_t = torch.zeros(1, dtype=torch.int64, device=dev)
    pt = cx.p(a.t_post_drop)
    logits = torch.zeros(B, ldl, dtype=torch.float32, device=dev)                  # row pitch padded to 16 B; the pad stays zero

    def reset():
        pos_t.zero_()
        stop_lens.fill_(max_len)

    def step(epoch):
        # TextPrenet (embedding, dropout) + PositionalEncoding (src/module.py:265-267) are produced inside the first contraction
        first = {"embed": (tokens, Emb, m.pe, math.sqrt(E), (cx.p(a.t_pre_drop), cx.stream()), (cx.p(0.1), cx.stream()))}
        z, ln = _run_layers(layers, None, first, pos_t, stop_lens)
        ops.decode_linear(z, P["text_m.postnet.fc1.weight"], P["text_m.postnet.fc1.bias"], logits, ln=ln, ln_drop=(pt, cx.stream()), seed=cx.seed)
        ops.decode_end_text(logits, V, tokens, stop_lens, max_len, EOS_IDX, pos_t, epoch)   # argmax -> tokens[:, pos+1]; stop rule; pos += 1

    def finish(steps):
        T = min(_exit_step(stop_lens, max_len), steps)
        res = tokens[:, 1:T + 1].contiguous()
        keep = torch.arange(T, device=dev)[None, :] < stop_lens[:, None]
        res = res * keep
        return res, stop_lens

    return _Generation(step, reset, finish, pos_t, stop_lens, max_len)


@torch.no_grad()
def infer_speech(m, cx, memory, lens_mem, max_len, speech_prenet_step, postnet_fn):
    """SpeechTransformer.infer_sequence (src/network.py:219-252)."""
    return speech_generation(m, cx, memory, lens_mem, max_len, speech_prenet_step, postnet_fn).run()


@torch.no_grad()
def speech_generation(m, cx, memory, lens_mem, max_len, speech_prenet_step, postnet_fn):
    """The state of SpeechTransformer.infer_sequence as a _Generation."""
    B, Tk, E = memory.shape
    dev = memory.device
    a = m.args
    M = a.num_mels
    mem2d = memory.contiguous().view(B * Tk, E)
    layers = [_LayerStep(cx, "speech_m.decoder.transformer_decoder.layers.%d." % l, mem2d, lens_mem, B, Tk, max_len, a.nhead, a.d_drop)
              for l in range(a.num_layers)]
    st = cx.st
    Wh = st.span("speech_m.postnet.linear_project.weight", "speech_m.postnet.stop_linear.weight", (M + 1, E))
    bh = st.span("speech_m.postnet.linear_project.bias", "speech_m.postnet.stop_linear.bias", (M + 1,))
    ldh = (M + 1 + 3) // 4 * 4
    outputs = torch.zeros(B, max_len + 1, M, dtype=torch.float32, device=dev)     # position 0 = all-zero "go" frame
    stops = torch.zeros(B, max_len + 1, dtype=torch.float32, device=dev)
    stop_lens = torch.full((B,), max_len, dtype=torch.int64, device=dev)
    pos_t = torch.zeros(1, dtype=torch.int64, device=dev)
    head = torch.zeros(B, ldh, dtype=torch.float32, device=dev)                   # [mel | stop] rows, pitch padded to 16 B

    def reset():
        pos_t.zero_()
        stop_lens.fill_(max_len)

    def step(epoch):
        x = speech_prenet_step(cx, outputs, pos_t)                                # on the frame at position pos
        first = {"posenc": (m.pe, math.sqrt(E), (cx.p(0.1), cx.stream()))}
        z, ln = _run_layers(layers, x, first, pos_t, stop_lens)
        ops.decode_linear(z, Wh, bh, head, ln=ln)
        ops.decode_end_speech(head, M, outputs, stops, stop_lens, max_len, pos_t, epoch)   # frame/stop -> pos+1; stop rule (src/network.py:242)

    @torch.no_grad()
    def finish(steps):
        T = min(_exit_step(stop_lens, max_len), steps)
        outs = outputs[:, :T + 1].contiguous()
        post = postnet_fn(cx, outs)                                              # outputs + postnet(outputs), [B,T+1,M]
        pre_res = outs[:, 1:].contiguous()
        res = post[:, 1:].contiguous()
        res_stop = stops[:, 1:T + 1].contiguous().unsqueeze(-1)
        ops.mask_by_len(pre_res, stop_lens)
        ops.mask_by_len(res, stop_lens)
        ops.mask_by_len(res_stop, stop_lens)
        return pre_res, res, res_stop.squeeze(-1), stop_lens

    return _Generation(step, reset, finish, pos_t, stop_lens, max_len)
