"""The train step replayed from a captured HIP graph (unast_amd.graphed) against the same steps launched kernel by kernel."""
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def build(L, lr, use_discriminator=True, sched_type=None):
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, use_discriminator=use_discriminator, lr=lr,
                     sched_type=sched_type, warmup_steps=3)
    train.DEVICE = D
    utils.set_seed(0)
    _, _, model, opt, sched = train.initialize_model(args)
    sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L, use_discriminator=use_discriminator).items()}
    model.load_state_dict(sd)
    return args, model, opt, sched


def batches_for(i, B=4, Tt=28, Tm=96):
    from unast_amd.portable import synth_batch
    mk = lambda s: tuple(torch.from_numpy(x).to(D) for x in synth_batch(B, Tt, Tm, seed=s, ragged=True))
    return dict(unsup=[mk(3 * i)], sup=[mk(3 * i + 1)], disc=[mk(3 * i + 2)], cm=[])


def _run_steps(graphed, use_disc, lr, n=6):
    from unast_amd import train
    from unast_amd.engine import join_streams
    from unast_amd.graphed import GraphedTrainStep
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    args, model, opt, sched = build(2, lr, use_discriminator=use_disc, sched_type="linear")
    args.epochs, args.epoch_steps = 1, 12
    _, _, model, opt, sched = train.initialize_model(args)       # scheduler built with these totals
    model.load_state_dict({k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(2, use_discriminator=use_disc).items()})
    sched.step()                                                 # lr > 0 at the first step
    losses = defaultdict(list)
    stepper = GraphedTrainStep(model, opt, sched, args) if graphed else None
    for i in range(n):
        if graphed:
            stepper(losses, batches_for(i), i)
        else:
            train.train_step(losses, model, opt, sched, batches_for(i), i, args, defer_d_phase=True)
    if graphed:
        assert len(stepper.graphs) == 1
        from unast_amd import graphed as G
        rec = next(iter(stepper.graphs.values()))
        if G.REPLAY == "streams":          # the executor understood every node of the captured step (no silent fallback)
            assert rec.plan and rec.plan_info["kernels"] > 300 and rec.plan_info["cross_stream_edges"] > 5, rec.plan_info
        else:
            assert not rec.plan
        stepper.flush(losses)
    join_streams()
    torch.cuda.synchronize()
    return ({k: [float(x) for x in v] for k, v in losses.items()}, model._store().flat.detach().cpu().clone(), dict(opt._steps), opt.param_groups[0]["lr"])


@pytest.mark.parametrize("replay", ["streams", "hipgraph"])
@pytest.mark.parametrize("use_disc", [True, False])
def test_graph_replay_and_eager_steps_agree(use_disc, replay, monkeypatch):
    """Six outer steps with a different batch per sub-step and a learning-rate schedule (linear warm-up then decay, so every
    step has another lr) through train_step(defer_d_phase=True) and through GraphedTrainStep (two eager calls, one capture,
    three pure replays).  RNG sites off, so the two differ by accumulation-order noise only.
    (a) lr ~ 1e-7: parameters effectively frozen -> every loss of every sub-step agrees to 2e-5: the replayed kernels see the
        right inputs in the right order (static input buffers, stream joins, loss snapshots);
    (b) lr = 4e-4: the first three steps (eager generator phase, eager shifted step, first replay) agree to 3e-4, the later ones to 2e-2 --
        Adam moves elements with a near-zero gradient by +-lr on rounding noise and the text side's first layer amplifies that
        (DESIGN.md section 3); step counts, final lr and the parameters (to a few lr) agree."""
    from unast_amd import utils, graphed
    monkeypatch.setattr(graphed, "REPLAY", replay)       # "streams": csrc/graph_exec.cpp re-issues the captured nodes; "hipgraph": hipGraphLaunch
    utils.set_deterministic(True)
    (la, pa, sa, lra), (lb, pb, sb, lrb) = _run_steps(False, use_disc, 1e-7), _run_steps(True, use_disc, 1e-7)
    assert sa == sb and lra == lrb and set(la) == set(lb)
    for k in la:
        assert len(la[k]) == len(lb[k]) == 6, (k, len(la[k]), len(lb[k]))
        for i, (x, y) in enumerate(zip(la[k], lb[k])):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), ("frozen", k, i, x, y)
    (la, pa, sa, lra), (lb, pb, sb, lrb) = _run_steps(False, use_disc, 4e-4), _run_steps(True, use_disc, 4e-4)
    assert sa == sb and lra == lrb and set(la) == set(lb)
    for k in la:
        for i, (x, y) in enumerate(zip(la[k], lb[k])):
            assert abs(x - y) <= (3e-4 if i < 3 else 2e-2) * max(1.0, abs(x)), (k, i, x, y)
    d = (pa - pb).abs()
    assert float(d.max()) <= 8 * 4e-4 and float((d > 2e-5).float().mean()) < 0.3, (float(d.max()), float((d > 2e-5).float().mean()))


@pytest.mark.parametrize("use_disc", [True, False])
def test_fixed_summation_order_makes_runs_agree_to_the_bit(use_disc, monkeypatch):
    """utils.set_deterministic(True, fixed_sums=True): every fp32 sum of the step is formed in a fixed order (two-kernel attention
    backward, ordered bias / embedding gradients, ungrouped weight gradients; ops.deterministic_sums).  Then six steps at lr = 4e-4 --
    the setting in which rounding noise grows tenfold per step otherwise (profiles/r04_replay_noise.txt) -- give (a) the SAME losses and
    parameters, bit for bit, in two eager runs, and (b) the same in the captured and stream-replayed run as in the eager one: any race
    or mis-ordered replay shows up as a difference instead of hiding under a noise bound."""
    from unast_amd import utils, graphed
    monkeypatch.setattr(graphed, "REPLAY", "streams")
    utils.set_deterministic(True, fixed_sums=True)
    try:
        (la, pa, sa, lra) = _run_steps(False, use_disc, 4e-4)
        (lb, pb, sb, lrb) = _run_steps(False, use_disc, 4e-4)
        from unast_amd import engine
        pruned0 = engine.PRUNED[0]
        (lc, pc, sc, lrc) = _run_steps(True, use_disc, 4e-4)
        assert engine.PRUNED[0] > pruned0, "the capture pruned no inherited dependency: is engine._Segment._backward's pruning off?"
        # ... and with the streams of the captured step shifted against each other by random spins (baked into the capture): a dependency
        # the capture lost -- engine._Segment._backward prunes what a stream inherits through the origin's relay -- has to show here
        from unast_amd import config
        monkeypatch.setattr(config, "STREAM_JITTER", 300)
        for seed in (11, 12):
            monkeypatch.setattr(config, "STREAM_JITTER_SEED", seed)
            (lj, pj, sj, lrj) = _run_steps(True, use_disc, 4e-4)
            assert lj == la and torch.equal(pj, pa), ("jittered replay", seed, float((pj - pa).abs().max()))
        monkeypatch.setattr(config, "STREAM_JITTER", 0)
    finally:
        utils.set_deterministic(True, fixed_sums=False)
    assert la == lb, ("two eager runs", {k: (la[k], lb[k]) for k in la if la[k] != lb[k]})
    assert torch.equal(pa, pb), float((pa - pb).abs().max())
    assert sa == sc and lra == lrc
    worst = max(abs(x - y) / max(1.0, abs(x)) for k in la for x, y in zip(la[k], lc[k]))
    assert worst == 0.0 and torch.equal(pa, pc), ("eager vs replayed", worst, float((pa - pc).abs().max()))


def test_graph_replays_draw_fresh_masks_and_permutations():
    """With the RNG sites on, replays of ONE captured graph on the SAME batch give different losses (dropout / noise /
    SpecAugment masks and the discriminator's row permutation follow the RNG epoch in device memory), all finite, and the
    parameters keep moving."""
    from unast_amd import utils
    from unast_amd.graphed import GraphedTrainStep
    utils.set_deterministic(False)
    try:
        args, model, opt, sched = build(2, 2e-4)
        stepper = GraphedTrainStep(model, opt, None, args)
        losses = defaultdict(list)
        b = batches_for(0)
        for i in range(6):
            stepper(losses, b, i)
        stepper.flush(losses)
        torch.cuda.synchronize()
        for k, v in losses.items():
            vals = [float(x) for x in v]
            assert len(vals) == 6 and all(np.isfinite(vals)), (k, vals)
            assert len(set(round(x, 6) for x in vals[3:])) == 3, (k, vals)      # calls 3-5 are pure replays
        # same (seed, stream) but another epoch -> another permutation
        from unast_amd import ops
        ops.set_step_state(1, {}); p1 = ops.randperm(64, 5, 1, D)
        ops.set_step_state(2, {}); p2 = ops.randperm(64, 5, 1, D)
        ops.set_step_state(1, {}); p3 = ops.randperm(64, 5, 1, D)
        assert sorted(p1.tolist()) == list(range(64)) and sorted(p2.tolist()) == list(range(64))
        assert p1.tolist() != p2.tolist() and p1.tolist() == p3.tolist()
    finally:
        utils.set_deterministic(True)
        from unast_amd import ops
        ops.rng_epoch_counter().zero_()


def test_masks_of_neighbouring_sites_do_not_repeat_across_epochs():
    """Sites of one call use consecutive stream ids and replays consecutive epochs: the mask of site s at epoch e + 1 must not be
    the mask site s + 1 drew at epoch e (it was, while stream id and epoch were added under one hash), and the same
    (site, epoch) must reproduce."""
    from unast_amd import ops
    x = torch.ones(4096, 80, device=D)

    def mask(stream_id, epoch):
        ops.set_step_state(epoch, {})
        y = torch.empty_like(x)
        ops.rowmask(x, y, 0.3, 77, stream_id)
        return y[:, 0].clone()
    try:
        a = mask(4, 5)
        assert torch.equal(a, mask(4, 5))
        assert 0.25 < 1.0 - float(a.mean()) < 0.35
        for s, e in ((3, 6), (5, 4), (4, 6), (5, 5)):
            b = mask(s, e)
            agree = float((a == b).float().mean())                  # independent masks agree on 0.7^2 + 0.3^2 = 0.58 of the rows
            assert 0.52 < agree < 0.64, (s, e, agree)
    finally:
        ops.rng_epoch_counter().zero_()


def test_randperm_is_uniform():
    """Every position receives every value about equally often (4096 draws of a 16-permutation)."""
    from unast_amd import ops
    ops.rng_epoch_counter().zero_()
    n, draws = 16, 4096
    counts = torch.zeros(n, n)
    for s in range(draws):
        p = ops.randperm(n, s, 1, D).cpu()
        counts[torch.arange(n), p] += 1
    exp = draws / n
    assert float((counts - exp).abs().max()) < 6 * (exp ** 0.5), float((counts - exp).abs().max())


def test_graph_replay_with_alternating_input_shapes():
    """Batches of two different padded shapes alternate (A A A A B B B B A A A A): each shape keeps its own capture AND the
    static input buffers that capture was recorded with (a capture replayed against re-allocated buffers would read freed
    memory: token ids and lengths of garbage).  Frozen parameters (lr ~ 1e-7), RNG sites off: every loss equals the eager run's."""
    from unast_amd import train, utils
    from unast_amd.engine import join_streams
    from unast_amd.graphed import GraphedTrainStep
    shapes = [(4, 28, 96)] * 4 + [(3, 20, 64)] * 4 + [(4, 28, 96)] * 4
    out = []
    for graphed in (True, False):
        utils.set_deterministic(True)
        try:
            args, model, opt, sched = build(2, 1e-7)
            losses = defaultdict(list)
            stepper = GraphedTrainStep(model, opt, None, args) if graphed else None
            for i, (B, Tt, Tm) in enumerate(shapes):
                b = batches_for(i, B, Tt, Tm)
                if graphed:
                    stepper(losses, b, i)
                else:
                    train.train_step(losses, model, opt, None, b, i, args, defer_d_phase=True)
            if graphed:
                assert len(stepper.graphs) == 2 and len(stepper.static_gen) == 2 and len(stepper.static_disc) == 2
                assert stepper.stats["captures"] == 2 and stepper.stats["replays"] == 7 and stepper.stats["evictions"] == 0, stepper.stats
                stepper.flush(losses)
            join_streams(); torch.cuda.synchronize()
        finally:
            utils.set_deterministic(False)
        out.append({k: [float(x) for x in v] for k, v in losses.items()})
    g, e = out
    for k in e:
        assert len(g[k]) == len(e[k]), k
        for x, y in zip(g[k], e[k]):
            assert abs(x - y) < 2e-5 * max(1.0, abs(y)), (k, x, y)
    assert all(np.isfinite(v).all() for v in g.values())


def test_odd_shapes_and_step_layouts_eager_vs_replay():
    """tools/stress_shapes.py: tiny and odd lengths, batch 1, more input signatures than cached captures (eviction), accumulation
    over two sub-steps of each kind, generator only -- finite losses and eager == replay at frozen parameters, in a fresh process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_shapes.py")], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0 and b"stress ok" in out.stdout, out.stdout.decode()[-3000:]



class _ShapedBatches:
    """Batch getter whose batches follow a list of (B, T_text, T_mel) shapes, one entry per outer step (ae 1, sp 1, d 1)."""

    def __init__(self, shapes):
        self.shapes, self.calls = shapes, 0

    def _next(self):
        from unast_amd.portable import synth_batch
        B, Tt, Tm = self.shapes[(self.calls // 3) % len(self.shapes)]
        self.calls += 1
        return tuple(torch.from_numpy(x) for x in synth_batch(B, Tt, Tm, seed=self.calls, ragged=True))

    get_supervised_batch = get_unsupervised_batch = get_discriminator_batch = _next


def test_twelve_shapes_through_train_are_captured_once_each():
    """train(use_hip_graphs=True) over a loader with twelve distinct padded shapes in runs of three steps, two passes: every pair of
    shapes met twice is captured exactly once and stays cached (no eviction, no re-capture), later meetings replay."""
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    shapes = [(3, 8 + 2 * i, 32 + 8 * i) for i in range(12)]
    seq = [s for _ in range(2) for s in shapes for _ in range(3)]
    utils.set_deterministic(False)
    train.DEVICE = D
    args = make_args(num_layers=2, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, epochs=1, epoch_steps=len(seq), use_hip_graphs=True, train_batch_size=3)
    model, hist = train.train(args, batch_getter=_ShapedBatches(seq))
    rep = model.__dict__["_graph_stepper"].cache_report()
    assert all(v == v for v in hist[0].values())
    # 71 bodies (the first step has none).  Pass 1: per run one cross pair (eager) and the same-shape pair twice (eager, then capture +
    # replay); pass 2: the pair (last shape, first shape) is new (eager), the 11 other cross pairs are met again (captured), same-shape
    # pairs replay
    assert rep["evictions"] == 0 and rep["captures"] == rep["cached"] == 12 + 11, rep
    assert rep["shapes_gen"] == 12 and rep["shapes_disc"] == 12, rep
    assert rep["eager_bodies"] == 12 + 11 + 1, rep
    assert rep["replays"] == 71 - 24, rep
    sigs = [(c["disc_shapes"], c["gen_shapes"]) for c in rep["per_capture"]]
    assert len(set(sigs)) == len(sigs)
    assert all(c["capture_ms"] > 0 and c["replays"] >= 1 for c in rep["per_capture"])


def test_lru_keeps_the_hot_shape_while_cold_ones_pass_through(monkeypatch):
    """Four cached captures, one hot shape met between runs of five cold ones: the hot capture is never evicted nor re-captured (a
    first-in-first-out cache would drop it after four cold captures)."""
    import unast_amd.graphed as G
    from unast_amd import utils
    monkeypatch.setattr(G, "MAX_GRAPHS", 4)
    hot = (3, 12, 40)
    seq = [hot] * 4
    for i in range(5):
        seq += [(2, 8 + 2 * i, 48 + 8 * i)] * 3 + [hot] * 3
    utils.set_deterministic(False)
    args, model, opt, sched = build(2, 1e-7)
    stepper = G.GraphedTrainStep(model, opt, None, args)
    captured = []
    real = stepper._capture
    monkeypatch.setattr(stepper, "_capture", lambda sig: (captured.append(sig), real(sig))[1])
    losses = defaultdict(list)
    for i, (B, Tt, Tm) in enumerate(seq):
        stepper(losses, batches_for(i, B, Tt, Tm), i)
    stepper.flush(losses)
    torch.cuda.synchronize()
    hot_sig = [s for s in captured if s[0][0][2][1] == (hot[0], hot[2], 80) and s[1][0][2][1] == (hot[0], hot[2], 80)]
    assert len(hot_sig) == 1, captured
    assert len(captured) == len(set(captured)), "a capture still cached was made again"
    assert stepper.stats["evictions"] >= 1 and len(stepper.graphs) <= 4 and hot_sig[0] in stepper.graphs, stepper.stats
    assert all(np.isfinite([float(x) for x in v]).all() for v in losses.values())


def test_mutual_waits_between_side_streams_are_refused_inside_a_capture():
    """ROCm 7.2's hipStreamEndCapture crashes (inside the runtime) when two streams forked from the capture's origin wait on each other
    (A waits for B after B waited for A: tools/debug_capture.py T1).  The package never builds that topology and its wait helper refuses
    it with a Python error while the capture is still open -- exactly that topology here, through engine.wait; one-way waits and waits
    through the origin stay legal, and outside a capture nothing is checked."""
    from unast_amd import engine
    from unast_amd.inference import _capture
    x = torch.zeros(1 << 16, device=D)
    A, B = torch.cuda.Stream(), torch.cuda.Stream()

    def legal():
        O = torch.cuda.current_stream()
        engine.wait(A, O); engine.wait(B, O)
        with torch.cuda.stream(A):
            x.add_(1.0)
        engine.wait(B, A)                       # one way: fine
        with torch.cuda.stream(B):
            x.add_(1.0)
        engine.wait(O, B); engine.wait(A, O)    # ... and back through the origin: fine
        with torch.cuda.stream(A):
            x.add_(1.0)
        engine.wait(O, A)
    g = _capture(legal)
    x.zero_(); g.replay(); torch.cuda.synchronize()
    assert float(x[0]) == 3.0

    def cyclic():
        O = torch.cuda.current_stream()
        engine.wait(A, O); engine.wait(B, O)
        with torch.cuda.stream(A):
            x.add_(1.0)
        engine.wait(B, A)
        with torch.cuda.stream(B):
            x.add_(1.0)
        try:
            engine.wait(A, B)                   # B has waited for A: refused
        finally:
            engine.wait(O, A); engine.wait(O, B)
    with pytest.raises(RuntimeError, match="may not wait on each other"):
        _capture(cyclic)
    torch.cuda.synchronize()
    engine.wait(A, B); engine.wait(B, A)        # outside a capture: ordinary waits
    torch.cuda.synchronize()
    g2 = _capture(legal)                        # and the next capture works
    x.zero_(); g2.replay(); torch.cuda.synchronize()
    assert float(x[0]) == 3.0


def test_loss_workspaces_are_zero_whenever_they_are_handed_out(monkeypatch):
    """config.DEBUG_WORKSPACES: every loss-workspace slot (train._loss_ws ring, ops.masked_mse per stream) is checked to be zero on entry
    -- the invariant the zero-on-exit kernels rest on -- over 70 sub-step calls (the 64-slot ring wraps) and masked_mse on two streams."""
    from unast_amd import config, train, ops
    from unast_amd.portable import synth_batch
    monkeypatch.setattr(config, "DEBUG_WORKSPACES", True)
    args, model, opt = build(2, 1e-3)[:3]
    batch = tuple(torch.from_numpy(x) for x in synth_batch(2, 12, 24, seed=1, ragged=True))
    losses = defaultdict(list)
    for i in range(12):
        train.train_step(losses, model, opt, None, dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]), i, args)
    g = torch.Generator().manual_seed(0)
    gold, pred = torch.rand(5, 7, 80, generator=g).to(D), torch.rand(5, 7, 80, generator=g).to(D)
    mask = (torch.rand(5, 7, 80, generator=g) > 0.3).float().to(D)
    ref = float(((gold - pred) ** 2 * mask).sum() / mask.sum())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for _ in range(3):
        a = train.masked_mse(gold, pred, mask)
        with torch.cuda.stream(side):
            b = train.masked_mse(gold, pred, mask)
    torch.cuda.synchronize()
    assert abs(float(a) - ref) < 1e-5 * ref and abs(float(b) - ref) < 1e-5 * ref
    with pytest.raises(RuntimeError, match="forward-only"):
        train.masked_mse(gold, pred.clone().requires_grad_(True), mask)


def test_loss_workspace_ring_wraps_without_harm():
    """Thirty eager steps: the loss kernels' zero-on-entry workspaces come from a 64-slot ring shared by the text and the speech loss, so
    slots change hands between the two kernels after about ten steps (the text loss's saved weight sum must not be taken for the speech
    loss's arrival counter); losses stay finite and equal those of a run that gets a fresh zeroed workspace per call."""
    from unast_amd import train, utils
    from unast_amd.engine import join_streams
    out = []
    for fresh in (False, True):
        utils.set_deterministic(True)
        try:
            args, model, opt, sched = build(2, 1e-7)            # frozen parameters: a step's losses depend on its batch only
            real = train._loss_ws
            if fresh:
                train._loss_ws = lambda dev: torch.zeros(8, dtype=torch.float64, device=dev)
            losses = defaultdict(list)
            try:
                for i in range(30):
                    train.train_step(losses, model, opt, None, batches_for(i % 3, 3, 20, 48), i, args, defer_d_phase=True)
            finally:
                train._loss_ws = real
            join_streams(); torch.cuda.synchronize()
            out.append({k: [float(x) for x in v] for k, v in losses.items()})
        finally:
            utils.set_deterministic(False)
    a, b = out
    for k in a:
        assert len(a[k]) == 30 and np.isfinite(a[k]).all(), k
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-6), (k, np.abs(np.array(a[k]) - np.array(b[k])).max())


def test_encoder_backward_without_padded_query_tiles_gives_the_same_gradients(monkeypatch):
    """config.ENC_SKIP_PAD_GRADS (opt-in): the backward of the encoders' self-attention stops at each sequence's length.  In the train step
    every gradient that reaches a padded encoder position is exactly zero, so the generator's gradients must not change (ragged batch)."""
    from unast_amd import config, train, utils
    from unast_amd.engine import join_streams
    grads = []
    utils.set_deterministic(True)
    try:
        for flag in (False, True):
            monkeypatch.setattr(config, "ENC_SKIP_PAD_GRADS", flag)
            args, model, opt, sched = build(2, 1e-7)
            losses = defaultdict(list)
            b = batches_for(5, 4, 28, 96)
            train.freeze_model_parameters(model.discriminator)
            train.train_ae_step(losses, model, b["unsup"][0], 0, 2, args)
            train.train_sp_step(losses, model, b["sup"][0], 0, 2, args)
            join_streams(); torch.cuda.synchronize()
            st = model._store()
            a, e = st.regions["gen"]
            grads.append(st.grad[a:e].clone())
    finally:
        utils.set_deterministic(False)
    g0, g1 = grads
    assert bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0
    assert float((g0 - g1).abs().max()) <= 2e-6 * float(g0.abs().max()), float((g0 - g1).abs().max())
