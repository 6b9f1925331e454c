"""Rows f-1 / f-3 of SURVEY.md section 8: outer train loop (grad accumulation, LR schedule) and reference-layout checkpoints."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def small_args(**kw):
    from unast_amd.configs import make_args
    d = dict(num_layers=1, ae_steps=2, sp_steps=1, d_steps=1, cm_steps=0, epochs=2, epoch_steps=3, train_batch_size=2, warmup_steps=8, lr=0.01)
    d.update(kw)
    return make_args(**d)


def test_train_loop_reduces_loss_and_follows_schedule(tmp_path):
    from unast_amd import train, utils
    train.DEVICE = D
    utils.set_deterministic(False)
    args = small_args(checkpoint_path=str(tmp_path / "ckpt"), epochs=3, epoch_steps=6)
    lrs = []
    model, hist = train.train(args, batch_getter=train.SyntheticBatchGetter(args, t_text=12, t_mel=32),
                              on_epoch_end=lambda e, m, o, h: lrs.append(o.param_groups[0]["lr"]))
    assert len(hist) == 3 and set(hist[0]) == {"t_ae", "s_ae", "d_ae", "asr_", "tts_", "sp_d", "d"}
    assert hist[-1]["s_ae"] < hist[0]["s_ae"] and hist[-1]["tts_"] < hist[0]["tts_"]          # it learns something
    exp = [args.lr * (s / 8 ** 1.5 if s < 8 else 1 / s ** 0.5) for s in (6, 12, 18)]
    assert all(abs(a - b) < 1e-9 for a, b in zip(lrs, exp)), (lrs, exp)
    assert os.path.isfile(tmp_path / "ckpt" / "model_most_recent.ckpt")


def test_train_loop_through_captured_graphs(tmp_path):
    """train(args) with args.use_hip_graphs: the same loop (2 AE + 1 SP sub-steps accumulated, D phase, LR schedule,
    checkpoints at epoch ends behind a flush of the pending discriminator phase) driven by GraphedTrainStep: one capture for
    the fixed batch shape, every loss key gets its epoch_steps x sub-steps entries, the schedule is followed, it learns."""
    from unast_amd import train, utils
    train.DEVICE = D
    utils.set_deterministic(False)
    args = small_args(checkpoint_path=str(tmp_path / "ckpt"), epochs=3, epoch_steps=6, use_hip_graphs=True)
    lrs, counts = [], []
    model, hist = train.train(args, batch_getter=train.SyntheticBatchGetter(args, t_text=12, t_mel=32, ragged=True),
                              on_epoch_end=lambda e, m, o, h: lrs.append(o.param_groups[0]["lr"]))
    assert len(hist) == 3 and set(hist[0]) == {"t_ae", "s_ae", "d_ae", "asr_", "tts_", "sp_d", "d"}
    assert all(v == v and abs(v) < 1e6 for h in hist for v in h.values())
    assert hist[-1]["s_ae"] < hist[0]["s_ae"] and hist[-1]["tts_"] < hist[0]["tts_"]
    exp = [args.lr * (s / 8 ** 1.5 if s < 8 else 1 / s ** 0.5) for s in (6, 12, 18)]
    assert all(abs(a - b) < 1e-9 for a, b in zip(lrs, exp)), (lrs, exp)
    assert os.path.isfile(tmp_path / "ckpt" / "model_most_recent.ckpt")
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_train_loop_through_captured_graphs_with_changing_batch_shapes(tmp_path):
    """The same loop when the padded batch shape changes from step to step (what a length-sorted loader produces): three shapes
    cycle, sub-steps of one outer step may differ too.  The stepper keeps one capture and one set of input buffers per shape,
    runs the pending discriminator phase before it switches -- losses stay finite, every key gets all its entries, it learns."""
    from unast_amd import train, utils
    from unast_amd.portable import synth_batch
    train.DEVICE = D
    utils.set_deterministic(False)
    args = small_args(checkpoint_path=str(tmp_path / "ckpt"), epochs=3, epoch_steps=8, use_hip_graphs=True)

    class Cycling:
        shapes = [(12, 32), (9, 24), (12, 32), (15, 40)]

        def __init__(self):
            self.n = 0

        def _next(self):
            self.n += 1
            tt, tm = self.shapes[(self.n // 3) % len(self.shapes)]
            return tuple(torch.from_numpy(x) for x in synth_batch(args.train_batch_size, tt, tm, seed=self.n, ragged=True))
        get_supervised_batch = get_unsupervised_batch = get_discriminator_batch = _next

    model, hist = train.train(args, batch_getter=Cycling())
    assert len(hist) == 3 and set(hist[0]) == {"t_ae", "s_ae", "d_ae", "asr_", "tts_", "sp_d", "d"}
    assert all(v == v and abs(v) < 1e6 for h in hist for v in h.values())
    assert hist[-1]["s_ae"] < hist[0]["s_ae"] and hist[-1]["tts_"] < hist[0]["tts_"]
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_checkpoint_roundtrip_and_torch_adamw_format(tmp_path):
    from collections import defaultdict
    from unast_amd import train, utils
    from unast_amd.checkpoint import save_ckp, load_ckp
    from unast_amd.portable import synth_batch
    train.DEVICE = D
    utils.set_deterministic(True, fixed_sums=True)        # every fp32 sum in a fixed order: the resumed step must equal the original one to the bit
    args = small_args(ae_steps=1)
    batch = tuple(torch.from_numpy(x) for x in synth_batch(2, 12, 32, seed=5, ragged=True))
    batches = dict(unsup=[batch], sup=[batch], disc=[batch])
    utils.set_seed(3)
    _, _, model, opt, _ = train.initialize_model(args)
    opt.param_groups[0]["lr"] = 1e-3
    train.train_step(defaultdict(list), model, opt, None, batches, 0, args)
    save_ckp(0, 1.0, model, opt, True, str(tmp_path))
    ck = torch.load(tmp_path / "model_best.ckpt", weights_only=False)
    assert set(ck) == {"epoch", "valid_loss_min", "state_dict", "optimizer"} and ck["epoch"] == 1
    # the optimizer entry must load into a stock torch.optim.AdamW over parameters of the reference shapes
    ref_params = [torch.nn.Parameter(v.clone().float()) for k, v in ck["state_dict"].items()
                  if v.dtype.is_floating_point and "running_" not in k and not k.endswith(".pe")]
    topt = torch.optim.AdamW(ref_params, lr=1e-3, weight_decay=args.weight_decay)
    topt.load_state_dict(ck["optimizer"])
    assert len(topt.state) == len(ref_params) - 2                         # reduce_c_W.{weight,bias} never updated
    i_conv = [k for k, v in ck["state_dict"].items() if v.dtype.is_floating_point and "running_" not in k and not k.endswith(".pe")].index("text_m.prenet.conv1.conv.weight")
    assert tuple(topt.state[ref_params[i_conv]]["exp_avg"].shape) == (256, 256, 5)
    # resume: a second step from the restored state equals a second step of the original run
    train.train_step(defaultdict(list), model, opt, None, batches, 1, args)
    after_orig = model._store().flat.detach().clone()
    utils.set_seed(99)
    _, _, model2, opt2, _ = train.initialize_model(args)
    opt2.param_groups[0]["lr"] = 1e-3
    ep, vl, model2, opt2 = load_ckp(str(tmp_path / "model_best.ckpt"), model2, opt2)
    assert ep == 1 and vl == 1.0
    train.train_step(defaultdict(list), model2, opt2, None, batches, 1, args)
    st2 = model2._store()
    dd = (st2.flat - after_orig).abs()
    diff = dd.max().item()
    worst = max((n for n, o in st2.offsets.items() if o <= int(dd.argmax())), key=lambda n: st2.offsets[n])
    utils.set_deterministic(False, fixed_sums=False)
    assert diff == 0.0, (diff, worst, int((dd > 0).sum()))
