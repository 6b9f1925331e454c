"""The generator phase as one forward and one backward (train.train_gen_joint_step: both sub-steps' encoder passes batched, the frozen
discriminator called once) against the two sub-steps run one after the other, as the reference runs them
(/root/reference/src/train.py:609-628): same losses, same gradients, same BatchNorm running statistics."""
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def build(L, lr, use_discriminator=True):
    from unast_amd import train, utils
    from unast_amd.configs import make_args
    from unast_amd.portable import portable_tensor
    from unast_amd.spec import state_dict_spec
    args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, use_discriminator=use_discriminator)
    train.DEVICE = D
    utils.set_seed(0)
    utils.set_deterministic(True)
    _, _, model, opt, _ = train.initialize_model(args)
    sd = {k: torch.from_numpy(portable_tensor(k, shp, 1234)) for k, shp in state_dict_spec(L, use_discriminator=use_discriminator).items()}
    model.load_state_dict(sd)
    opt.param_groups[0]["lr"] = lr
    return args, model, opt


@pytest.mark.parametrize("use_disc", [True, False])
@pytest.mark.parametrize("shape", [(4, 24, 64), (3, 70, 50), (2, 180, 800)])
def test_joint_generator_step_equals_the_two_substeps(shape, use_disc):
    from unast_amd import train
    from unast_amd.portable import synth_batch
    ae = tuple(torch.from_numpy(x) for x in synth_batch(*shape, seed=1, ragged=True))
    sp = tuple(torch.from_numpy(x) for x in synth_batch(*shape, seed=2, ragged=True))           # another batch of the same shape
    from unast_amd import functional as F
    res = []
    for joint in (False, True):
        before = dict(F.FUSED_STATS)
        args, model, opt = build(2, 1e-3, use_disc)
        losses = defaultdict(list)
        model.train()
        if use_disc:
            train.freeze_model_parameters(model.discriminator)
        if joint:
            assert train.joint_generator_phase(args, ae, sp)
            train.train_gen_joint_step(losses, model, ae, sp, 0, 2, args)
        else:
            train.train_ae_step(losses, model, ae, 0, 2, args)
            train.train_sp_step(losses, model, sp, 0, 2, args)
        model.expose_grads()
        torch.cuda.synchronize()
        used = {k: F.FUSED_STATS[k] - before[k] for k in before}
        if joint:       # the heads computed their losses and gradients themselves, and the loss calls took those gradients as they were
            assert used["text_head"] == 2 and used["text_grad_direct"] == 2 and used["text_grad_general"] == 0, used
            assert used["speech_head"] == 2 and used["speech_grad_direct"] == 2 and used["speech_grad_general"] == 0, used
        else:
            assert all(v == 0 for v in used.values()), used
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        bufs = {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
        res.append(({k: float(v[0]) for k, v in losses.items()}, grads, bufs))
    (la, ga, ba), (lb, gb, bb) = res
    assert set(la) == set(lb) and len(la) == (6 if use_disc else 4)
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(la[k])), (k, la[k], lb[k])
    assert set(ga) == set(gb)
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ga.values())))
    for n in ga:
        d = (ga[n].double() - gb[n].double()).norm().item()
        assert d <= 2e-5 * ga[n].double().norm().item() + 2e-7 * tot, (n, d, ga[n].double().norm().item())
    for k in ba:
        assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-6, atol=1e-7), k


def test_train_step_takes_the_joint_path_only_when_it_can(monkeypatch):
    from unast_amd import config, train
    from unast_amd.portable import synth_batch
    args, model, opt = build(1, 1e-3)
    a = tuple(torch.from_numpy(x) for x in synth_batch(2, 12, 24, seed=1, ragged=True))
    b = tuple(torch.from_numpy(x) for x in synth_batch(2, 14, 24, seed=2, ragged=True))
    assert train.joint_generator_phase(args, a, a) and not train.joint_generator_phase(args, a, b)
    monkeypatch.setattr(config, "JOINT_GEN", False)
    assert not train.joint_generator_phase(args, a, a)
    monkeypatch.setattr(config, "JOINT_GEN", True)
    args.ae_steps = 2
    assert not train.joint_generator_phase(args, a, a)
    args.ae_steps = 1
    calls = []
    orig = train.train_gen_joint_step
    monkeypatch.setattr(train, "train_gen_joint_step", lambda *x, **k: (calls.append(1), orig(*x, **k))[1])
    losses = defaultdict(list)
    train.train_step(losses, model, opt, None, dict(unsup=[a], sup=[a], disc=[a], cm=[]), 0, args)
    train.train_step(losses, model, opt, None, dict(unsup=[a], sup=[b], disc=[a], cm=[]), 1, args)     # other shape: the two sub-steps
    torch.cuda.synchronize()
    assert len(calls) == 1 and len(losses["t_ae"]) == 2 and all(np.isfinite(float(v)) for vs in losses.values() for v in vs)
