"""Hot-path helpers with the reference's names (src/utils.py): constants, seeding, masks, noise, SpecAugment,
schedule helpers.  Everything that touches activations runs through the HIP kernels."""
import random

import numpy as np
import torch

from . import ops

PAD_IDX = 0   # src/utils.py:19-21
SOS_IDX = 1
EOS_IDX = 2

_RNG = {"seed": 0, "counter": 0, "deterministic": False}


def set_seed(seed):
    """src/utils.py:85-98; additionally seeds the counter RNG of the HIP dropout/noise kernels."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    _RNG["seed"], _RNG["counter"] = int(seed), 0


def next_seed():
    """A fresh 32-bit seed per public model call: masks differ between calls and steps, and are reproducible."""
    _RNG["counter"] += 1
    x = (_RNG["seed"] * 0x9E3779B1 + _RNG["counter"] * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return x


def set_deterministic(flag=True):
    """Parity mode: identity permutation in discriminator_shuffle_batch and no SpecAugment (dropout is governed by the
    configured rates / model.eval()).  Mirrors the oracle harness of SURVEY.md Appendix C."""
    _RNG["deterministic"] = bool(flag)


def is_deterministic():
    return _RNG["deterministic"]


def lens_i32(lens, device=None):
    if lens.dtype == torch.int32 and (device is None or lens.device == device):
        return lens
    return lens.to(device=device or lens.device, dtype=torch.int32)


def sent_lens_to_mask(lens, max_length):
    """src/utils.py:77-83 without the B*T host loop: mask[b,t] = t < lens[b].  (The kernels never need this tensor;
    it exists for API compatibility.)"""
    return torch.arange(max_length, device=lens.device)[None, :] < lens[:, None]


def noise_fn(to_noise, mask_p=.3, swap_p=0):
    """src/utils.py:40-49 on the GPU: zero whole timesteps with probability mask_p, no rescale."""
    B, T, Dm = to_noise.shape
    x = to_noise.contiguous().view(B * T, Dm)
    y = torch.empty_like(x)
    ops.rowmask(x, y, mask_p, next_seed(), 1)
    return y.view(B, T, Dm)


def specaugment(mel, mel_len, freq_mask=20, time_mask=100, replace_with_zero=False):
    """src/utils.py:51-75 (two time spans replaced by the per-sample mean)."""
    if replace_with_zero:
        raise NotImplementedError("replace_with_zero is never used on the train path (src/train.py:236)")
    mel = mel.detach().contiguous()
    out = torch.empty_like(mel)
    ops.specaugment(mel, lens_i32(mel_len, mel.device), out, next_seed(), 1, freq_mask, time_mask)
    return out


def init_device(args):
    """src/utils.py:101-106 — but there is no CPU path: the HIP kernels are the product."""
    if not (torch.cuda.is_available() and getattr(args, "use_gpu", True)):
        raise RuntimeError("unast_amd needs an MI355X (ROCm) device; no CPU fallback exists")
    return torch.device("cuda")


class TeacherRatio():
    """src/utils.py:116-136."""

    def __init__(self, args):
        self.iter = 0
        self.val = args.teacher_init_val
        self.gamma = args.teacher_gamma
        self.start_step = args.teacher_decay_start
        self.stop_step = args.teacher_decay_end

    def step(self):
        self.iter += 1

    def get_val(self):
        if self.start_step <= self.iter:
            power = min(self.iter, self.stop_step) - self.start_step
            return self.val * (self.gamma ** power)
        return self.val


def get_teacher_ratio(args):
    return TeacherRatio(args)
