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


def set_deterministic(flag=True, fixed_sums=None):
    """Parity mode: identity permutation in discriminator_shuffle_batch and no SpecAugment (dropout is governed by the
    configured rates / model.eval()).  Mirrors the oracle harness of SURVEY.md Appendix C.
    fixed_sums (True / False; None leaves it as it is): additionally form every fp32 sum of the step in a fixed order
    (config.DETERMINISTIC_SUMS, ops.deterministic_sums): two runs of a step, eager or replayed, then agree to the bit -- at the
    price of slower kernels (two-kernel attention backward, ungrouped weight gradients, ordered column sums)."""
    _RNG["deterministic"] = bool(flag)
    if fixed_sums is not None:
        from . import config
        config.DETERMINISTIC_SUMS = bool(fixed_sums)


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


class TeacherRatio:
    """Teacher-forcing ratio schedule with the reference's attribute names (src/utils.py:116-136): constant `val` until
    `start_step`, then val * gamma^(min(iter, stop_step) - start_step).  (The reference constructs it and never advances it:
    its `.step()` call is commented out at src/train.py:664, so the ratio stays at teacher_init_val on the hot path.)"""

    def __init__(self, args):
        self.val, self.gamma = args.teacher_init_val, args.teacher_gamma
        self.start_step, self.stop_step = args.teacher_decay_start, args.teacher_decay_end
        self.iter = 0

    def step(self):
        self.iter += 1

    def get_val(self):
        decayed_for = min(self.iter, self.stop_step) - self.start_step
        return self.val if self.iter < self.start_step else self.val * self.gamma ** decayed_for


def get_teacher_ratio(args):
    return TeacherRatio(args)


# ---------------------------------------------------------------------------------------------------------------
# Evaluation helpers (src/utils.py:24-38)
# ---------------------------------------------------------------------------------------------------------------
def edit_distance(ref, hyp):
    """Levenshtein distance between two integer sequences (unit costs).  One DP row per reference symbol; the
    insertion chain inside a row is resolved with a running minimum, so each row is a handful of numpy vector ops."""
    ref = np.asarray(ref, dtype=np.int64).ravel()
    hyp = np.asarray(hyp, dtype=np.int64).ravel()
    n, m = ref.size, hyp.size
    if n == 0 or m == 0:
        return int(max(n, m))
    ar = np.arange(m + 1, dtype=np.int64)
    prev = ar.copy()
    cur = np.empty(m + 1, dtype=np.int64)
    for i in range(1, n + 1):
        cur[0] = i
        np.minimum(prev[1:] + 1, prev[:-1] + (hyp != ref[i - 1]), out=cur[1:])
        cur[:] = np.minimum.accumulate(cur - ar) + ar              # cur[j] = min(cur[j], cur[j-1] + 1) for all j
        prev, cur = cur, prev
    return int(prev[m])


def compute_per(ground_truth, hypothesis, ground_truth_lengths, hypothesis_lengths):
    """Phoneme error rate of a batch (src/utils.py:24-34).  The reference joins each id sequence into a string of
    "words" and calls jiwer.wer (jiwer==2.2.0, requirements.txt:10 -- not installed in this image).  That version's
    published algorithm: both sentence lists are flattened into ONE word list each, and
    wer = (S + D + I) / (H + S + D) = Levenshtein(truth, hypothesis) / len(truth)."""
    gt, hy = _to_lists(ground_truth), _to_lists(hypothesis)
    gl, hl = _to_lists(ground_truth_lengths), _to_lists(hypothesis_lengths)
    ref = [t for b in range(len(gt)) for t in gt[b][:gl[b]]]
    hyp = [t for b in range(len(gt)) for t in hy[b][:hl[b]]]
    if not ref:
        raise ValueError("compute_per: empty ground truth")
    return edit_distance(ref, hyp) / float(len(ref))


def _to_lists(t):
    return t.detach().cpu().tolist() if torch.is_tensor(t) else np.asarray(t).tolist()


def compare_outputs(ground_truth, hypothesis, gt_len, hyp_len):
    """src/utils.py:36-38; prints ids (the symbol table belongs to the reference's text front end, outside this path)."""
    print("Model prediction of length %d " % int(hyp_len), _to_lists(hypothesis)[:int(hyp_len)])
    print("Ground Truth of length %d " % int(gt_len), _to_lists(ground_truth)[:int(gt_len)])
