"""Portable, seeded weights and synthetic LJSpeech-shaped batches.

Shared by the golden-vector generator (tools/gen_golden.py, which loads these weights into the
imported reference), the oracle tests and bench.py, so that both sides of every parity check
start from bit-identical inputs without shipping 69 MB of weights.  Pure numpy: no HIP, no
reference import.

Batch contract follows the reference's collate_fn_transformer (src/preprocess.py:82-118):
text int64 [B,Tt] zero padded with EOS(2) as last real token, mel float32 [B,Tm,80] in [0,1)
zero padded, lengths int64, batch sorted by text length descending.
"""
import zlib

import numpy as np

N_SYMBOLS = 46      # len(data.symbols.symbols), src/data/symbols.py:12-26
PAD_IDX, SOS_IDX, EOS_IDX = 0, 1, 2   # src/utils.py:19-21


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def portable_tensor(name, shape, seed=1234, template=None):
    """Deterministic value for one state_dict entry, a function of (seed, name, shape) only."""
    r = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if leaf == "pe":                      # PositionalEncoding buffer: keep the analytic table
        return positional_table(shape[1], shape[2])[None]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_mean":
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    if leaf == "running_var":
        return (1.0 + r.uniform(0.0, 0.5, shape)).astype(np.float32)
    is_norm = ("norm" in name) and leaf in ("weight", "bias")
    if is_norm and leaf == "weight":
        return (1.0 + r.uniform(-0.2, 0.2, shape)).astype(np.float32)
    if len(shape) == 1:
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    if ".rnn.rnn." in name:               # LSTM: U(-1/sqrt(H), 1/sqrt(H)), H = shape[0]/4
        b = 1.0 / np.sqrt(shape[0] / 4.0)
        return r.uniform(-b, b, shape).astype(np.float32)
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    b = np.sqrt(6.0 / (fan_in + fan_out))
    w = r.uniform(-b, b, shape).astype(np.float32)
    if name.endswith("embed.weight"):
        w[PAD_IDX] = 0.0                  # nn.Embedding(padding_idx=0), src/module.py:189
    return w


def portable_state_dict(template, seed=1234):
    """template: mapping name -> array-like with .shape (a state_dict).  Returns name -> np.ndarray."""
    return {k: portable_tensor(k, tuple(v.shape), seed) for k, v in template.items()}


def positional_table(max_len, d_model):
    """sin/cos table exactly as src/module.py:255-262 builds it (float32 arithmetic)."""
    import torch
    import math
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


def synth_batch(B, Tt, Tm, seed=0, ragged=False, num_mels=80):
    """Synthetic batch, SURVEY.md section 8(d).  Returns numpy (text, mel, text_len, mel_len)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    text_len = torch.full((B,), Tt, dtype=torch.int64)
    mel_len = torch.full((B,), Tm, dtype=torch.int64)
    if ragged and B > 1:
        text_len[1:] = torch.randint(max(Tt // 2, 1), Tt + 1, (B - 1,), generator=g)
        mel_len[1:] = torch.randint(max(Tm // 2, 1), Tm + 1, (B - 1,), generator=g)
        text_len, _ = torch.sort(text_len, descending=True)   # collate sorts by text length
    text = torch.randint(3, N_SYMBOLS, (B, Tt), generator=g)
    mel = torch.rand((B, Tm, num_mels), generator=g)
    for b in range(B):
        text[b, text_len[b] - 1] = EOS_IDX
        text[b, text_len[b]:] = PAD_IDX
        mel[b, mel_len[b]:] = 0.0
    return text.numpy(), mel.numpy(), text_len.numpy(), mel_len.numpy()
