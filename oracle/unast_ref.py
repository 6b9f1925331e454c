"""ORACLE — test infrastructure only.  CPU fp32 restatement of UNAST's train-step hot path.

This file is the CHECKER for unast_amd's HIP path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product package (unast_amd/) never does.

It restates, in batch-first `lens`-masked form with primitive torch CPU ops, what the reference
computes through torch.nn modules.  Parity is PINNED: tests/test_oracle_golden.py checks every
function below against golden vectors produced by the reference itself (tools/gen_golden.py
imports /root/reference/src and runs its own train_ae_step / train_sp_step /
train_discriminator_step / optimizer_step; fixtures in tests/golden/).

Reference citations (file:line into /root/reference):
  sent_lens_to_mask            src/utils.py:77-83          -> lens_mask
  PositionalEncoding           src/module.py:249-267       -> pos_enc
  TextPrenet / forward_fcn     src/module.py:174-230       -> text_prenet_convs
  SpeechPrenet                 src/module.py:76-110        -> speech_prenet
  SpeechPostnet                src/module.py:113-171       -> speech_postnet, mel_and_stop
  TextPostnet                  src/module.py:233-246       -> (inline linear)
  TransformerEncoder/Decoder   src/module.py:270-293 (torch.nn.Transformer*) -> encoder_stack, decoder_stack
  RNNEncoder/LSTMDiscriminator src/module.py:297-336, src/network.py:172-186 -> lstm_discriminator
  Text/SpeechTransformer       src/network.py:188-276, 417-500 -> text_encode, text_decode_sequence, speech_*
  UNAST.text_ae/speech_ae/tts/asr  src/network.py:97-145
  losses                       src/train.py:100-122, 147-164
  discriminator_shuffle_batch  src/train.py:296-329 (RNG off: identity permutation)
  autoencoder/supervised/discriminator_step, train_*_step  src/train.py:199-259, 337-463
  optimizer_step (clip + AdamW)    src/train.py:358-363, 929-932
"""
import math

import torch

PAD_IDX, SOS_IDX, EOS_IDX = 0, 1, 2
LN_EPS = 1e-5
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# Optional emulation of the GPU's MFMA operand rounding, to size tolerances on the CPU:
#   None     exact fp32 products (the oracle proper)
#   "bf16"   both operands rounded to bf16, fp32 accumulate
#   "bf16x3" split-bf16: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
MATMUL_EMU = None


def _split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def mm(a, b):
    """a @ b with optional operand-rounding emulation."""
    if MATMUL_EMU is None:
        return a @ b
    if MATMUL_EMU == "bf16":
        return a.to(torch.bfloat16).float() @ b.to(torch.bfloat16).float()
    ah, al = _split(a)
    bh, bl = _split(b)
    return ah @ bh + ah @ bl + al @ bh


def linear(x, w, b=None):
    y = mm(x, w.t())
    return y if b is None else y + b


def lens_mask(lens, T):
    """mask[b,t] = t < lens[b]  (src/utils.py:77-83)."""
    return torch.arange(T)[None, :] < lens[:, None]


def pos_enc(x, pe):
    """x*sqrt(d) + pe[:T]; dropout off (src/module.py:265-267)."""
    return x * math.sqrt(x.shape[-1]) + pe[:, : x.shape[1]]


def layer_norm(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def batch_norm_train(x, w, b, stats=None):
    """x [B,T,C]; statistics over all B*T positions, pads included.
    stats: optional dict(running_mean, running_var) updated in place (momentum .1, unbiased var)."""
    n = x.shape[0] * x.shape[1]
    mu = x.mean((0, 1))
    var = ((x - mu) ** 2).mean((0, 1))
    if stats is not None:
        with torch.no_grad():
            stats["running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mu.detach())
            stats["running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
    return (x - mu) / torch.sqrt(var + BN_EPS) * w + b


def conv1d_k5(x, w, b, pad_left):
    """x [B,T,Cin], w [Cout,Cin,5] (torch layout). out[b,t] = sum_j x[b,t+j-pad_left] w[:,:,j] + b.
    pad_left=2: 'same' conv (TextPrenet); pad_left=4: causal conv (SpeechPostnet: pad 4, drop last 4)."""
    B, T, C = x.shape
    xp = torch.nn.functional.pad(x, (0, 0, pad_left, 4 - pad_left))
    cols = torch.cat([xp[:, j: j + T] for j in range(5)], dim=-1)          # [B,T,5*Cin], (j,c) order
    wm = w.permute(0, 2, 1).reshape(w.shape[0], -1)                        # [Cout, 5*Cin]
    return linear(cols, wm, b)


def mha(xq, xkv, P, pre, nhead, lens_k, causal):
    """torch.nn.MultiheadAttention semantics, dropout off.
    xq [B,Tq,E], xkv [B,Tk,E]; keys >= lens_k[b] masked; causal masks tk > tq. Padded queries not masked."""
    W, bias = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    E = xq.shape[-1]
    hd = E // nhead
    q = linear(xq, W[:E], bias[:E])
    k = linear(xkv, W[E:2 * E], bias[E:2 * E])
    v = linear(xkv, W[2 * E:], bias[2 * E:])
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    q = q.view(B, Tq, nhead, hd).transpose(1, 2) / math.sqrt(hd)
    k = k.view(B, Tk, nhead, hd).transpose(1, 2)
    v = v.view(B, Tk, nhead, hd).transpose(1, 2)
    s = mm(q, k.transpose(-1, -2))                                          # [B,H,Tq,Tk]
    neg = ~lens_mask(lens_k, Tk)[:, None, None, :]
    if causal:
        neg = neg | (torch.arange(Tk)[None, :] > torch.arange(Tq)[:, None])[None, None]
    s = s.masked_fill(neg, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = mm(p, v).transpose(1, 2).reshape(B, Tq, E)
    return linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


def ffn(x, P, pre):
    h = torch.relu(linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"]))
    return linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"])


def encoder_stack(x, lens, P, pre, L, nhead):
    """L post-LN encoder layers. pre e.g. 'text_m.encoder.transformer_encoder.layers.'"""
    for i in range(L):
        p = "%s%d." % (pre, i)
        x = layer_norm(x + mha(x, x, P, p + "self_attn.", nhead, lens, False), P[p + "norm1.weight"], P[p + "norm1.bias"])
        x = layer_norm(x + ffn(x, P, p), P[p + "norm2.weight"], P[p + "norm2.bias"])
    return x


def decoder_stack(x, lens_q, mem, lens_k, P, pre, L, nhead):
    for i in range(L):
        p = "%s%d." % (pre, i)
        x = layer_norm(x + mha(x, x, P, p + "self_attn.", nhead, lens_q, True), P[p + "norm1.weight"], P[p + "norm1.bias"])
        x = layer_norm(x + mha(x, mem, P, p + "multihead_attn.", nhead, lens_k, False), P[p + "norm2.weight"], P[p + "norm2.bias"])
        x = layer_norm(x + ffn(x, P, p), P[p + "norm3.weight"], P[p + "norm3.bias"])
    return x


class Model:
    """Functional view over a reference-layout state_dict."""

    def __init__(self, state_dict, num_layers=4, nhead=4, requires_grad=True):
        self.P = {}
        self.buf = {}
        for k, v in state_dict.items():
            t = torch.as_tensor(v).clone()
            if t.dtype.is_floating_point and "running_" not in k and not k.endswith(".pe"):
                self.P[k] = t.requires_grad_(requires_grad)
            else:
                self.buf[k] = t
        self.L = num_layers
        self.nhead = nhead
        self.update_bn = True

    # -- helpers ----------------------------------------------------------------------------
    def _bn(self, x, pre):
        st = None
        if self.update_bn:
            st = {"running_mean": self.buf[pre + "running_mean"], "running_var": self.buf[pre + "running_var"]}
        return batch_norm_train(x, self.P[pre + "weight"], self.P[pre + "bias"], st)

    def _embed(self, ids):
        """nn.Embedding(padding_idx=0): row 0 is used as stored but receives no gradient (src/module.py:189)."""
        W = self.P["text_m.prenet.embed.weight"]
        return torch.cat([W[:1].detach(), W[1:]], dim=0)[ids]

    def param_names(self, prefix=""):
        return [k for k in self.P if k.startswith(prefix)]

    # -- text side (src/network.py:417-500) ---------------------------------------------------
    def text_encode(self, text, text_len):
        P = self.P
        x = self._embed(text)
        for i in (1, 2, 3):
            x = conv1d_k5(x, P["text_m.prenet.conv%d.conv.weight" % i], P["text_m.prenet.conv%d.conv.bias" % i], 2)
            x = torch.relu(self._bn(x, "text_m.prenet.batch_norm%d." % i))
        x = pos_enc(x, self.buf["text_m.pos_emb.pe"])
        return encoder_stack(x, text_len, P, "text_m.encoder.transformer_encoder.layers.", self.L, self.nhead)

    def text_decode_sequence(self, text, text_len, mem, mem_len):
        P = self.P
        sos = torch.full((text.shape[0], 1), SOS_IDX, dtype=text.dtype)
        tgt = torch.cat([sos, text[:, :-1]], dim=1)
        x = pos_enc(self._embed(tgt), self.buf["text_m.pos_emb.pe"])   # no convs (src/network.py:435-438)
        x = decoder_stack(x, text_len, mem, mem_len, P, "text_m.decoder.transformer_decoder.layers.", self.L, self.nhead)
        return linear(x, P["text_m.postnet.fc1.weight"], P["text_m.postnet.fc1.bias"])

    # -- speech side (src/network.py:188-276) -----------------------------------------------
    def speech_prenet(self, mel):
        P = self.P
        h = torch.relu(linear(mel, P["speech_m.prenet.layer.fc1.linear_layer.weight"], P["speech_m.prenet.layer.fc1.linear_layer.bias"]))
        return torch.relu(linear(h, P["speech_m.prenet.layer.fc2.linear_layer.weight"], P["speech_m.prenet.layer.fc2.linear_layer.bias"]))

    def speech_encode(self, mel, mel_len):
        x = pos_enc(self.speech_prenet(mel), self.buf["speech_m.pos_emb.pe"])
        return encoder_stack(x, mel_len, self.P, "speech_m.encoder.transformer_encoder.layers.", self.L, self.nhead)

    def speech_postnet(self, x):
        P = self.P
        x = conv1d_k5(x, P["speech_m.postnet.conv1.conv.weight"], P["speech_m.postnet.conv1.conv.bias"], 4)
        x = torch.tanh(self._bn(x, "speech_m.postnet.pre_batchnorm."))
        for i in range(3):
            x = conv1d_k5(x, P["speech_m.postnet.conv_list.%d.conv.weight" % i], P["speech_m.postnet.conv_list.%d.conv.bias" % i], 4)
            x = torch.tanh(self._bn(x, "speech_m.postnet.batch_norm_list.%d." % i))
        return conv1d_k5(x, P["speech_m.postnet.conv2.conv.weight"], P["speech_m.postnet.conv2.conv.bias"], 4)

    def speech_decode_sequence(self, mel, mel_len, mem, mem_len):
        P = self.P
        tgt = torch.cat([torch.zeros_like(mel[:, :1]), mel[:, :-1]], dim=1)
        x = pos_enc(self.speech_prenet(tgt), self.buf["speech_m.pos_emb.pe"])
        x = decoder_stack(x, mel_len, mem, mem_len, P, "speech_m.decoder.transformer_decoder.layers.", self.L, self.nhead)
        pre = linear(x, P["speech_m.postnet.linear_project.weight"], P["speech_m.postnet.linear_project.bias"])
        stop = linear(x, P["speech_m.postnet.stop_linear.weight"], P["speech_m.postnet.stop_linear.bias"]).squeeze(-1)
        return pre, pre + self.speech_postnet(pre), stop

    # -- UNAST task graph (src/network.py:97-145) ----------------------------------------------
    def text_ae(self, text, text_len):
        enc = self.text_encode(text, text_len)
        return self.text_decode_sequence(text, text_len, enc, text_len), enc

    def speech_ae(self, mel, mel_len):
        enc = self.speech_encode(mel, mel_len)
        return self.speech_decode_sequence(mel, mel_len, enc, mel_len) + (enc,)

    def tts(self, text, text_len, mel, mel_len):
        enc = self.text_encode(text, text_len)
        return self.speech_decode_sequence(mel, mel_len, enc, text_len) + (enc,)

    def asr(self, text, text_len, mel, mel_len):
        enc = self.speech_encode(mel, mel_len)
        return self.text_decode_sequence(text, text_len, enc, mel_len), enc

    # -- discriminator (src/network.py:172-186, src/module.py:297-336) ------------------------
    def lstm_discriminator(self, x, lens, hid=64, layers=2):
        """Packed bi-LSTM: only steps t < lens[b] are processed; gate order i,f,g,o; the reverse
        direction starts at t = lens[b]-1.  Output: fc2(leaky_relu(reduce_h_W([h_fwd,h_bwd]) of top layer)).
        With self.packed_lstm = True the recurrence runs through torch's own packed-sequence LSTM (what the reference
        calls, src/module.py:306,315-316) instead of the explicit time loop below; tests pin the two against each other."""
        if getattr(self, "packed_lstm", False):
            return self._lstm_discriminator_packed(x, lens, hid, layers)
        P = self.P
        B, T, _ = x.shape
        valid = lens_mask(lens, T).float().unsqueeze(-1)
        inp = x
        finals = None
        for l in range(layers):
            outs, fin = [], []
            for suffix, rev in (("", False), ("_reverse", True)):
                wih = P["discriminator.rnn.rnn.weight_ih_l%d%s" % (l, suffix)]
                whh = P["discriminator.rnn.rnn.weight_hh_l%d%s" % (l, suffix)]
                bsum = P["discriminator.rnn.rnn.bias_ih_l%d%s" % (l, suffix)] + P["discriminator.rnn.rnn.bias_hh_l%d%s" % (l, suffix)]
                xp = linear(inp, wih, bsum)
                h = torch.zeros(B, hid)
                c = torch.zeros(B, hid)
                hs = [None] * T
                for t in (range(T - 1, -1, -1) if rev else range(T)):
                    g = xp[:, t] + mm(h, whh.t())
                    i, f, gg, o = g.chunk(4, dim=-1)
                    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                    h_new = torch.sigmoid(o) * torch.tanh(c_new)
                    m = valid[:, t]
                    c = m * c_new + (1 - m) * c
                    h = m * h_new + (1 - m) * h
                    hs[t] = h * m
                outs.append(torch.stack(hs, dim=1))
                fin.append(h)
            inp = torch.cat(outs, dim=-1)
            finals = fin
        hcat = torch.cat(finals, dim=-1)
        new_h = linear(hcat, P["discriminator.rnn.reduce_h_W.weight"], P["discriminator.rnn.reduce_h_W.bias"])
        a = torch.nn.functional.leaky_relu(new_h, 0.2)
        return linear(a, P["discriminator.fc2.weight"], P["discriminator.fc2.bias"]).squeeze(-1)


def _lstm_discriminator_packed(self, x, lens, hid=64, layers=2):
    P = self.P
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens.cpu(), batch_first=True, enforce_sorted=False)
    flat = []
    for l in range(layers):
        for suf in ("", "_reverse"):
            flat += [P["discriminator.rnn.rnn.%s_l%d%s" % (k, l, suf)] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    B = x.shape[0]
    h0 = torch.zeros(layers * 2, B, hid)
    out, hn, cn = torch._VF.lstm(packed.data, packed.batch_sizes, (h0, h0.clone()), flat, True, layers, 0.0, False, True)
    hn = hn[:, packed.unsorted_indices] if packed.unsorted_indices is not None else hn
    h = hn.view(layers, 2, B, hid)
    hcat = torch.cat((h[-1, 0], h[-1, 1]), dim=-1)
    new_h = linear(hcat, P["discriminator.rnn.reduce_h_W.weight"], P["discriminator.rnn.reduce_h_W.bias"])
    a = torch.nn.functional.leaky_relu(new_h, 0.2)
    return linear(a, P["discriminator.fc2.weight"], P["discriminator.fc2.bias"]).squeeze(-1)


Model._lstm_discriminator_packed = _lstm_discriminator_packed


# ---- losses (src/train.py:100-122, 147-164) ---------------------------------------------------
def masked_mse(gold, pred, mask):
    return (((gold - pred) ** 2) * mask).sum() / mask.sum()


def speech_loss(gold_mel, gold_stop, pre, post, mel_len, stop, eos_weight):
    mask = lens_mask(mel_len, pre.shape[1]).unsqueeze(-1).expand_as(pre).float()
    pw = torch.where(gold_stop == 1, torch.tensor(float(eos_weight)), torch.tensor(1.0))
    lw = 1 + (pw - 1) * gold_stop
    bce = (1 - gold_stop) * stop + lw * (torch.log1p(torch.exp(-stop.abs())) + torch.clamp(-stop, min=0))
    return masked_mse(gold_mel, pre, mask) + masked_mse(gold_mel, post, mask) + bce.mean()


def text_loss(gold, logits, eos_weight):
    """logits [B,T,V]; weighted CE with ignore_index=PAD (F.cross_entropy semantics)."""
    V = logits.shape[-1]
    w = torch.ones(V)
    w[EOS_IDX] = eos_weight
    lse = torch.logsumexp(logits, dim=-1)
    nll = lse - logits.gather(-1, gold.unsqueeze(-1)).squeeze(-1)
    wy = w[gold] * (gold != PAD_IDX).float()
    return (nll * wy).sum() / wy.sum()


def bce_logits_mean(x, y):
    return ((1 - y) * x + torch.log1p(torch.exp(-x.abs())) + torch.clamp(-x, min=0)).mean()


def discriminator_batch(t_hid, t_len, s_hid, s_len, train_discriminator):
    """src/train.py:296-329 with the permutation fixed to identity (RNG off)."""
    Tmax = max(t_hid.shape[1], s_hid.shape[1])
    pad = lambda h: torch.nn.functional.pad(h, (0, 0, 0, Tmax - h.shape[1]))
    d_hid = torch.cat([pad(t_hid), pad(s_hid)], dim=0)
    d_len = torch.cat([t_len, s_len])
    tgt = torch.cat([torch.full((t_hid.shape[0],), 0.9), torch.full((s_hid.shape[0],), 0.1)])
    if not train_discriminator:
        tgt = 1 - tgt
    return d_hid, d_len, tgt


# ---- train-step surface (src/train.py:199-259, 337-463, 602-638) ------------------------------
def generator_losses(model, batch, s_eos_weight=5.0, t_eos_weight=1.0, use_discriminator=True):
    """AE + SP sub-steps; returns dict of the six loss scalars (autograd-connected)."""
    text, mel, text_len, mel_len = batch
    gold_stop = torch.nn.functional.one_hot(mel_len - 1, mel.shape[1]).float()
    out = {}
    logits, t_enc = model.text_ae(text, text_len)
    pre, post, stop, s_enc = model.speech_ae(mel, mel_len)
    out["t_ae"] = text_loss(text, logits, t_eos_weight)
    out["s_ae"] = speech_loss(mel, gold_stop, pre, post, mel_len, stop, s_eos_weight)
    if use_discriminator:
        dh, dl, dt = discriminator_batch(t_enc, text_len, s_enc, mel_len, False)
        out["d_ae"] = bce_logits_mean(model.lstm_discriminator(dh, dl), dt)
    out["_ae_out"] = (logits, pre, post, stop, t_enc, s_enc)
    return out


def supervised_losses(model, batch, s_eos_weight=5.0, t_eos_weight=1.0, use_discriminator=True):
    text, mel, text_len, mel_len = batch
    gold_stop = torch.nn.functional.one_hot(mel_len - 1, mel.shape[1]).float()
    out = {}
    pre, post, stop, t_enc = model.tts(text, text_len, mel, mel_len)
    logits, s_enc = model.asr(text, text_len, mel, mel_len)          # specaugment off (RNG off)
    out["tts_"] = speech_loss(mel, gold_stop, pre, post, mel_len, stop, s_eos_weight)
    out["asr_"] = text_loss(text, logits, t_eos_weight)
    if use_discriminator:
        dh, dl, dt = discriminator_batch(t_enc, text_len, s_enc, mel_len, False)
        out["sp_d"] = bce_logits_mean(model.lstm_discriminator(dh, dl), dt)
    return out


def discriminator_loss_step(model, batch):
    text, mel, text_len, mel_len = batch
    with torch.no_grad():
        t_enc = model.text_encode(text, text_len)
        s_enc = model.speech_encode(mel, mel_len)
    dh, dl, dt = discriminator_batch(t_enc, text_len, s_enc, mel_len, True)
    return bce_logits_mean(model.lstm_discriminator(dh, dl), dt)


class AdamW:
    """clip_grad_norm_ + torch.optim.AdamW restated (src/train.py:358-363, 929-932).
    Parameters whose grad is None are skipped entirely (no decay, no moment update)."""

    def __init__(self, params, lr, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8):
        self.params = params          # dict name -> tensor
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.state = {}

    def step(self, grad_clip=1.0):
        ps = [(n, p) for n, p in self.params.items() if p.grad is not None]
        total = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in ps)).float()
        if grad_clip > 0:
            coef = torch.clamp(grad_clip / (total + 1e-6), max=1.0)
            for _, p in ps:
                p.grad.mul_(coef)
        b1, b2 = self.betas
        with torch.no_grad():
            for n, p in ps:
                st = self.state.setdefault(n, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
                st["step"] += 1
                g = p.grad
                p.mul_(1 - self.lr * self.wd)
                st["m"].mul_(b1).add_(g, alpha=1 - b1)
                st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1 = 1 - b1 ** st["step"]
                bc2 = 1 - b2 ** st["step"]
                denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(st["m"], denom, value=-self.lr / bc1)
        for _, p in self.params.items():
            p.grad = None
        return total


def transformer_schedule(step, warmup):
    """LambdaLR factor of get_transformer_paper_schedule (src/train.py:886-907)."""
    if step < warmup:
        return float(step) / max(1.0, float(warmup) ** 1.5)
    return 1.0 / max(1.0, float(step) ** 0.5)


def linear_schedule(step, warmup, total):
    """src/train.py:859-884."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def full_step(model, opt, batch, grad_clip=1.0, use_discriminator=True):
    """One outer train step with ae_steps=sp_steps=d_steps=1, cm_steps=0 (src/train.py:602-638).
    Returns dict of loss floats + grad norms."""
    rec = {}
    disc = [p for n, p in model.P.items() if n.startswith("discriminator.")]
    for p in disc:
        p.requires_grad_(False)
    ae = generator_losses(model, batch, use_discriminator=use_discriminator)
    ae.pop("_ae_out")
    (sum(ae.values()) / 2).backward()
    sp = supervised_losses(model, batch, use_discriminator=use_discriminator)
    (sum(sp.values()) / 2).backward()
    rec.update({k: v.item() for k, v in ae.items()})
    rec.update({k: v.item() for k, v in sp.items()})
    rec["gen_grad_norm"] = opt.step(grad_clip).item()
    if use_discriminator:
        for p in disc:
            p.requires_grad_(True)
        d = discriminator_loss_step(model, batch)
        d.backward()
        rec["d"] = d.item()
        rec["d_grad_norm"] = opt.step(grad_clip).item()
    return rec
