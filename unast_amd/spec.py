"""state_dict contract of the reference's transformer UNAST.

Key names, shapes and ORDER are those of `UNAST(TextTransformer, SpeechTransformer, LSTMDiscriminator)
.state_dict()` in the reference (src/network.py:88-276,417-500; src/module.py:76-336), so reference
checkpoints load by name and ours load in the reference.
"""
from collections import OrderedDict

N_SYMBOLS = 46
MAX_LEN = 5000


def _bn(sd, pre, c):
    sd[pre + "weight"] = (c,)
    sd[pre + "bias"] = (c,)
    sd[pre + "running_mean"] = (c,)
    sd[pre + "running_var"] = (c,)
    sd[pre + "num_batches_tracked"] = ()


def _attn(sd, pre, e):
    sd[pre + "in_proj_weight"] = (3 * e, e)
    sd[pre + "in_proj_bias"] = (3 * e,)
    sd[pre + "out_proj.weight"] = (e, e)
    sd[pre + "out_proj.bias"] = (e,)


def _ffn_norms(sd, pre, e, f, n_norm):
    sd[pre + "linear1.weight"] = (f, e)
    sd[pre + "linear1.bias"] = (f,)
    sd[pre + "linear2.weight"] = (e, f)
    sd[pre + "linear2.bias"] = (e,)
    for i in range(1, n_norm + 1):
        sd[pre + "norm%d.weight" % i] = (e,)
        sd[pre + "norm%d.bias" % i] = (e,)


def _enc_dec(sd, pre, L, e, f):
    for i in range(L):
        p = "%sencoder.transformer_encoder.layers.%d." % (pre, i)
        _attn(sd, p + "self_attn.", e)
        _ffn_norms(sd, p, e, f, 2)
    for i in range(L):
        p = "%sdecoder.transformer_decoder.layers.%d." % (pre, i)
        _attn(sd, p + "self_attn.", e)
        _attn(sd, p + "multihead_attn.", e)
        _ffn_norms(sd, p, e, f, 3)


def state_dict_spec(num_layers=4, e=256, ffn=1024, num_mels=80, s_pre_hid=256, t_emb=256,
                    disc_hid=64, disc_layers=2, disc_bidirectional=True, use_discriminator=True):
    """OrderedDict name -> shape, in the reference's state_dict order."""
    sd = OrderedDict()
    sd["text_m.prenet.embed.weight"] = (N_SYMBOLS, t_emb)
    for i, cin in ((1, t_emb), (2, e), (3, e)):
        sd["text_m.prenet.conv%d.conv.weight" % i] = (e, cin, 5)
        sd["text_m.prenet.conv%d.conv.bias" % i] = (e,)
    for i in (1, 2, 3):
        _bn(sd, "text_m.prenet.batch_norm%d." % i, e)
    sd["text_m.pos_emb.pe"] = (1, MAX_LEN, e)
    _enc_dec(sd, "text_m.", num_layers, e, ffn)
    sd["text_m.postnet.fc1.weight"] = (N_SYMBOLS, e)
    sd["text_m.postnet.fc1.bias"] = (N_SYMBOLS,)
    sd["speech_m.prenet.layer.fc1.linear_layer.weight"] = (s_pre_hid, num_mels)
    sd["speech_m.prenet.layer.fc1.linear_layer.bias"] = (s_pre_hid,)
    sd["speech_m.prenet.layer.fc2.linear_layer.weight"] = (e, s_pre_hid)
    sd["speech_m.prenet.layer.fc2.linear_layer.bias"] = (e,)
    sd["speech_m.pos_emb.pe"] = (1, MAX_LEN, e)
    _enc_dec(sd, "speech_m.", num_layers, e, ffn)
    sd["speech_m.postnet.conv1.conv.weight"] = (e, num_mels, 5)
    sd["speech_m.postnet.conv1.conv.bias"] = (e,)
    for i in range(3):
        sd["speech_m.postnet.conv_list.%d.conv.weight" % i] = (e, e, 5)
        sd["speech_m.postnet.conv_list.%d.conv.bias" % i] = (e,)
    sd["speech_m.postnet.conv2.conv.weight"] = (num_mels, e, 5)
    sd["speech_m.postnet.conv2.conv.bias"] = (num_mels,)
    for i in range(3):
        _bn(sd, "speech_m.postnet.batch_norm_list.%d." % i, e)
    _bn(sd, "speech_m.postnet.pre_batchnorm.", e)
    sd["speech_m.postnet.stop_linear.weight"] = (1, e)
    sd["speech_m.postnet.stop_linear.bias"] = (1,)
    sd["speech_m.postnet.linear_project.weight"] = (num_mels, e)
    sd["speech_m.postnet.linear_project.bias"] = (num_mels,)
    if use_discriminator:
        nd = 2 if disc_bidirectional else 1
        for l in range(disc_layers):
            din = e if l == 0 else disc_hid * nd
            for suf in (("", "_reverse") if disc_bidirectional else ("",)):
                sd["discriminator.rnn.rnn.weight_ih_l%d%s" % (l, suf)] = (4 * disc_hid, din)
                sd["discriminator.rnn.rnn.weight_hh_l%d%s" % (l, suf)] = (4 * disc_hid, disc_hid)
                sd["discriminator.rnn.rnn.bias_ih_l%d%s" % (l, suf)] = (4 * disc_hid,)
                sd["discriminator.rnn.rnn.bias_hh_l%d%s" % (l, suf)] = (4 * disc_hid,)
        if disc_bidirectional:
            sd["discriminator.rnn.reduce_h_W.weight"] = (disc_hid, 2 * disc_hid)
            sd["discriminator.rnn.reduce_h_W.bias"] = (disc_hid,)
            sd["discriminator.rnn.reduce_c_W.weight"] = (disc_hid, 2 * disc_hid)
            sd["discriminator.rnn.reduce_c_W.bias"] = (disc_hid,)
        sd["discriminator.fc2.weight"] = (1, disc_hid)
        sd["discriminator.fc2.bias"] = (1,)
    return sd
