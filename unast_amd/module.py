"""Parameter containers with the reference's class names, constructor signatures, initialisation and state_dict keys
(src/module.py:18-336).  They hold nn.Parameters only; the arithmetic of the hot path is in unast_amd.functional
(HIP kernels), driven from unast_amd.network.  torch.nn container classes (Transformer*Layer, LSTM, BatchNorm1d, ...)
are used purely as named parameter holders so that initial distributions and checkpoint keys equal the reference's —
their torch forward() is never called."""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

N_SYMBOLS = 46   # len(data.symbols.symbols), src/data/symbols.py:26


def _no_forward(self, *a, **k):
    raise RuntimeError("%s is a parameter container in unast_amd; run it through TextTransformer / SpeechTransformer / "
                       "LSTMDiscriminator, whose forward passes are HIP kernels" % type(self).__name__)


class Linear(nn.Module):
    """src/module.py:18-39."""

    def __init__(self, in_dim, out_dim, bias=True, w_init='linear'):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init))
    forward = _no_forward


class Conv(nn.Module):
    """src/module.py:42-73."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=True, w_init='linear'):
        super().__init__()
        if kernel_size != 5 or stride != 1 or dilation != 1:
            raise NotImplementedError("the HIP implicit-GEMM conv supports kernel_size=5, stride=1, dilation=1 (all the hot path uses)")
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init))
    forward = _no_forward


class SpeechPrenet(nn.Module):
    """src/module.py:76-110 (note the duplicate 'dropout2' key: ONE dropout, after fc1)."""

    def __init__(self, num_mels, hidden_size, output_size, p=0.5):
        super().__init__()
        self.input_size, self.output_size, self.hidden_size, self.p = num_mels, output_size, hidden_size, p
        self.layer = nn.Sequential(OrderedDict([
            ('fc1', Linear(self.input_size, self.hidden_size)),
            ('fc2', Linear(self.hidden_size, self.output_size)),
        ]))
    forward = _no_forward


class SpeechPostnet(nn.Module):
    """src/module.py:113-171."""

    def __init__(self, num_mels, num_hidden, p=0.1):
        super().__init__()
        self.num_mels, self.p = num_mels, p
        self.conv1 = Conv(in_channels=num_mels, out_channels=num_hidden, kernel_size=5, padding=4, w_init='tanh')
        self.conv_list = nn.ModuleList([Conv(in_channels=num_hidden, out_channels=num_hidden, kernel_size=5, padding=4, w_init='tanh')
                                        for _ in range(3)])
        self.conv2 = Conv(in_channels=num_hidden, out_channels=num_mels, kernel_size=5, padding=4)
        self.batch_norm_list = nn.ModuleList([nn.BatchNorm1d(num_hidden) for _ in range(3)])
        self.pre_batchnorm = nn.BatchNorm1d(num_hidden)
        self.stop_linear = nn.Linear(num_hidden, 1)
        self.linear_project = nn.Linear(num_hidden, num_mels)
    forward = _no_forward


class TextPrenet(nn.Module):
    """src/module.py:174-230."""

    def __init__(self, embedding_size, num_hidden, p=0.5):
        super().__init__()
        self.embedding_size, self.p = embedding_size, p
        self.embed = nn.Embedding(N_SYMBOLS, embedding_size, padding_idx=0)
        pad = int(np.floor(5 / 2))
        self.conv1 = Conv(in_channels=embedding_size, out_channels=num_hidden, kernel_size=5, padding=pad, w_init='relu')
        self.conv2 = Conv(in_channels=num_hidden, out_channels=num_hidden, kernel_size=5, padding=pad, w_init='relu')
        self.conv3 = Conv(in_channels=num_hidden, out_channels=num_hidden, kernel_size=5, padding=pad, w_init='relu')
        self.batch_norm1 = nn.BatchNorm1d(num_hidden)
        self.batch_norm2 = nn.BatchNorm1d(num_hidden)
        self.batch_norm3 = nn.BatchNorm1d(num_hidden)
    forward = _no_forward


class TextPostnet(nn.Module):
    """src/module.py:233-246."""

    def __init__(self, hidden, p=.2):
        super().__init__()
        self.fc1 = nn.Linear(hidden, N_SYMBOLS)
        self.p = p
    forward = _no_forward


class PositionalEncoding(nn.Module):
    """src/module.py:249-267 (dropout fixed at the ctor default 0.1 — the reference never passes another value)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.p = dropout
        self.d_model_scale = math.sqrt(d_model)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))
    forward = _no_forward


class TransformerEncoder(nn.Module):
    """src/module.py:270-280."""

    def __init__(self, ninp, nhead, nhid, dropout, nlayers):
        super().__init__()
        if ninp // nhead != 64:
            raise NotImplementedError("the HIP attention kernels support head_dim 64 (ninp/nhead); got %d" % (ninp // nhead))
        layer = torch.nn.TransformerEncoderLayer(ninp, nhead, nhid, dropout)
        self.transformer_encoder = torch.nn.TransformerEncoder(layer, nlayers, enable_nested_tensor=False)
        self.p, self.nlayers, self.nhead = dropout, nlayers, nhead
    forward = _no_forward


class TransformerDecoder(nn.Module):
    """src/module.py:283-293."""

    def __init__(self, ninp, nhead, ffn_dim, dropout, nlayers):
        super().__init__()
        layer = torch.nn.TransformerDecoderLayer(ninp, nhead, ffn_dim, dropout)
        self.transformer_decoder = torch.nn.TransformerDecoder(layer, nlayers)
        self.p, self.nlayers, self.nhead = dropout, nlayers, nhead
    forward = _no_forward


class RNNEncoder(nn.Module):
    """src/module.py:297-336 (parameter container for the LSTM discriminator)."""

    def __init__(self, d_in, hidden, dropout=.2, num_layers=1, bidirectional=False):
        super().__init__()
        self.hidden, self.num_layers, self.num_dir = hidden, num_layers, 2 if bidirectional else 1
        self.rnn = nn.LSTM(d_in, hidden, num_layers=num_layers, bidirectional=bidirectional, batch_first=True, dropout=dropout if num_layers > 1 else 0.0)
        if self.num_dir == 2:
            self.reduce_h_W = nn.Linear(hidden * 2, hidden, bias=True)
            self.reduce_c_W = nn.Linear(hidden * 2, hidden, bias=True)
    forward = _no_forward
