"""CNN / DCNN / DRCNN families -- drop-in for libdl/nn_models/basic_cnns.py.

Constructor signatures, defaults, attribute names and ``state_dict`` keys follow
the reference (basic_cnns.py:152, :363); the arithmetic runs in HIP kernels via
multipitch_architectures_amd.ops.
"""
import torch.nn as nn

from .layers import (Conv2d, ConvActPoolDrop, Dropout, LayerNorm, LeakyReLU, LogSoftmax, MaxPool2d, OutputHead, Sigmoid)
from .. import ops


def _head(n_ch_prev, n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout):
    """conv2 (binning to MIDI pitches), conv3 (time reduction), conv4 (chroma reduction): basic_cnns.py:389-408."""
    last_kernel_size = n_bins_in // 3 + 1 - n_bins_out
    conv2 = ConvActPoolDrop(
        Conv2d(n_ch_prev, n_ch[1], kernel_size=(3, 3), padding=(1, 0), stride=(1, 3)),
        LeakyReLU(negative_slope=a_lrelu),
        MaxPool2d(kernel_size=(13, 1), stride=(1, 1), padding=(6, 0)),
        Dropout(p=p_dropout))
    conv3 = ConvActPoolDrop(
        Conv2d(n_ch[1], n_ch[2], kernel_size=(75, 1), padding=(0, 0), stride=(1, 1)),
        LeakyReLU(negative_slope=a_lrelu),
        Dropout(p=p_dropout))
    conv4 = OutputHead(
        Conv2d(n_ch[2], n_ch[3], kernel_size=(1, 1), padding=(0, 0), stride=(1, 1)),
        LeakyReLU(negative_slope=a_lrelu),
        Dropout(p=p_dropout),
        Conv2d(n_ch[3], 1, kernel_size=(1, last_kernel_size), padding=(0, 0), stride=(1, 1)),
        Sigmoid())
    return conv2, conv3, conv4


def _prefilter(n_in, n_out, a_lrelu, p_dropout):
    return ConvActPoolDrop(
        Conv2d(n_in, n_out, kernel_size=(15, 15), padding=(7, 7), stride=(1, 1)),
        LeakyReLU(negative_slope=a_lrelu),
        MaxPool2d(kernel_size=(3, 1), stride=(1, 1), padding=(1, 0)),
        Dropout(p=p_dropout))


class basic_cnn_segm_sigmoid(nn.Module):
    """basic_cnns.py:133-195 -- HCQT (B,6,T>=75,216) -> (B,1,T-74,n_bins_out) pitch activations."""

    def __init__(self, n_chan_input=6, n_chan_layers=[20, 20, 10, 1], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2):
        super().__init__()
        n_in, n_ch = n_chan_input, n_chan_layers
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        self.conv1 = _prefilter(n_in, n_ch[0], a_lrelu, p_dropout)
        self.conv2, self.conv3, self.conv4 = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)

    def forward(self, x):
        x_norm = self.layernorm.forward_cf(x)
        conv1_lrelu = self.conv1(x_norm)
        conv2_lrelu = self.conv2(conv1_lrelu)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred


class deep_cnn_segm_sigmoid(nn.Module):
    """basic_cnns.py:342-423 -- n_prefilt_layers 15x15 prefilter blocks, optional residual adds."""

    def __init__(self, n_chan_input=6, n_chan_layers=[20, 20, 10, 1], n_prefilt_layers=1, residual=False, n_bins_in=216,
                 n_bins_out=12, a_lrelu=0.3, p_dropout=0.2):
        super().__init__()
        n_in, n_ch = n_chan_input, n_chan_layers
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        self.conv1 = _prefilter(n_in, n_ch[0], a_lrelu, p_dropout)
        self.n_prefilt_layers = n_prefilt_layers
        self.prefilt_list = nn.ModuleList()
        for p in range(1, n_prefilt_layers):
            self.prefilt_list.append(_prefilter(n_ch[0], n_ch[0], a_lrelu, p_dropout))
        self.residual = residual
        self.conv2, self.conv3, self.conv4 = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)

    def forward(self, x):
        x_norm = self.layernorm.forward_cf(x)
        x = self.conv1(x_norm)
        for p in range(0, self.n_prefilt_layers - 1):
            prefilt_layer = self.prefilt_list[p]
            if self.residual:
                if prefilt_layer._forward_hooks or prefilt_layer._forward_pre_hooks:
                    x = ops.add(prefilt_layer(x), x)             # a hooked stage returns x_new, as the reference's does
                else:
                    xa, xb = ops.fanout(x)       # two consumers: their gradients are added by an in-tree kernel
                    x = prefilt_layer.forward(xa, residual=xb)   # x_new + x: the add is part of the stage's last kernel
            else:
                x = prefilt_layer(x)
        conv2_lrelu = self.conv2(x)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred


class basic_cnn_pool(nn.Module):
    """basic_cnns.py:68-130 -- the "pool" variant of Zeitler's basic CNN: long max-poolings instead of strided convolutions
    (conv1 15x15 + pool (8,1); conv2 3x3 "same" + pool (3,3); conv3 (3,1); conv4 as everywhere).  HCQT (B,6,75,216) ->
    (B,1,1,n_bins_out).  No experiment script uses it; built from the same blocks as the classes above."""

    def __init__(self, n_chan_input=6, n_chan_layers=[20, 20, 10, 1], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2):
        super().__init__()
        n_in, n_ch = n_chan_input, n_chan_layers
        last_kernel_size = n_bins_in // 3 + 1 - n_bins_out
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        self.conv1 = ConvActPoolDrop(
            Conv2d(n_in, n_ch[0], kernel_size=(15, 15), padding=(7, 7), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            MaxPool2d(kernel_size=(8, 1), stride=(8, 1), padding=(0, 0)),
            Dropout(p=p_dropout))
        self.conv2 = ConvActPoolDrop(
            Conv2d(n_ch[0], n_ch[1], kernel_size=(3, 3), padding=(1, 1), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            MaxPool2d(kernel_size=(3, 3), stride=(3, 3), padding=(0, 0)),
            Dropout(p=p_dropout))
        self.conv3 = ConvActPoolDrop(
            Conv2d(n_ch[1], n_ch[2], kernel_size=(3, 1), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            Dropout(p=p_dropout))
        self.conv4 = OutputHead(
            Conv2d(n_ch[2], n_ch[3], kernel_size=(1, 1), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            Dropout(p=p_dropout),
            Conv2d(n_ch[3], 1, kernel_size=(1, last_kernel_size), padding=(0, 0), stride=(1, 1)),
            Sigmoid())

    def forward(self, x):
        x_norm = self.layernorm.forward_cf(x)
        conv1_lrelu = self.conv1(x_norm)
        conv2_lrelu = self.conv2(conv1_lrelu)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred


class _LogSoftmaxHead(nn.Sequential):
    """conv4 of basic_cnn_segm_logsoftmax: Conv 1x1 + LeakyReLU + Dropout + Conv 1xk + LogSoftmax(dim=1) (basic_cnns.py:249-255)"""

    def forward(self, x):
        conv_a, act, drop, conv_b, lsm = list(self)
        return lsm(conv_b(drop(conv_a(x, act.act, act.slope))))


class basic_cnn_segm_logsoftmax(nn.Module):
    """basic_cnns.py:198-263 -- basic_cnn_segm_sigmoid with n_ch_out output channels and a log-softmax across them instead of
    the sigmoid: (B,6,T>=75,216) -> (B,n_ch_out,T-74,n_bins_out) log-probabilities.  No experiment script uses it."""

    def __init__(self, n_chan_input=6, n_chan_layers=[20, 20, 10, 1], n_ch_out=2, n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2):
        super().__init__()
        n_in, n_ch = n_chan_input, n_chan_layers
        last_kernel_size = n_bins_in // 3 + 1 - n_bins_out
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        self.conv1 = _prefilter(n_in, n_ch[0], a_lrelu, p_dropout)
        self.conv2, self.conv3, _ = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)
        self.conv4 = _LogSoftmaxHead(
            Conv2d(n_ch[2], n_ch[3], kernel_size=(1, 1), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            Dropout(p=p_dropout),
            Conv2d(n_ch[3], n_ch_out, kernel_size=(1, last_kernel_size), padding=(0, 0), stride=(1, 1)),
            LogSoftmax(dim=1))

    def forward(self, x):
        x_norm = self.layernorm.forward_cf(x)
        conv1_lrelu = self.conv1(x_norm)
        conv2_lrelu = self.conv2(conv1_lrelu)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred


class basic_cnn_segm_blank_logsoftmax(nn.Module):
    """basic_cnns.py:267-340 -- as above with an extra "blank" column (MCTC): conv5b spans all 72 bins, its single column is
    stacked in front of conv5a's n_bins_out columns and the log-softmax runs across the n_ch_out channels of the stack:
    (B,6,T>=75,216) -> (B,n_ch_out,T-74,1+n_bins_out).  No experiment script uses it."""

    def __init__(self, n_chan_input=6, n_chan_layers=[20, 20, 10, 1], n_ch_out=2, n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2):
        super().__init__()
        n_in, n_ch = n_chan_input, n_chan_layers
        last_kernel_size = n_bins_in // 3 + 1 - n_bins_out
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        self.conv1 = _prefilter(n_in, n_ch[0], a_lrelu, p_dropout)
        self.conv2, self.conv3, _ = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)
        self.conv4 = ConvActPoolDrop(
            Conv2d(n_ch[2], n_ch[3], kernel_size=(1, 1), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            Dropout(p=p_dropout))
        self.conv5a = Conv2d(n_ch[3], n_ch_out, kernel_size=(1, last_kernel_size), padding=(0, 0), stride=(1, 1))
        self.conv5b = Conv2d(n_ch[3], n_ch_out, kernel_size=(1, 72), padding=(0, 0), stride=(1, 1))
        self.logsoftmax7 = LogSoftmax(dim=1)

    def forward(self, x):
        x_norm = self.layernorm.forward_cf(x)
        conv1_lrelu = self.conv1(x_norm)
        conv2_lrelu = self.conv2(conv1_lrelu)
        conv3_lrelu = self.conv3(conv2_lrelu)
        conv4_lrelu = self.conv4(conv3_lrelu)
        # torch.cat((conv5b(.), conv5a(.)), dim=3) and the log-softmax over dim 1 in one kernel
        y_pred = self.logsoftmax7(self.conv5b(conv4_lrelu), self.conv5a(conv4_lrelu))
        return y_pred


class basic_cnn(nn.Module):
    """basic_cnns.py:5-65 -- Zeitler's basic CNN with strided convolutions: conv1 15x15 + pool (2,1); conv2 3x3 with stride
    (3,3) + pool (2,1); conv3 (6,1); conv4 as everywhere.  HCQT (B,6,75,216) -> (B,1,1,n_bins_out).  No experiment script uses
    it.  (The stride-(3,3) layer's backward-data runs as a 1x1 convolution to 9 phase channels: conv_plan.h: bwd_data_geom.)"""

    def __init__(self, n_chan_input=6, n_chan_layers=[20, 20, 10, 1], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2):
        super().__init__()
        n_in, n_ch = n_chan_input, n_chan_layers
        last_kernel_size = n_bins_in // 3 + 1 - n_bins_out
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        self.conv1 = ConvActPoolDrop(
            Conv2d(n_in, n_ch[0], kernel_size=(15, 15), padding=(7, 7), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            MaxPool2d(kernel_size=(2, 1), stride=(2, 1), padding=(0, 0)),
            Dropout(p=p_dropout))
        self.conv2 = ConvActPoolDrop(
            Conv2d(n_ch[0], n_ch[1], kernel_size=(3, 3), padding=(0, 0), stride=(3, 3)),
            LeakyReLU(negative_slope=a_lrelu),
            MaxPool2d(kernel_size=(2, 1), stride=(2, 1), padding=(0, 0)),
            Dropout(p=p_dropout))
        self.conv3 = ConvActPoolDrop(
            Conv2d(n_ch[1], n_ch[2], kernel_size=(6, 1), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            Dropout(p=p_dropout))
        self.conv4 = OutputHead(
            Conv2d(n_ch[2], n_ch[3], kernel_size=(1, 1), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            Dropout(p=p_dropout),
            Conv2d(n_ch[3], 1, kernel_size=(1, last_kernel_size), padding=(0, 0), stride=(1, 1)),
            Sigmoid())

    def forward(self, x):
        x_norm = self.layernorm.forward_cf(x)
        conv1_lrelu = self.conv1(x_norm)
        conv2_lrelu = self.conv2(conv1_lrelu)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred
