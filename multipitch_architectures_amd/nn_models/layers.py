"""Parameter-holding building blocks with torch.nn-compatible ``state_dict`` keys.

Each class mirrors the torch.nn module the reference instantiates at the cited
line, keeps its parameter names / shapes / default initialisation, and runs its
arithmetic through multipitch_architectures_amd.ops (HIP kernels).  Marker
modules (``ReLU``, ``LeakyReLU``, ``Dropout``, ``Sigmoid``) only occupy the
``nn.Sequential`` slot of the reference so that child indices -- and therefore
checkpoint keys such as ``inc.double_conv.4.weight`` -- match.
"""
import math

import torch
import torch.nn as nn

from .. import ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2d(nn.Module):
    """nn.Conv2d(in, out, kernel_size, stride, padding) -- zero padding, bias, cross-correlation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=(1, 1), padding=(0, 0)):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, act, slope)

    def forward_stats(self, x):
        """(y, partial sums of y and y^2 per pixel tile and channel) for the BatchNorm2d that follows"""
        return ops.conv2d_stats(x, self.weight, self.bias, self.stride, self.padding)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}"


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(C) defaults (eps 1e-5, momentum 0.1, affine, track_running_stats)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, relu=False, partials=None):
        if x.shape[1] != self.num_features:
            raise RuntimeError(f"BatchNorm2d: expected {self.num_features} channels, got {x.shape[1]}")
        return ops.batchnorm_relu(x, self.weight, self.bias, self.running_mean, self.running_var,
                                  self.num_batches_tracked, self.training, self.momentum, relu, partials)


class LayerNorm(nn.Module):
    """nn.LayerNorm(normalized_shape): [E] over rows, or [C,F] applied on x.transpose(1,2) (see model forwards)."""

    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        self.normalized_shape = tuple(normalized_shape) if not isinstance(normalized_shape, int) else (normalized_shape,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(self.normalized_shape))

    def forward_cf(self, x):
        """x (B,C,T,F): equals self(x.transpose(1,2)).transpose(1,2) of the reference (unet_cnns.py:560)."""
        if tuple(x.shape[1:2] + x.shape[3:4]) != self.normalized_shape:
            raise RuntimeError(f"LayerNorm: expected (C,F)={self.normalized_shape}, got input {tuple(x.shape)}")
        return ops.layernorm_cf(x, self.weight, self.bias)

    def forward(self, x, residual=None):
        return ops.layernorm_rows(x, residual, self.weight, self.bias)


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act=ops.ACT_NONE):
        return ops.linear(x, self.weight, self.bias, act)


class MultiheadAttention(nn.Module):
    """nn.MultiheadAttention(embed_dim, num_heads) with batch_first=False: fed (B,S,E) tensors the reference
    attends over dim 0 = the batch (unet_cnns.py:134,153)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        if embed_dim % num_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, q, k, v):
        o = ops.mha_batchaxis(q, k, v, self.in_proj_weight, self.in_proj_bias, self.num_heads)
        return (self.out_proj(o), None)


class LSTM(nn.Module):
    """nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True)."""

    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True, bidirectional=True):
        super().__init__()
        assert batch_first and bidirectional, "only the configuration the reference uses (unet_cnns.py:232)"
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        k = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            isz = input_size if layer == 0 else 2 * hidden_size
            for suffix in ("", "_reverse"):
                for name, shape in ((f"weight_ih_l{layer}{suffix}", (4 * hidden_size, isz)),
                                    (f"weight_hh_l{layer}{suffix}", (4 * hidden_size, hidden_size)),
                                    (f"bias_ih_l{layer}{suffix}", (4 * hidden_size,)),
                                    (f"bias_hh_l{layer}{suffix}", (4 * hidden_size,))):
                    p = nn.Parameter(torch.empty(shape))
                    nn.init.uniform_(p, -k, k)
                    self.register_parameter(name, p)

    def forward(self, x):
        if x.shape[-1] != self.input_size:
            raise RuntimeError(f"input.size(-1) must be equal to input_size. Expected {self.input_size}, got {x.shape[-1]}")
        h = x
        for layer in range(self.num_layers):
            params = []
            for suffix in ("", "_reverse"):
                params += [getattr(self, f"{n}_l{layer}{suffix}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            h = ops.blstm_layer(h, params)
        return (h, None)


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size, stride=None, padding=(0, 0), return_indices=False):
        super().__init__()
        self.kernel_size = _pair(kernel_size)
        self.stride = self.kernel_size if stride is None else _pair(stride)
        self.padding = _pair(padding)
        self.return_indices = bool(return_indices)
        if self.return_indices and (self.stride != self.kernel_size or self.padding != (0, 0)):
            raise NotImplementedError("return_indices is built for non-overlapping windows (stride == kernel, no padding)")

    def forward(self, x):
        if self.return_indices:
            return ops.max_pool2d_with_indices(x, self.kernel_size)
        return ops.max_pool2d(x, self.kernel_size, self.stride, self.padding)


class MaxUnpool2d(nn.Module):
    """nn.MaxUnpool2d(kernel_size) (stride == kernel): unet_cnns.py:1751-1765"""

    def __init__(self, kernel_size, stride=None, padding=(0, 0)):
        super().__init__()
        self.kernel_size = _pair(kernel_size)
        if (stride is not None and _pair(stride) != self.kernel_size) or _pair(padding) != (0, 0):
            raise NotImplementedError("MaxUnpool2d is built for stride == kernel_size, no padding")

    def forward(self, x, indices):
        return ops.max_unpool2d(x, indices, self.kernel_size)


class _Pointwise(nn.Module):
    act, slope = ops.ACT_NONE, 0.0

    def forward(self, x):
        return ops.activation(x, self.act, self.slope)


class ReLU(_Pointwise):
    act = ops.ACT_RELU

    def __init__(self, inplace=False):
        super().__init__()


class LeakyReLU(_Pointwise):
    act = ops.ACT_LRELU

    def __init__(self, negative_slope=0.01):
        super().__init__()
        self.slope = self.negative_slope = negative_slope


class Sigmoid(_Pointwise):
    act = ops.ACT_SIGMOID


class LogSoftmax(nn.Module):
    """nn.LogSoftmax(dim=1) of a (B,C,R,W) tensor (basic_cnns.py:255, 331) -- only the channel axis is built."""

    def __init__(self, dim=1):
        super().__init__()
        if dim != 1:
            raise NotImplementedError("only nn.LogSoftmax(dim=1), the one the reference uses, is built")
        self.dim = dim

    def forward(self, x, other=None):
        return ops.logsoftmax_cat(x, other)


class SELU(_Pointwise):
    """nn.SELU -- the frequency U-Nets (unet_cnns.py:1711-1765)"""
    act = ops.ACT_SELU

    def __init__(self, inplace=False):
        super().__init__()


class ELU(_Pointwise):
    """nn.ELU(alpha=1.0) -- double_conv's alt_order branch (unet_cnns.py:60-70)"""
    act = ops.ACT_ELU

    def __init__(self, alpha=1.0, inplace=False):
        super().__init__()
        if alpha != 1.0:
            raise NotImplementedError("only nn.ELU(alpha=1.0), the value the reference uses, is built")
        self.alpha = alpha


class Dropout(nn.Module):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class ConvActPoolDrop(nn.Sequential):
    """The reference's ``nn.Sequential(Conv2d, LeakyReLU[, MaxPool2d], Dropout)`` stages (conv1/prefilt/conv2/conv3;
    basic_cnns.py:371-401, unet_cnns.py:538-549) run as conv+bias+LeakyReLU fused in the conv epilogue, then pool,
    then dropout."""

    def forward(self, x, residual=None):
        mods = list(self)
        conv, act = mods[0], mods[1]
        rest = mods[2:]
        if len(rest) == 2 and isinstance(rest[0], MaxPool2d) and isinstance(rest[1], Dropout):
            (kh, kw), pool = rest[0].kernel_size, rest[0]
            if kw == 1 and kh in ops.POOLROWS_KH and pool.stride == (1, 1) and pool.padding == (kh // 2, 0):
                # the stage's tail (pool over 3 / 13 frames, dropout, residual add) is one kernel -- which also applies the
                # activation's backward pass (an element that wins a window is that window's maximum: its sign is kept with
                # the argmax), unless a hook wants the convolution module's own gradient or the conv takes its GEMM path
                plain = conv.kernel_size[0] == 1 or conv.kernel_size[0] != x.shape[2] or conv.padding[0] != 0
                fuse = (plain and act.act in (ops.ACT_RELU, ops.ACT_LRELU) and torch.is_grad_enabled() and
                        not (conv._forward_hooks or conv._backward_hooks or act._forward_hooks or act._backward_hooks))
                if fuse:
                    h = ops.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, act.act, act.slope, act_bwd_by_consumer=True)
                    return ops.poolrows_dropout_add(h, residual, kh, rest[1].p, self.training,
                                                    producer_slope=act.slope if act.act == ops.ACT_LRELU else 0.0)
                return ops.poolrows_dropout_add(conv(x, act.act, act.slope), residual, kh, rest[1].p, self.training)
        h = conv(x, act.act, act.slope)
        for m in rest:
            h = m(h)
        return h if residual is None else ops.add(h, residual)


class OutputHead(nn.Sequential):
    """conv4: Conv 1x1 + LeakyReLU + Dropout + Conv 1xk + Sigmoid (unet_cnns.py:551-557)."""

    def forward(self, x, return_logits=False):
        conv_a, act, drop, conv_b, sig = list(self)
        h = drop(conv_a(x, act.act, act.slope))
        logits = conv_b(h)
        y = sig(logits)
        return (y, logits) if return_logits else y
