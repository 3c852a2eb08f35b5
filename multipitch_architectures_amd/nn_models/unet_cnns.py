"""U-Net family -- drop-in for the classes of libdl/nn_models/unet_cnns.py that the experiments use.

Constructor signatures, defaults, attribute names and ``state_dict`` keys follow
the reference (file:line cited per class); the arithmetic runs in HIP kernels
via multipitch_architectures_amd.ops.
"""
import torch
import torch.nn as nn

from .. import ops
from .basic_cnns import _head
from .layers import (BatchNorm2d, Conv2d, ConvActPoolDrop, Dropout, ELU, LayerNorm, LeakyReLU, Linear, LSTM, MaxPool2d,
                     MaxUnpool2d, MultiheadAttention, ReLU, SELU, Sigmoid)


class _DoubleConvSeq(nn.Sequential):
    """Conv -> BN -> ReLU -> Dropout, twice.  In training the convolution's store epilogue produces the partial sums
    the BatchNorm statistics are made of (no statistics pass over its output); BN-apply + ReLU run as one kernel."""

    def forward(self, x):
        mods = list(self)
        h = x
        i = 0
        while i < len(mods):
            conv, bn = mods[i], mods[i + 1]
            if bn.training and not (conv._forward_hooks or conv._forward_pre_hooks):
                # (a hooked convolution goes through nn.Module.__call__, as in the reference's nn.Sequential)
                y, partials = conv.forward_stats(h)
                h = bn(y, relu=True, partials=partials)
            else:
                h = bn(conv(h), relu=True)
            i += 3
            if i < len(mods) and isinstance(mods[i], Dropout):
                h = mods[i](h)
                i += 1
        return h


class _AltOrderSeq(nn.Sequential):
    """ELU -> BN -> Dropout -> Conv, twice (double_conv(alt_order=True)); BatchNorm without the fused ReLU."""

    def forward(self, x):
        h = x
        for m in self:
            h = m(h, relu=False) if isinstance(m, BatchNorm2d) else m(h)
        return h


class double_conv(nn.Module):
    """ Two convolutional layers, each followed by batch normalization and ReLU  (unet_cnns.py:30-82)"""

    def __init__(self, in_channels, out_channels, mid_channels=None, kernel_size=(3, 3),
                 padding=(1, 1), convdrop=0, residual=False, alt_order=False):
        super().__init__()
        self.residual = residual
        self.out_channels = out_channels
        if not mid_channels:
            mid_channels = out_channels
        if alt_order:             # unet_cnns.py:60-70: ELU, BN, Dropout, Conv, twice -> Sequential indices 1, 3, 5, 7
            self.double_conv = _AltOrderSeq(
                ELU(alpha=1.0, inplace=False), BatchNorm2d(in_channels), Dropout(p=convdrop),
                Conv2d(in_channels, mid_channels, kernel_size=kernel_size, padding=padding),
                ELU(alpha=1.0, inplace=False), BatchNorm2d(mid_channels), Dropout(p=convdrop),
                Conv2d(mid_channels, out_channels, kernel_size=kernel_size, padding=padding))
        elif convdrop is None:      # unet_cnns.py:40-48: no Dropout slots -> Sequential indices 0,1,3,4
            self.double_conv = _DoubleConvSeq(
                Conv2d(in_channels, mid_channels, kernel_size=kernel_size, padding=padding),
                BatchNorm2d(mid_channels), ReLU(inplace=True),
                Conv2d(mid_channels, out_channels, kernel_size=kernel_size, padding=padding),
                BatchNorm2d(out_channels), ReLU(inplace=True))
        else:                     # unet_cnns.py:49-59 (default convdrop=0): indices 0,1,4,5
            self.double_conv = _DoubleConvSeq(
                Conv2d(in_channels, mid_channels, kernel_size=kernel_size, padding=padding),
                BatchNorm2d(mid_channels), ReLU(inplace=True), Dropout(p=convdrop),
                Conv2d(mid_channels, out_channels, kernel_size=kernel_size, padding=padding),
                BatchNorm2d(out_channels), ReLU(inplace=True), Dropout(p=convdrop))
        if residual:
            self.resize = Conv2d(in_channels, out_channels, kernel_size=(1, 1), padding=(0, 0))

    def forward(self, x):
        if self.residual:
            x, x_res = ops.fanout(x)                 # two consumers: their gradients are added by an in-tree kernel
            x_conv = self.double_conv(x)
            x_resized = self.resize(x_res)
            return ops.add(x_resized, x_conv)
        return self.double_conv(x)


class unet_up_concat_padding(nn.Module):
    """ 2-dimensional upsampling and concatenation with fixing padding issues (unet_cnns.py:85-104)"""

    def __init__(self, upsamp_fac=(2, 2), bilinear=True):
        super().__init__()
        self.upsamp_fac = tuple(int(f) for f in upsamp_fac)
        if len(self.upsamp_fac) != 2 or not all(1 <= f <= 4 for f in self.upsamp_fac):
            raise NotImplementedError("bilinear upsampling factors 1..4 per axis are built ((2,2), and (2,3) of the temporal U-Nets)")

    def forward(self, x1, x2):
        return ops.upconcat(x1, x2, self.upsamp_fac)


def _sinusoidal_pe(max_len, embed_dim):
    """unet_cnns.py:118-124 (built on the CPU; moved to the input's device on first use)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, embed_dim, 2) * (-torch.log(torch.tensor(10000.0)) / embed_dim))
    pe = torch.zeros(max_len, embed_dim, requires_grad=False)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


class transformer_enc_layer(nn.Module):
    """ Transformer encoder layer, with multi-head self-attention and fully connected network (MLP)
    (unet_cnns.py:107-159).  The attention runs over the batch axis, as in the reference (Appendix C.1)."""

    MAX_LEN = 600

    def __init__(self, embed_dim=32, num_heads=8, mlp_dim=512, p_dropout=0.2, pos_encoding=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.pos_encoding = pos_encoding
        max_len = self.MAX_LEN
        if pos_encoding == 'sinusoidal':
            self.pe = _sinusoidal_pe(max_len, embed_dim)          # plain attribute, not in state_dict (Appendix C.3)
            self.dropout_pe = Dropout(p=p_dropout)
        elif pos_encoding == 'learnable':
            self.pe = nn.Parameter(torch.zeros(max_len, embed_dim), requires_grad=True)
            nn.init.kaiming_uniform_(self.pe)
            self.dropout_pe = Dropout(p=p_dropout)
        self.q_linear = Linear(embed_dim, embed_dim, bias=False)
        self.v_linear = Linear(embed_dim, embed_dim, bias=False)
        self.k_linear = Linear(embed_dim, embed_dim, bias=False)
        self.attn = MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads)
        self.o_linear = Linear(embed_dim, embed_dim, bias=False)
        self.mlp = nn.Sequential(
            Linear(embed_dim, mlp_dim),
            ReLU(),
            Linear(mlp_dim, embed_dim)
        )
        self.dropout1 = Dropout(p=p_dropout)
        self.layernorm1 = LayerNorm(normalized_shape=[embed_dim])
        self.dropout2 = Dropout(p=p_dropout)
        self.layernorm2 = LayerNorm(normalized_shape=[embed_dim])

    def _pe_on(self, device):
        pe = self.pe
        if not isinstance(pe, nn.Parameter) and pe.device != device:
            pe = self.pe = pe.to(device)
        return pe

    def forward(self, x):
        B, E, H, W = x.shape
        S = H * W
        if E != self.embed_dim:
            raise RuntimeError(f"transformer_enc_layer: expected {self.embed_dim} channels, got {E}")
        xf = x.reshape(B, E, S)                                  # view
        return ops.transpose_last2(self._encode_tokens(self._tokens(xf))).reshape(B, E, H, W)

    def _tokens(self, xf):
        """(B,E,S) -> (B,S,E) tokens, with the positional table added (and its dropout) when the layer has one"""
        if self.pos_encoding is not None:
            if xf.shape[2] > self.pe.shape[0]:
                raise RuntimeError(f"sequence of {xf.shape[2]} positions exceeds the positional table ({self.pe.shape[0]})")
            return self.dropout_pe(ops.transpose_last2(xf, self._pe_on(xf.device)))
        return ops.transpose_last2(xf)

    def _encode_tokens(self, t):
        """the encoder layer proper on (B,S,E) tokens: attention over the batch axis + MLP, post-norm (unet_cnns.py:153-158)"""
        # t and x1_norm each feed a projection and a residual branch: explicit fan-outs, so that their two gradients are
        # added by an in-tree kernel (ops.fanout) instead of by autograd's accumulation
        t, t_res = ops.fanout(t)
        q, k, v = ops.qkv_linear(t, self.q_linear.weight, self.k_linear.weight, self.v_linear.weight)
        x1 = self.attn(q, k, v)[0]
        x1_proj = self.o_linear(x1)
        x1_norm, x1_res = ops.fanout(self.layernorm1(t_res, self.dropout1(x1_proj)))
        if self.mlp[0]._forward_hooks or self.mlp[2]._forward_hooks:       # a hook wants the hidden tensor of its own module
            x2 = self.mlp[2](self.mlp[0](x1_norm, ops.ACT_RELU))
        else:                           # one node: the ReLU's backward pass inside the GEMM that produces its input
            x2 = ops.mlp_relu(x1_norm, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias)
        return self.layernorm2(x1_res, self.dropout2(x2))


class transformer_temporal_enc_layer(transformer_enc_layer):
    """ Transformer encoder layer only over time dimension (unet_cnns.py:162-217): the tokens are the T' time frames, their
    features the channels x frequency bins (C*F' == embed_dim).  As in transformer_enc_layer the (B,T',E) tokens go to an
    nn.MultiheadAttention with batch_first=False, so the attention still runs over the batch axis (Appendix C.1).  Same
    parameters and state_dict keys as transformer_enc_layer; positional table of 174 rows."""
    MAX_LEN = 174

    def __init__(self, embed_dim=32, num_heads=8, mlp_dim=512, p_dropout=0.2, pos_encoding=None):
        super().__init__(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout,
                         pos_encoding=pos_encoding)
        self.flatten = nn.Flatten(start_dim=-3, end_dim=-2)      # (attribute of the reference; no parameters)

    def forward(self, x):
        B, C, T, Fq = x.shape
        if C * Fq != self.embed_dim:
            raise RuntimeError(f"transformer_temporal_enc_layer: needs C*F' == embed_dim, got {C}*{Fq} vs {self.embed_dim}")
        # x.transpose(2,3) -> flatten(C,F') -> (B, C*F', T): feature index c*F' + f, as blstm_temporal_enc_layer
        xt = ops.transpose_last2(x.reshape(B * C, T, Fq)).reshape(B, C * Fq, T)
        out = self._encode_tokens(self._tokens(xt))                # (B,T,C*F')
        y = ops.transpose_last2(out).reshape(B * C, Fq, T)
        return ops.transpose_last2(y).reshape(B, C, T, Fq)


class blstm_temporal_enc_layer(nn.Module):
    """ BLSTM layer over time dimension (unet_cnns.py:220-243); ignores batch_first / bidirectional like the
    reference does (always True, :232)."""

    def __init__(self, embed_dim=32, hidden_size=512, num_layers=1, batch_first=True, bidirectional=True):
        super().__init__()
        self.embed_dim = embed_dim
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.blstm = LSTM(input_size=embed_dim, hidden_size=hidden_size, num_layers=num_layers, batch_first=True,
                          bidirectional=True)

    def forward(self, x):
        B, C, T, Fq = x.shape
        if C * Fq != self.embed_dim or 2 * self.hidden_size != self.embed_dim:
            raise RuntimeError(f"blstm_temporal_enc_layer: needs C*F' == embed_dim == 2*hidden_size, got C*F'={C * Fq}, "
                               f"embed_dim={self.embed_dim}, hidden_size={self.hidden_size}")
        # x.transpose(2,3) -> (B,C,F',T) -> flatten(C,F') -> (B,C*F',T) -> transpose -> (B,T,C*F')
        xt = ops.transpose_last2(x.reshape(B * C, T, Fq)).reshape(B, C * Fq, T)
        seq = ops.transpose_last2(xt)                              # (B,T,C*F')
        out = self.blstm(seq)[0]                                   # (B,T,2H)
        y = ops.transpose_last2(out).reshape(B * C, Fq, T)         # (B,2H,T) viewed (B,C,F',T)
        return ops.transpose_last2(y).reshape(B, C, T, Fq)


class _UNetTrunk(nn.Module):
    """Layers shared by simple_u_net_largekernels and its descendants (unet_cnns.py:345-393)."""

    # kernel size per level, top (75x216) to bottom: the large-kernel family; simple_u_net uses 3 everywhere
    LARGE = (15, 15, 9, 5, 3)

    def _build_trunk(self, n_in, n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc, convdrop=0, residual=False,
                     alt_order=False, inc_alt_order=False, ks=LARGE):
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        kw = dict(convdrop=convdrop, residual=residual, alt_order=alt_order)
        kp = lambda k: dict(kernel_size=(k, k), padding=(k // 2, k // 2))
        self.inc = double_conv(in_channels=n_in, mid_channels=64 // sc, out_channels=64 // sc, convdrop=convdrop,
                               alt_order=inc_alt_order, **kp(ks[0]))
        self.down1 = nn.Sequential(MaxPool2d((2, 2)), double_conv(in_channels=64 // sc, out_channels=128 // sc, mid_channels=128 // sc, **kp(ks[1]), **kw))
        self.down2 = nn.Sequential(MaxPool2d((2, 2)), double_conv(in_channels=128 // sc, out_channels=256 // sc, mid_channels=256 // sc, **kp(ks[2]), **kw))
        self.down3 = nn.Sequential(MaxPool2d((2, 2)), double_conv(in_channels=256 // sc, out_channels=512 // sc, mid_channels=512 // sc, **kp(ks[3]), **kw))
        self.down4 = nn.Sequential(MaxPool2d((2, 2)), double_conv(in_channels=512 // sc, out_channels=1024 // (sc * 2), mid_channels=1024 // (sc * 2), **kp(ks[4]), **kw))

    def _build_decoder(self, n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc, convdrop=0, residual=False,
                       alt_order=False, ks=LARGE, head=True):
        kw = dict(convdrop=convdrop, residual=residual, alt_order=alt_order)
        kp = lambda k: dict(kernel_size=(k, k), padding=(k // 2, k // 2))
        self.upconcat = unet_up_concat_padding((2, 2))
        self.upconv1 = double_conv(in_channels=1024 // sc, out_channels=512 // (sc * 2), mid_channels=1024 // (sc * 2), **kp(ks[4]), **kw)
        self.upconv2 = double_conv(in_channels=512 // sc, out_channels=256 // (sc * 2), mid_channels=512 // (sc * 2), **kp(ks[3]), **kw)
        self.upconv3 = double_conv(in_channels=256 // sc, out_channels=128 // (sc * 2), mid_channels=256 // (sc * 2), **kp(ks[2]), **kw)
        self.upconv4 = double_conv(in_channels=128 // sc, out_channels=n_ch[0], mid_channels=128 // (sc * 2), **kp(ks[0]), **kw)
        if head:
            self.conv2, self.conv3, self.conv4 = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)
        else:
            self.conv2 = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)[0]

    @staticmethod
    def _down(stage, x):
        """stage(x) for stage = Sequential(MaxPool2d, double_conv), and x as the decoder will consume it.  When gradients
        flow, the pooling and the hand-over to the skip connection are one autograd node (ops.pool_skip): the backward
        pass then adds the two gradients of x inside the pool's backward kernel instead of in an accumulation kernel of
        autograd.  Hooked modules go through nn.Module.__call__ as in the reference's nn.Sequential."""
        pool = stage[0]
        plain = not (torch.is_grad_enabled() and x.requires_grad and len(stage) == 2 and type(pool) is MaxPool2d)
        for m in (stage, pool):
            plain = plain or bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)
        if plain:
            return stage(x), x
        y, skip = ops.pool_skip(x, pool.kernel_size, pool.stride, pool.padding)
        return stage[1](y), skip

    def _encode(self, x):
        x_norm = self.layernorm.forward_cf(x)
        x1 = self.inc(x_norm)
        x2, x1 = self._down(self.down1, x1)
        x3, x2 = self._down(self.down2, x2)
        x4, x3 = self._down(self.down3, x3)
        x5, x4 = self._down(self.down4, x4)
        return x1, x2, x3, x4, x5

    def _decode(self, x1, x2, x3, x4, x5):
        x = self.upconv1(self.upconcat(x5, x4))
        x = self.upconv2(self.upconcat(x, x3))
        x = self.upconv3(self.upconcat(x, x2))
        x = self.upconv4(self.upconcat(x, x1))
        conv2_lrelu = self.conv2(x)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred


class simple_u_net_largekernels(_UNetTrunk):
    """unet_cnns.py:333-407 (Unet:S..XL, exp160*)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216,
                 n_bins_out=12, a_lrelu=0.3, p_dropout=0.2, scalefac=16):
        super().__init__()
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def forward(self, x):
        return self._decode(*self._encode(x))


class simple_u_net_doubleselfattn(_UNetTrunk):
    """unet_cnns.py:496-575 (SAUnet, exp180*): two transformer encoder layers on the bottleneck x5."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, convdrop=0, residual=False, alt_order=False, scalefac=16, embed_dim=4 * 8,
                 num_heads=8, mlp_dim=512, pos_encoding=None):
        super().__init__()
        sc = scalefac
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc, convdrop, residual,
                          alt_order, inc_alt_order=alt_order)
        # note: built with the class default p_dropout=0.2, not the model's p_dropout (unet_cnns.py:528-529)
        self.attention1 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, pos_encoding=pos_encoding)
        self.attention2 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc, convdrop, residual, alt_order)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        x5 = self.attention1(x5)
        x5 = self.attention2(x5)
        return self._decode(x1, x2, x3, x4, x5)


class simple_u_net_doubleselfattn_twolayers(_UNetTrunk):
    """unet_cnns.py:670-754 (SAUSnet, exp181*): attention1/2 on x5 and attention3/4 on the skip x4."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, convdrop=0, residual=False, scalefac=16, embed_dim=4 * 8, num_heads=8,
                 mlp_dim=512, pos_encoding=None):
        super().__init__()
        sc = scalefac
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc, convdrop, residual)
        self.attention1 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout, pos_encoding=pos_encoding)
        self.attention2 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout)
        self.attention3 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout, pos_encoding=pos_encoding)
        self.attention4 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc, convdrop, residual)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        x5 = self.attention1(x5)
        x5 = self.attention2(x5)
        x4 = self.attention3(x4)
        x4 = self.attention4(x4)
        return self._decode(x1, x2, x3, x4, x5)


class u_net_blstm_varlayers(_UNetTrunk):
    """unet_cnns.py:1000-1101 (BLUnet, exp186*): BiLSTM over time on the bottleneck (and optionally the skips)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2, scalefac=8, embed_dim=4 * 16, hidden_size=512, lstm_depth=0, lstm_number=2):
        super().__init__()
        self.lstm_depth = lstm_depth
        self.lstm_number = lstm_number
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        for depth, name in ((0, "lstm5"), (1, "lstm4"), (2, "lstm3"), (3, "lstm2"), (4, "lstm1")):
            if lstm_depth > depth:
                setattr(self, name, blstm_temporal_enc_layer(embed_dim=embed_dim, hidden_size=hidden_size,
                                                             num_layers=lstm_number, batch_first=True, bidirectional=True))
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        if self.lstm_depth > 0:
            x5 = self.lstm5(x5)
        if self.lstm_depth > 1:
            x4 = self.lstm4(x4)
        x = self.upconv1(self.upconcat(x5, x4))
        if self.lstm_depth > 2:
            x3 = self.lstm3(x3)
        x = self.upconv2(self.upconcat(x, x3))
        if self.lstm_depth > 3:
            x2 = self.lstm2(x2)
        x = self.upconv3(self.upconcat(x, x2))
        if self.lstm_depth > 4:
            x1 = self.lstm1(x1)
        x = self.upconv4(self.upconcat(x, x1))
        conv2_lrelu = self.conv2(x)
        conv3_lrelu = self.conv3(conv2_lrelu)
        y_pred = self.conv4(conv3_lrelu)
        return y_pred


class _PolyHead(nn.Sequential):
    """convP: Conv(2,5) + LeakyReLU + MaxPool(2,5)/s(1,2) + Dropout + Conv(2,3) (unet_cnns.py:2311-2318)."""

    def forward(self, x):
        conv_a, act, pool, drop, conv_b = list(self)
        return conv_b(drop(pool(conv_a(x, act.act, act.slope))))


class simple_u_net_polyphony_classif_softmax(_UNetTrunk):
    """unet_cnns.py:2251-2335 (PUnet, exp195*): returns (y_pred, n_pred) with degree-of-polyphony logits."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=16, num_polyphony_steps=24):
        super().__init__()
        sc = scalefac
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc)
        self.convP = _PolyHead(
            Conv2d(1024 // (sc * 2), 1024 // (sc * 4), kernel_size=(2, 5), padding=(0, 0), stride=(1, 1)),
            LeakyReLU(negative_slope=a_lrelu),
            MaxPool2d(kernel_size=(2, 5), stride=(1, 2), padding=(0, 0)),
            Dropout(p=p_dropout),
            Conv2d(1024 // (sc * 4), num_polyphony_steps, kernel_size=(2, 3), padding=(0, 0), stride=(1, 1)))

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        y_pred = self._decode(x1, x2, x3, x4, x5)
        n_pred = self.convP(x5)
        return y_pred, n_pred


# ---------------------------------------------------------------------------------------------------------------------
# Variants the reference exports but no experiment script instantiates: re-compositions of the
# blocks above, kept drop-in (constructor signatures, attribute names = state_dict keys, forward order).  Parity:
# tests/golden/xcls-*.npz, produced by the reference classes themselves (oracle/make_goldens_variants.py).

class simple_u_net(_UNetTrunk):
    """unet_cnns.py:251-330: the U-Net with 3x3 kernels on every level."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216,
                 n_bins_out=12, a_lrelu=0.3, p_dropout=0.2, scalefac=8):
        super().__init__()
        ks = (3, 3, 3, 3, 3)
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac, ks=ks)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac, ks=ks)

    def forward(self, x):
        return self._decode(*self._encode(x))


class simple_u_net_selfattn(_UNetTrunk):
    """unet_cnns.py:415-492: one transformer encoder layer on the bottleneck."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=16, embed_dim=4 * 8, num_heads=8, mlp_dim=512):
        super().__init__()
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self.attention = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim)    # :447
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        return self._decode(x1, x2, x3, x4, self.attention(x5))


class simple_u_net_sixselfattn(_UNetTrunk):
    """unet_cnns.py:579-666: six transformer encoder layers on the bottleneck (class-default dropout, :611-616)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=16, embed_dim=4 * 8, num_heads=8, mlp_dim=512, pos_encoding=None):
        super().__init__()
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self.attention1 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, pos_encoding=pos_encoding)
        for i in range(2, 7):
            setattr(self, f"attention{i}", transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim))
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        for i in range(1, 7):
            x5 = getattr(self, f"attention{i}")(x5)
        return self._decode(x1, x2, x3, x4, x5)


class simple_u_net_doubleselfattn_varlayers(_UNetTrunk):
    """unet_cnns.py:863-996: `self_attn_number` (0..2) transformer encoder layers on the bottleneck and on the skip
    connections of the deepest `self_attn_depth` levels (embed_dim, embed_dim, /2, /4, /8 channels)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2, scalefac=8, embed_dim=4 * 16, num_heads=8, mlp_dim=512, self_attn_depth=0,
                 self_attn_number=2, pos_encoding=None):
        super().__init__()
        self.attn_depth = self_attn_depth
        self.attn_number = self_attn_number
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self._attn_layers(embed_dim, num_heads, mlp_dim, p_dropout, pos_encoding, self_attn_depth, self_attn_number)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def _attn_layers(self, embed_dim, num_heads, mlp_dim, p_dropout, pos_encoding, depth, number):
        for lvl, (level, div) in enumerate(((5, 1), (4, 1), (3, 2), (2, 4), (1, 8))):
            if depth > lvl:
                if number > 0:
                    setattr(self, f"attention{level}a", transformer_enc_layer(embed_dim=embed_dim // div, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout, pos_encoding=pos_encoding))
                if number > 1:
                    setattr(self, f"attention{level}b", transformer_enc_layer(embed_dim=embed_dim // div, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout))

    def _attend(self, t, level, lvl):
        if self.attn_depth > lvl:
            if self.attn_number > 0:
                t = getattr(self, f"attention{level}a")(t)
            if self.attn_number > 1:
                t = getattr(self, f"attention{level}b")(t)
        return t

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        x5 = self._attend(x5, 5, 0)
        x4 = self._attend(x4, 4, 1)
        x = self.upconv1(self.upconcat(x5, x4))
        x3 = self._attend(x3, 3, 2)
        x = self.upconv2(self.upconcat(x, x3))
        x2 = self._attend(x2, 2, 3)
        x = self.upconv3(self.upconcat(x, x2))
        x1 = self._attend(x1, 1, 4)
        x = self.upconv4(self.upconcat(x, x1))
        return self.conv4(self.conv3(self.conv2(x)))


class simple_u_net_doubleselfattn_transenc(simple_u_net_doubleselfattn_varlayers):
    """unet_cnns.py:1370-1521: the _varlayers U-Net (its skip / bottleneck transformer layers without positional encoding)
    whose time reduction is done by two transformer_temporal_enc_layer's on the conv2 output -- tokens = the T frames,
    features = 72 bins x n_chan_layers[1] channels (so time_embed_dim must equal 72 * n_chan_layers[1]) -- followed by a crop
    to the centre frames and a 1x1 `reduction` convolution (which takes n_chan_layers[2] channels: the class only runs with
    n_chan_layers[1] == n_chan_layers[2], as upstream).  Six attention_time layers are constructed, two are used
    (:1511-1516).  Output (B, 1, 1, T-74, 72) -- the reference's `.unsqueeze(1)` of a 4-D tensor."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2, scalefac=8, embed_dim=4 * 16, num_heads=8, mlp_dim=512,
                 self_attn_depth=0, self_attn_number=2, time_embed_dim=256, pos_encoding=None):
        _UNetTrunk.__init__(self)
        self.attn_depth = self_attn_depth
        self.attn_number = self_attn_number
        context_frames = 75
        self.half_context = context_frames // 2
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self._attn_layers(embed_dim, num_heads, mlp_dim, p_dropout, None, self_attn_depth, self_attn_number)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac, head=False)
        self.flatten = nn.Flatten(start_dim=-2, end_dim=-1)
        for i in range(1, 7):
            setattr(self, f"attention_time{i}", transformer_temporal_enc_layer(
                embed_dim=time_embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout,
                pos_encoding=pos_encoding if i == 1 else None))
        self.reduction = nn.Sequential(
            Conv2d(in_channels=n_chan_layers[2], out_channels=1, kernel_size=(1, 1), padding=(0, 0), stride=(1, 1)),
            Sigmoid())

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        x5 = self._attend(x5, 5, 0)
        x4 = self._attend(x4, 4, 1)
        x = self.upconv1(self.upconcat(x5, x4))
        x3 = self._attend(x3, 3, 2)
        x = self.upconv2(self.upconcat(x, x3))
        x2 = self._attend(x2, 2, 3)
        x = self.upconv3(self.upconcat(x, x2))
        x1 = self._attend(x1, 1, 4)
        x = self.upconv4(self.upconcat(x, x1))
        x = self.conv2(x)                                           # (B, n1, T, 72)
        B, n1, T, W = x.shape
        l1, l2 = self.attention_time1, self.attention_time2
        if W * n1 != l1.embed_dim:
            raise RuntimeError(f"simple_u_net_doubleselfattn_transenc: time_embed_dim must be {W} * n_chan_layers[1] = {W * n1}, "
                               f"got {l1.embed_dim}")
        # x.transpose(1,3) -> (B, 72, T, n1) into the temporal layer, whose tokens are then [b, t, w * n1 + ch]: one transpose
        # of the (B, n1, T*72) view instead of the reference's chain of four
        seq = ops.transpose_last2(x.reshape(B, n1, T * W)).reshape(B, T, W * n1)
        if l1.pos_encoding is not None:
            if T > l1.pe.shape[0]:
                raise RuntimeError(f"sequence of {T} frames exceeds the positional table ({l1.pe.shape[0]})")
            seq = l1.dropout_pe(ops.add_rows(seq, l1._pe_on(seq.device)[:T]))
        seq = l2._encode_tokens(l1._encode_tokens(seq))
        x = ops.transpose_last2(seq.reshape(B, T * W, n1)).reshape(B, n1, T, W)      # = layer output .transpose(1, 3)
        x = x[:, :, self.half_context:-self.half_context, :]
        conv, _sig = self.reduction[0], self.reduction[1]
        return conv(x.contiguous(), ops.ACT_SIGMOID, 0.0).unsqueeze(1)


class _TemporalUNetTrunk(_UNetTrunk):
    """Trunk of the two "temporal" U-Nets (unet_cnns.py:1117-1254, 1258-1365): pooling (2,3) -- 75x216 -> 37x72 -> 18x24 -> 9x8
    -> 4x2 -- with channels 16 / 48 / 144 / 432 / 1728 over scalefac, so that channels x bins is the same 3456 / scalefac on
    every level (the embed_dim of the time-axis layers); decoder with (2,3) bilinear upsampling."""

    def _build_temporal(self, n_in, n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc):
        self.layernorm = LayerNorm(normalized_shape=[n_in, n_bins_in])
        kp = lambda k: dict(kernel_size=(k, k), padding=(k // 2, k // 2))
        self.inc = double_conv(in_channels=n_in, mid_channels=16 // sc, out_channels=16 // sc, **kp(15))
        self.down1 = nn.Sequential(MaxPool2d((2, 3)), double_conv(in_channels=16 // sc, out_channels=48 // sc, mid_channels=48 // sc, **kp(15)))
        self.down2 = nn.Sequential(MaxPool2d((2, 3)), double_conv(in_channels=48 // sc, out_channels=144 // sc, mid_channels=144 // sc, **kp(9)))
        self.down3 = nn.Sequential(MaxPool2d((2, 3)), double_conv(in_channels=144 // sc, out_channels=432 // sc, mid_channels=432 // sc, **kp(5)))
        self.down4 = nn.Sequential(MaxPool2d((2, 3)), double_conv(in_channels=432 // sc, out_channels=1728 // sc, mid_channels=1728 // sc, **kp(3)))

    def _build_temporal_decoder(self, n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc):
        kp = lambda k: dict(kernel_size=(k, k), padding=(k // 2, k // 2))
        self.upconcatsize2 = unet_up_concat_padding((2, 2))
        self.upconcatsize3 = unet_up_concat_padding((2, 3))
        self.upconv1 = double_conv(in_channels=(1728 + 432) // sc, out_channels=144 // sc, mid_channels=(1728 + 432) // (2 * sc), **kp(3))
        self.upconv2 = double_conv(in_channels=2 * 144 // sc, out_channels=48 // sc, mid_channels=144 // sc, **kp(5))
        self.upconv3 = double_conv(in_channels=2 * 48 // sc, out_channels=16 // sc, mid_channels=48 // sc, **kp(9))
        self.upconv4 = double_conv(in_channels=2 * 16 // sc, out_channels=n_ch[0], mid_channels=48 // sc, **kp(15))
        self.conv2, self.conv3, self.conv4 = _head(n_ch[0], n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)

    def _decode_temporal(self, x1, x2, x3, x4, x5, on_skip):
        """on_skip(tensor, level): the level's time-axis layers (identity above the configured depth)"""
        x5 = on_skip(x5, 5)
        x4 = on_skip(x4, 4)
        x = self.upconv1(self.upconcatsize3(x5, x4))
        x3 = on_skip(x3, 3)
        x = self.upconv2(self.upconcatsize3(x, x3))
        x2 = on_skip(x2, 2)
        x = self.upconv3(self.upconcatsize3(x, x2))
        x1 = on_skip(x1, 1)
        x = self.upconv4(self.upconcatsize3(x, x1))
        return self.conv4(self.conv3(self.conv2(x)))


class u_net_temporal_selfattn_varlayers(_TemporalUNetTrunk):
    """unet_cnns.py:1117-1254: `self_attn_number` (0..2) transformer_temporal_enc_layer's on the bottleneck and on the skip
    connections of the deepest `self_attn_depth` levels; embed_dim must be 3456 // scalefac (channels x bins of a level)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2, scalefac=8, embed_dim=4 * 16, num_heads=8, mlp_dim=512, self_attn_depth=0,
                 self_attn_number=2, pos_encoding=None):
        super().__init__()
        self.attn_depth = self_attn_depth
        self.attn_number = self_attn_number
        self._build_temporal(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        for lvl, level in enumerate((5, 4, 3, 2, 1)):
            if self_attn_depth > lvl:
                if self_attn_number > 0:
                    setattr(self, f"attention{level}a", transformer_temporal_enc_layer(
                        embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout, pos_encoding=pos_encoding))
                if self_attn_number > 1:
                    setattr(self, f"attention{level}b", transformer_temporal_enc_layer(
                        embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, p_dropout=p_dropout))
        self._build_temporal_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def _on_skip(self, t, level):
        if self.attn_depth > 5 - level:
            if self.attn_number > 0:
                t = getattr(self, f"attention{level}a")(t)
            if self.attn_number > 1:
                t = getattr(self, f"attention{level}b")(t)
        return t

    def forward(self, x):
        return self._decode_temporal(*self._encode(x), self._on_skip)


class u_net_temporal_blstm_varlayers(_TemporalUNetTrunk):
    """unet_cnns.py:1258-1365: as u_net_temporal_selfattn_varlayers with a blstm_temporal_enc_layer (`lstm_number` stacked
    bidirectional LSTM layers) per level instead of the transformer layers; needs 2 * hidden_size == embed_dim ==
    3456 // scalefac."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12, a_lrelu=0.3,
                 p_dropout=0.2, scalefac=8, embed_dim=4 * 16, hidden_size=512, lstm_depth=0, lstm_number=2):
        super().__init__()
        self.lstm_depth = lstm_depth
        self.lstm_number = lstm_number
        self._build_temporal(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self.flatten = nn.Flatten(start_dim=-3, end_dim=-2)
        for lvl, level in enumerate((5, 4, 3, 2, 1)):
            if lstm_depth > lvl:
                setattr(self, f"lstm{level}", blstm_temporal_enc_layer(embed_dim=embed_dim, hidden_size=hidden_size,
                                                                       num_layers=lstm_number, batch_first=True, bidirectional=True))
        self._build_temporal_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)

    def _on_skip(self, t, level):
        return getattr(self, f"lstm{level}")(t) if self.lstm_depth > 5 - level else t

    def forward(self, x):
        return self._decode_temporal(*self._encode(x), self._on_skip)


class _FreqStage(nn.Sequential):
    """[BatchNorm2d ->] Conv2d -> SELU of the frequency U-Nets (unet_cnns.py:1710-1765)"""

    def forward(self, x):
        h = x
        for m in self:
            h = m(h, relu=False) if isinstance(m, BatchNorm2d) else m(h)
        return h


class freq_u_net_selfattn(nn.Module):
    """unet_cnns.py:1691-1814: a U-Net that pools along frequency only -- on (B, C, bins, frames) tensors: 216 -> 72 -> 9 -> 1
    bins by MaxPool2d((3,1)) / ((8,1)) / ((9,1)) with return_indices -- SELU activations, MaxUnpool2d with the transferred
    pooling indices as skip strategy, and at the one-bin bottleneck a transformer block on the (B, frames, channels) tokens
    (nn.MultiheadAttention with batch_first=False again: attention over the batch axis, Appendix C.1)."""

    N_BLOCKS = 1

    def __init__(self, n_chan_input=6, n_chan_layers=[32, 30, 20, 10], n_bins_in=216, n_bins_out=72, a_lrelu=0.3,
                 p_dropout=0.2, scalefac=1, embed_dim=64, num_heads=8, mlp_dim=512):
        super().__init__()
        n_ch, sc = n_chan_layers, scalefac
        assert embed_dim % num_heads == 0, 'embed_dim must be a multiple of num_heads!'
        self.num_heads, self.embed_dim, self.head_dim = num_heads, embed_dim, embed_dim // num_heads
        c1, c2, c3 = int(32 / sc), int(64 / sc), int(128 / sc)
        self.layernorm = LayerNorm(normalized_shape=[n_chan_input, n_bins_in])
        self.conv1 = _FreqStage(Conv2d(6, c1, 5, padding=2), SELU())
        self.pool1 = MaxPool2d((3, 1), return_indices=True)
        self.conv2 = _FreqStage(BatchNorm2d(c1), Conv2d(c1, c2, 5, padding=2), SELU())
        self.pool2 = MaxPool2d((8, 1), return_indices=True)
        self.conv3 = _FreqStage(BatchNorm2d(c2), Conv2d(c2, c3, 3, padding=1), SELU())
        self.pool3 = MaxPool2d((9, 1), return_indices=True)
        # transformer block(s) at the bottleneck: attribute names of the reference (no suffix, then "2")
        for blk in range(self.N_BLOCKS):
            sfx = "" if blk == 0 else str(blk + 1)
            n_attn, n_mlp = 5 + 2 * blk, 6 + 2 * blk
            setattr(self, "q_linear" + sfx, Linear(c3, embed_dim, bias=False))
            setattr(self, "v_linear" + sfx, Linear(c3, embed_dim, bias=False))
            setattr(self, "k_linear" + sfx, Linear(c3, embed_dim, bias=False))
            setattr(self, "attn" + sfx, MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads))
            setattr(self, "o_linear" + sfx, Linear(embed_dim, c3, bias=False))
            setattr(self, f"dropout{n_attn}", Dropout(p=p_dropout))
            setattr(self, f"layernorm{n_attn}", LayerNorm(normalized_shape=[c3]))
            setattr(self, f"mlp{n_mlp}", nn.Sequential(Linear(c3, mlp_dim), ReLU(), Linear(mlp_dim, c3)))
            setattr(self, f"dropout{n_mlp}", Dropout(p=p_dropout))
            setattr(self, f"layernorm{n_mlp}", LayerNorm(normalized_shape=[c3]))
        self.up_pool3 = MaxUnpool2d((9, 1))
        self.up_conv3 = _FreqStage(BatchNorm2d(c3), Conv2d(c3, c2, 3, padding=1), SELU())
        self.up_pool2 = MaxUnpool2d((8, 1))
        self.up_conv2 = _FreqStage(BatchNorm2d(c2), Conv2d(c2, c1, 5, padding=2), SELU())
        self.up_pool1 = MaxUnpool2d((3, 1))
        self.up_conv1 = _FreqStage(BatchNorm2d(c1), Conv2d(c1, int(n_ch[0] / sc), 5, padding=2), SELU())
        # binning to MIDI pitches, time reduction, chroma reduction: the usual head under the names conv4 / conv5 / conv6
        self.conv4, self.conv5, self.conv6 = _head(int(n_ch[0] / sc), n_ch, n_bins_in, n_bins_out, a_lrelu, p_dropout)

    @staticmethod
    def _swap(x):
        """(B, C, H, W) -> (B, C, W, H)"""
        B, C, H, W = x.shape
        return ops.transpose_last2(x.reshape(B * C, H, W)).reshape(B, C, W, H)

    def _block(self, t, blk):
        sfx = "" if blk == 0 else str(blk + 1)
        n_attn, n_mlp = 5 + 2 * blk, 6 + 2 * blk
        g = lambda name: getattr(self, name)
        t, t_res = ops.fanout(t)
        q, k, v = ops.qkv_linear(t, g("q_linear" + sfx).weight, g("k_linear" + sfx).weight, g("v_linear" + sfx).weight)
        a = g("o_linear" + sfx)(g("attn" + sfx)(q, k, v)[0])
        a_norm, a_res = ops.fanout(g(f"layernorm{n_attn}")(t_res, g(f"dropout{n_attn}")(a)))
        mlp = g(f"mlp{n_mlp}")
        m = mlp[2](mlp[0](a_norm, ops.ACT_RELU))
        return g(f"layernorm{n_mlp}")(a_res, g(f"dropout{n_mlp}")(m))

    def forward(self, x):
        x_norm = self._swap(self.layernorm.forward_cf(x))              # (B, 6, bins, frames)
        c1, ind1 = self.pool1(self.conv1(x_norm))
        c2, ind2 = self.pool2(self.conv2(c1))
        c3, ind3 = self.pool3(self.conv3(c2))
        B, C, one, T = c3.shape
        if one != 1:
            raise RuntimeError(f"freq_u_net: the three poolings must reduce the {x.shape[3]} bins to 1, got {one}")
        t = ops.transpose_last2(c3.reshape(B, C, T))                    # (B, frames, channels)
        for blk in range(self.N_BLOCKS):
            t = self._block(t, blk)
        bott = ops.transpose_last2(t).reshape(B, C, 1, T)
        u3 = self.up_conv3(self.up_pool3(bott, ind3))
        u2 = self.up_conv2(self.up_pool2(u3, ind2))
        u1 = self.up_conv1(self.up_pool1(u2, ind1))
        return self.conv6(self.conv5(self.conv4(self._swap(u1))))


class freq_u_net_doubleselfattn(freq_u_net_selfattn):
    """unet_cnns.py:1820-1961: freq_u_net_selfattn with two transformer blocks at the bottleneck (q_linear2 .. layernorm8)."""
    N_BLOCKS = 2


class simple_u_net_doubleselfattn_alllayers(simple_u_net_doubleselfattn_varlayers):
    """unet_cnns.py:758-860: two transformer encoder layers on the bottleneck and on every skip connection."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=8, embed_dim=4 * 16, num_heads=8, mlp_dim=512):
        _UNetTrunk.__init__(self)
        self.attn_depth, self.attn_number = 5, 2
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self._attn_layers(embed_dim, num_heads, mlp_dim, p_dropout, None, 5, 2)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)


class _PolyHeadReLU(nn.Sequential):
    """convP of the regression / classification variants: as _PolyHead with a final ReLU (unet_cnns.py:2040-2047)."""

    def forward(self, x):
        conv_a, act, pool, drop, conv_b, _relu = list(self)
        return conv_b(drop(pool(conv_a(x, act.act, act.slope))), ops.ACT_RELU)


def _conv_p(c_in, c_mid, c_out, a_lrelu, p_dropout):
    return _PolyHeadReLU(
        Conv2d(c_in, c_mid, kernel_size=(2, 5), padding=(0, 0), stride=(1, 1)), LeakyReLU(negative_slope=a_lrelu),
        MaxPool2d(kernel_size=(2, 5), stride=(1, 2), padding=(0, 0)), Dropout(p=p_dropout),
        Conv2d(c_mid, c_out, kernel_size=(2, 3), padding=(0, 0), stride=(1, 1)), ReLU())


class simple_u_net_polyphony_classif(_UNetTrunk):
    """unet_cnns.py:2163-2248: degree-of-polyphony head (ReLU output) on the bottleneck; returns (y_pred, n_pred)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=16, num_polyphony_steps=24):
        super().__init__()
        sc = scalefac
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, sc)
        self.convP = _conv_p(1024 // (sc * 2), 1024 // (sc * 4), num_polyphony_steps, a_lrelu, p_dropout)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        return self._decode(x1, x2, x3, x4, x5), self.convP(x5)


class _SAUnetPoly(_UNetTrunk):
    def _build(self, n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac, embed_dim, num_heads,
               mlp_dim, pos_encoding, poly_mid, poly_out):
        self._build_trunk(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self.attention1 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim, pos_encoding=pos_encoding)
        self.attention2 = transformer_enc_layer(embed_dim=embed_dim, num_heads=num_heads, mlp_dim=mlp_dim)
        self._build_decoder(n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac)
        self.convP = _conv_p(embed_dim, poly_mid, poly_out, a_lrelu, p_dropout)

    def forward(self, x):
        x1, x2, x3, x4, x5 = self._encode(x)
        x5_inner = self.attention1(x5)
        x5 = self.attention2(x5_inner)
        return self._decode(x1, x2, x3, x4, x5), self.convP(x5_inner)


class simple_u_net_doubleselfattn_polyphony(_SAUnetPoly):
    """unet_cnns.py:1977-2067: SAUnet with a one-channel polyphony regression head on the first attention layer's output."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=16, embed_dim=4 * 8, num_heads=8, mlp_dim=512, pos_encoding=None):
        super().__init__()
        self._build(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac, embed_dim, num_heads,
                    mlp_dim, pos_encoding, embed_dim // 4, 1)


class simple_u_net_doubleselfattn_polyphony_classif(_SAUnetPoly):
    """unet_cnns.py:2070-2160: as above with `num_polyphony_steps` classes (convP: embed_dim -> embed_dim/2 -> steps)."""

    def __init__(self, n_chan_input=6, n_chan_layers=[64, 30, 20, 10], n_bins_in=216, n_bins_out=12,
                 a_lrelu=0.3, p_dropout=0.2, scalefac=16, embed_dim=4 * 8, num_heads=8, mlp_dim=512, pos_encoding=None,
                 num_polyphony_steps=24):
        super().__init__()
        self._build(n_chan_input, n_chan_layers, n_bins_in, n_bins_out, a_lrelu, p_dropout, scalefac, embed_dim, num_heads,
                    mlp_dim, pos_encoding, embed_dim // 2, num_polyphony_steps)
