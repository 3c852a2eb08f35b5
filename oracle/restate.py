"""ORACLE -- test infrastructure only.  Never imported by the product path.

CPU fp32 restatement (torch-CPU tensor arithmetic) of the reference's
``libdl.nn_models`` forward pass, written functionally over a ``state_dict`` so
that it shares no code with the product modules in
``multipitch_architectures_amd/nn_models``.  The arithmetic of the reference
lives in a third-party dependency that is not vendored (PyTorch, pinned
``pytorch=1.6.0`` in /root/reference/environment.yml:21-24); this file restates
the published semantics of each op and is *pinned* by
the golden vectors in tests/golden/, which were produced by importing the
reference itself (oracle/make_goldens.py) -- see tests/test_oracle_goldens.py.

Only conv2d / max_pool2d use the library primitive (F.conv2d, F.max_pool2d) for
speed; LayerNorm, BatchNorm, bilinear upsampling, multi-head attention over the
batch axis, the BiLSTM recurrence, BCE and AdamW are written out explicitly.

Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

EPS = 1e-5


# --------------------------------------------------------------------------- primitive ops
def layernorm_cf(x, w, b):
    """LayerNorm([C,F]) on x.transpose(1,2) (unet_cnns.py:505,560; basic_cnns.py:160,190).
    x (B,C,T,F): every (b,t) slice of C*F values is normalised jointly."""
    mu = x.mean(dim=(1, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 3), keepdim=True)
    return (x - mu) / torch.sqrt(var + EPS) * w[None, :, None, :] + b[None, :, None, :]


def layernorm_last(x, w, b):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + EPS) * w + b


def batchnorm2d(x, sd, prefix, train, momentum=0.1):
    """nn.BatchNorm2d defaults: biased var for normalisation, unbiased into running_var."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if train:
        n = x.shape[0] * x.shape[2] * x.shape[3]
        mu = x.mean(dim=(0, 2, 3))
        var = ((x - mu[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        with torch.no_grad():
            sd[prefix + ".running_mean"].mul_(1 - momentum).add_(momentum * mu.detach())
            sd[prefix + ".running_var"].mul_(1 - momentum).add_(momentum * var.detach() * n / max(n - 1, 1))
            sd[prefix + ".num_batches_tracked"].add_(1)
    else:
        mu, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    xh = (x - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def lrelu(x, a):
    return torch.where(x >= 0, x, a * x)


def dropout(x, p, train):
    if train and p > 0:
        return F.dropout(x, p, True)
    return x


def conv(x, sd, prefix, stride=(1, 1), padding=(0, 0)):
    return F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"], stride=stride, padding=padding)


def upsample2x_bilinear_ac(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True): src = dst*(in-1)/(out-1)."""
    B, C, H, W = x.shape

    def axis(n_in):
        n_out = 2 * n_in
        if n_in == 1:
            z = torch.zeros(n_out, dtype=torch.long)
            return z, z, torch.zeros(n_out, dtype=x.dtype)
        # index arithmetic in the tensor's own precision, as ATen's area_pixel_compute_source_index does
        scale = torch.tensor(float(n_in - 1), dtype=x.dtype) / torch.tensor(float(n_out - 1), dtype=x.dtype)
        src = torch.arange(n_out, dtype=x.dtype) * scale
        i0 = src.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, src - i0.to(x.dtype)

    y0, y1, ly = axis(H)
    x0, x1, lx = axis(W)
    ly = ly[None, None, :, None]
    lx = lx[None, None, None, :]
    r0 = x[:, :, y0, :]
    r1 = x[:, :, y1, :]
    top = r0[:, :, :, x0] * (1 - lx) + r0[:, :, :, x1] * lx
    bot = r1[:, :, :, x0] * (1 - lx) + r1[:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def upconcat(x1, x2):
    """unet_up_concat_padding.forward (unet_cnns.py:93-104): skip channels first."""
    x1 = upsample2x_bilinear_ac(x1)
    dY = x2.shape[2] - x1.shape[2]
    dX = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])
    return torch.cat([x2, x1], dim=1)


def double_conv(x, sd, prefix, pad, train, residual=False, convdrop=0, alt_order=False):
    """double_conv default branch (unet_cnns.py:49-59): conv 0, BN 1, ReLU, Drop, conv 4, BN 5, ReLU, Drop;
    alt_order (unet_cnns.py:60-70): ELU 0, BN 1, Drop, conv 3, ELU 4, BN 5, Drop, conv 7."""
    p = prefix + ".double_conv"
    if alt_order:
        h = dropout(batchnorm2d(F.elu(x), sd, p + ".1", train), convdrop, train)
        h = conv(h, sd, p + ".3", padding=pad)
        h = dropout(batchnorm2d(F.elu(h), sd, p + ".5", train), convdrop, train)
        h = conv(h, sd, p + ".7", padding=pad)
        if residual:
            h = conv(x, sd, prefix + ".resize") + h
        return h
    h = conv(x, sd, p + ".0", padding=pad)
    h = dropout(torch.relu(batchnorm2d(h, sd, p + ".1", train)), convdrop, train)
    h = conv(h, sd, p + ".4", padding=pad)
    h = dropout(torch.relu(batchnorm2d(h, sd, p + ".5", train)), convdrop, train)
    if residual:
        h = conv(x, sd, prefix + ".resize") + h      # unet_cnns.py:73-80
    return h


def sinusoidal_pe(max_len, E):
    """unet_cnns.py:118-124."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, E, 2) * (-torch.log(torch.tensor(10000.0)) / E))
    pe = torch.zeros(max_len, E)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def transformer_enc_layer(x, sd, prefix, num_heads, train, p_dropout=0.2, pos_encoding=None):
    """transformer_enc_layer.forward (unet_cnns.py:148-159).  nn.MultiheadAttention is fed
    (B,S,E) with batch_first=False, i.e. softmax runs over the *batch* axis (Appendix C.1)."""
    B, E, H, W = x.shape
    S = H * W
    t = x.reshape(B, E, S).transpose(1, 2)                       # (B,S,E)
    if pos_encoding == "sinusoidal":
        t = dropout(t + sinusoidal_pe(600, E)[:S].to(t.dtype), p_dropout, train)
    elif pos_encoding == "learnable":
        t = dropout(t + sd[prefix + ".pe"][:S], p_dropout, train)
    q = t @ sd[prefix + ".q_linear.weight"].T
    k = t @ sd[prefix + ".k_linear.weight"].T
    v = t @ sd[prefix + ".v_linear.weight"].T
    Win, bin_ = sd[prefix + ".attn.in_proj_weight"], sd[prefix + ".attn.in_proj_bias"]
    q = q @ Win[:E].T + bin_[:E]
    k = k @ Win[E:2 * E].T + bin_[E:2 * E]
    v = v @ Win[2 * E:].T + bin_[2 * E:]
    d = E // num_heads
    # (L=B, N=S, h, d) -> (N, h, L, d)
    qh = q.reshape(B, S, num_heads, d).permute(1, 2, 0, 3) * (d ** -0.5)
    kh = k.reshape(B, S, num_heads, d).permute(1, 2, 0, 3)
    vh = v.reshape(B, S, num_heads, d).permute(1, 2, 0, 3)
    a = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1)         # (S,h,B,B)
    o = (a @ vh).permute(2, 0, 1, 3).reshape(B, S, E)
    o = o @ sd[prefix + ".attn.out_proj.weight"].T + sd[prefix + ".attn.out_proj.bias"]
    o = o @ sd[prefix + ".o_linear.weight"].T
    y1 = layernorm_last(t + dropout(o, p_dropout, train), sd[prefix + ".layernorm1.weight"], sd[prefix + ".layernorm1.bias"])
    m = torch.relu(y1 @ sd[prefix + ".mlp.0.weight"].T + sd[prefix + ".mlp.0.bias"])
    m = m @ sd[prefix + ".mlp.2.weight"].T + sd[prefix + ".mlp.2.bias"]
    y2 = layernorm_last(y1 + dropout(m, p_dropout, train), sd[prefix + ".layernorm2.weight"], sd[prefix + ".layernorm2.bias"])
    return y2.transpose(1, 2).reshape(B, E, H, W)


def blstm_temporal_enc_layer(x, sd, prefix, hidden, num_layers):
    """blstm_temporal_enc_layer.forward (unet_cnns.py:235-243) with the LSTM recurrence written out
    (gate order i,f,g,o; h0=c0=0; reverse direction t=T'-1..0; outputs concatenated)."""
    B, C, T, Fq = x.shape
    xs = x.transpose(2, 3).reshape(B, C * Fq, T).transpose(1, 2)  # (B,T,C*F')
    inp = xs
    for layer in range(num_layers):
        outs = []
        for suffix, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
            Wih = sd[f"{prefix}.blstm.weight_ih_l{layer}{suffix}"]
            Whh = sd[f"{prefix}.blstm.weight_hh_l{layer}{suffix}"]
            bih = sd[f"{prefix}.blstm.bias_ih_l{layer}{suffix}"]
            bhh = sd[f"{prefix}.blstm.bias_hh_l{layer}{suffix}"]
            h = torch.zeros(B, hidden, dtype=x.dtype)
            c = torch.zeros(B, hidden, dtype=x.dtype)
            hs = [None] * T
            for t in order:
                g = inp[:, t] @ Wih.T + bih + h @ Whh.T + bhh
                i, f, gg, o = g.split(hidden, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                hs[t] = h
            outs.append(torch.stack(hs, dim=1))
        inp = torch.cat(outs, dim=2)                              # (B,T,2H)
    return inp.transpose(1, 2).reshape(B, C, Fq, T).transpose(2, 3)


def head(x, sd, a_lrelu, p_dropout, train, taps=None):
    """conv2 / conv3 / conv4 head shared by every model (unet_cnns.py:538-557; basic_cnns.py:168-188)."""
    h = conv(x, sd, "conv2.0", stride=(1, 3), padding=(1, 0))
    h = F.max_pool2d(lrelu(h, a_lrelu), (13, 1), (1, 1), (6, 0))
    h = dropout(h, p_dropout, train)
    if taps is not None:
        taps["conv2"] = h
    h = dropout(lrelu(conv(h, sd, "conv3.0"), a_lrelu), p_dropout, train)
    if taps is not None:
        taps["conv3"] = h
    h = dropout(lrelu(conv(h, sd, "conv4.0"), a_lrelu), p_dropout, train)
    logits = conv(h, sd, "conv4.3")
    if taps is not None:
        taps["logits"] = logits
    return torch.sigmoid(logits)


# --------------------------------------------------------------------------- models
def _cnn_prefilter(x, sd, prefix, a_lrelu, p_dropout, train):
    h = lrelu(conv(x, sd, prefix + ".0", padding=(7, 7)), a_lrelu)
    return dropout(F.max_pool2d(h, (3, 1), (1, 1), (1, 0)), p_dropout, train)


def deep_cnn_segm_sigmoid(sd, x, train=False, taps=None, n_prefilt_layers=1, residual=False,
                          a_lrelu=0.3, p_dropout=0.2, **_):
    """basic_cnns.py:363-423 (n_prefilt_layers=1 is basic_cnn_segm_sigmoid, :152-195)."""
    h = layernorm_cf(x, sd["layernorm.weight"], sd["layernorm.bias"])
    if taps is not None:
        taps["x_norm"] = h
    h = _cnn_prefilter(h, sd, "conv1", a_lrelu, p_dropout, train)
    if taps is not None:
        taps["conv1"] = h
    for p in range(n_prefilt_layers - 1):
        hn = _cnn_prefilter(h, sd, f"prefilt_list.{p}", a_lrelu, p_dropout, train)
        if taps is not None:
            taps[f"prefilt{p}"] = hn
        h = hn + h if residual else hn
    return head(h, sd, a_lrelu, p_dropout, train, taps)


def basic_cnn_segm_sigmoid(sd, x, train=False, taps=None, **kw):
    kw.pop("n_prefilt_layers", None)
    kw.pop("residual", None)
    return deep_cnn_segm_sigmoid(sd, x, train, taps, n_prefilt_layers=1, residual=False, **kw)


_UNET_K = {"inc": 7, "down1.1": 7, "down2.1": 4, "down3.1": 2, "down4.1": 1,
           "upconv1": 1, "upconv2": 2, "upconv3": 4, "upconv4": 7}


def _unet(sd, x, train, taps, a_lrelu, p_dropout, convdrop=0, residual=False, bottleneck=None, skip4=None, alt_order=False):
    """Shared trunk of simple_u_net_largekernels and its descendants (unet_cnns.py:395-407)."""
    t = taps if taps is not None else {}
    dc = lambda h, name, res: double_conv(h, sd, name, (_UNET_K[name],) * 2, train, res, convdrop, alt_order)
    h = layernorm_cf(x, sd["layernorm.weight"], sd["layernorm.bias"])
    t["x_norm"] = h
    x1 = dc(h, "inc", False)
    x2 = dc(F.max_pool2d(x1, 2), "down1.1", residual)
    x3 = dc(F.max_pool2d(x2, 2), "down2.1", residual)
    x4 = dc(F.max_pool2d(x3, 2), "down3.1", residual)
    x5 = dc(F.max_pool2d(x4, 2), "down4.1", residual)
    t.update(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5)
    if bottleneck is not None:
        x5 = bottleneck(x5)
        t["x5b"] = x5
    if skip4 is not None:
        x4 = skip4(x4)
        t["x4b"] = x4
    u = dc(upconcat(x5, x4), "upconv1", residual)
    t["u1"] = u
    u = dc(upconcat(u, x3), "upconv2", residual)
    t["u2"] = u
    u = dc(upconcat(u, x2), "upconv3", residual)
    t["u3"] = u
    u = dc(upconcat(u, x1), "upconv4", residual)
    t["u4"] = u
    return head(u, sd, a_lrelu, p_dropout, train, taps), x5, t


def simple_u_net_largekernels(sd, x, train=False, taps=None, a_lrelu=0.3, p_dropout=0.2, **_):
    return _unet(sd, x, train, taps, a_lrelu, p_dropout)[0]


def simple_u_net_doubleselfattn(sd, x, train=False, taps=None, a_lrelu=0.3, p_dropout=0.2, convdrop=0,
                                residual=False, num_heads=8, pos_encoding=None, alt_order=False, **_):
    """unet_cnns.py:559-575.  attention1/2 are built with the *default* p_dropout=0.2 (:528-529)."""
    def bott(x5):
        x5 = transformer_enc_layer(x5, sd, "attention1", num_heads, train, 0.2, pos_encoding)
        return transformer_enc_layer(x5, sd, "attention2", num_heads, train, 0.2, None)
    return _unet(sd, x, train, taps, a_lrelu, p_dropout, convdrop, residual, bottleneck=bott, alt_order=alt_order)[0]


def simple_u_net_doubleselfattn_twolayers(sd, x, train=False, taps=None, a_lrelu=0.3, p_dropout=0.2, convdrop=0,
                                          residual=False, num_heads=8, pos_encoding=None, **_):
    """unet_cnns.py:739-754; here the transformer layers receive p_dropout (:702-705)."""
    def bott(x5):
        x5 = transformer_enc_layer(x5, sd, "attention1", num_heads, train, p_dropout, pos_encoding)
        return transformer_enc_layer(x5, sd, "attention2", num_heads, train, p_dropout, None)

    def sk4(x4):
        x4 = transformer_enc_layer(x4, sd, "attention3", num_heads, train, p_dropout, pos_encoding)
        return transformer_enc_layer(x4, sd, "attention4", num_heads, train, p_dropout, None)
    return _unet(sd, x, train, taps, a_lrelu, p_dropout, convdrop, residual, bottleneck=bott, skip4=sk4)[0]


def u_net_blstm_varlayers(sd, x, train=False, taps=None, a_lrelu=0.3, p_dropout=0.2, hidden_size=512,
                          lstm_depth=0, lstm_number=2, **_):
    """unet_cnns.py:1078-1101.  lstm_depth in {0,1} are the only depths that can run at all: `lstm4` (:1037-1038) is built
    with the same embed_dim as `lstm5`, but the skip x4 it receives (:1088) has C*27 features against x5's C*13, so for
    lstm_depth > 1 the reference's nn.LSTM raises RuntimeError("input.size(-1) must be equal to input_size ...") on the
    first forward (checked against the imported reference) -- restated here as the same exception type."""
    if lstm_depth > 1:
        x4_features = sd["down3.1.double_conv.4.weight"].shape[0] * (x.shape[3] // 8)
        raise RuntimeError(f"input.size(-1) must be equal to input_size. Expected {sd['lstm4.blstm.weight_ih_l0'].shape[1]}, "
                           f"got {x4_features}")
    bott = (lambda x5: blstm_temporal_enc_layer(x5, sd, "lstm5", hidden_size, lstm_number)) if lstm_depth > 0 else None
    return _unet(sd, x, train, taps, a_lrelu, p_dropout, bottleneck=bott)[0]


def simple_u_net_polyphony_classif_softmax(sd, x, train=False, taps=None, a_lrelu=0.3, p_dropout=0.2, **_):
    """unet_cnns.py:2320-2335: returns (y_pred, n_pred)."""
    y, x5, _t = _unet(sd, x, train, taps, a_lrelu, p_dropout)
    h = lrelu(conv(x5, sd, "convP.0"), a_lrelu)
    h = dropout(F.max_pool2d(h, (2, 5), (1, 2)), p_dropout, train)
    n = conv(h, sd, "convP.4")
    if taps is not None:
        taps["n_pred"] = n
    return y, n


MODELS = {f.__name__: f for f in (
    basic_cnn_segm_sigmoid, deep_cnn_segm_sigmoid, simple_u_net_largekernels, simple_u_net_doubleselfattn,
    simple_u_net_doubleselfattn_twolayers, u_net_blstm_varlayers, simple_u_net_polyphony_classif_softmax)}


# --------------------------------------------------------------------------- losses / optimizer (caller side, a11)
class _BCEMean(torch.autograd.Function):
    """torch.nn.BCELoss(reduction='mean'): forward with the log terms clamped at -100, backward
    (p - y) / max(p (1 - p), 1e-12) / N  (ATen's binary_cross_entropy_backward).  Differentiating the clamped logs
    instead would give 0 * inf = NaN as soon as a probability saturates to exactly 0 or 1."""

    @staticmethod
    def forward(ctx, p, y):
        ctx.save_for_backward(p, y)
        lp = torch.clamp(torch.log(p), min=-100.0)
        lq = torch.clamp(torch.log(1 - p), min=-100.0)
        return -(y * lp + (1 - y) * lq).mean()

    @staticmethod
    def backward(ctx, g):
        p, y = ctx.saved_tensors
        return g * (p - y) / torch.clamp((1 - p) * p, min=1e-12) / p.numel(), None


def bce_loss(p, y):
    """torch.nn.BCELoss(reduction='mean') (exp126a...py:87)."""
    return _BCEMean.apply(p, y)


def punet_loss(y_pred, n_pred, y):
    """exp195f...py:331-334: BCE + CrossEntropy(n_pred, sum_pitch y)/25."""
    n_target = y.sum(dim=3).long()                                # (B,1,T')
    logits = n_pred                                               # (B,K,1,1)
    lse = torch.logsumexp(logits, dim=1)                          # (B,1,1)
    picked = torch.gather(logits, 1, n_target.unsqueeze(1)).squeeze(1)
    return bce_loss(y_pred, y) + (lse - picked).mean() / 25.0


def adamw_step(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
    """torch.optim.AdamW semantics (exp126a...py:103-108,293): decoupled decay, bias correction."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k]
            m = state.setdefault("m." + k, torch.zeros_like(p))
            v = state.setdefault("v." + k, torch.zeros_like(p))
            p.mul_(1 - lr * weight_decay)
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


PARAM_SUFFIX_BUFFERS = ("running_mean", "running_var", "num_batches_tracked")


def split_state(sd):
    """Clone a state_dict into (all tensors, names of trainable ones) with requires_grad set."""
    out, names = {}, []
    for k, v in sd.items():
        t = v.detach().clone()
        if not k.endswith(PARAM_SUFFIX_BUFFERS):
            t.requires_grad_(True)
            names.append(k)
        out[k] = t
    return out, names
