"""Experiment -> (model class, constructor kwargs) table.

Restates the ``model_params`` dicts of the reference's experiment scripts
(experiments/Exp1_SectionIV-B/exp*.py:73-86 each).  The
"tiny" entries are build-side reductions of the same classes used to keep CPU
parity tests fast; they exist in no experiment script.
"""

_COMMON = dict(n_chan_input=6, n_bins_in=216, n_bins_out=72, a_lrelu=0.3, p_dropout=0.2)


def _c(cls, lr=1e-3, **kw):
    d = dict(_COMMON)
    d.update(kw)
    return {"cls": cls, "kwargs": d, "lr": lr}


CONFIGS = {
    # paper name      class                                   kwargs
    "CNN:XS": _c("basic_cnn_segm_sigmoid", n_chan_layers=[20, 20, 10, 1]),                       # exp126a
    "CNN:S": _c("basic_cnn_segm_sigmoid", n_chan_layers=[100, 100, 50, 10]),                     # exp126b
    "CNN:M": _c("basic_cnn_segm_sigmoid", n_chan_layers=[250, 150, 100, 100]),                   # exp126c
    "CNN:L": _c("basic_cnn_segm_sigmoid", n_chan_layers=[280, 180, 120, 100]),                   # exp126d
    "DCNN:S": _c("deep_cnn_segm_sigmoid", n_chan_layers=[20, 20, 10, 1], n_prefilt_layers=5, residual=False),  # exp127a
    "DCNN:M": _c("deep_cnn_segm_sigmoid", lr=2e-4, n_chan_layers=[40, 40, 30, 10], n_prefilt_layers=5, residual=False),  # exp127b
    "DCNN:L": _c("deep_cnn_segm_sigmoid", lr=2e-4, n_chan_layers=[70, 70, 50, 10], n_prefilt_layers=5, residual=False),  # exp127c
    "DRCNN:S": _c("deep_cnn_segm_sigmoid", n_chan_layers=[20, 20, 10, 1], n_prefilt_layers=5, residual=True),  # exp128a
    "DRCNN:M": _c("deep_cnn_segm_sigmoid", lr=2e-4, n_chan_layers=[40, 40, 30, 10], n_prefilt_layers=5, residual=True),  # exp128b
    "DRCNN:L": _c("deep_cnn_segm_sigmoid", lr=2e-4, n_chan_layers=[70, 70, 50, 10], n_prefilt_layers=5, residual=True),  # exp128c
    "Unet:S": _c("simple_u_net_largekernels", n_chan_layers=[64, 30, 20, 10], scalefac=8),       # exp160d2
    "Unet:M": _c("simple_u_net_largekernels", n_chan_layers=[128, 100, 80, 50], scalefac=8),     # exp160g
    "Unet:L": _c("simple_u_net_largekernels", n_chan_layers=[128, 150, 100, 80], scalefac=4),    # exp160e3
    "Unet:XL": _c("simple_u_net_largekernels", n_chan_layers=[128, 180, 150, 100], scalefac=2),  # exp160f
    "SAUnet:M": _c("simple_u_net_doubleselfattn", n_chan_layers=[64, 30, 20, 10], scalefac=8, embed_dim=64,
                   num_heads=8, mlp_dim=1024, pos_encoding="sinusoidal"),                        # exp180b
    "SAUnet:L": _c("simple_u_net_doubleselfattn", n_chan_layers=[128, 80, 50, 30], scalefac=4, embed_dim=128,
                   num_heads=8, mlp_dim=8192, pos_encoding="sinusoidal"),                        # exp180d
    "SAUnet:XL": _c("simple_u_net_doubleselfattn", n_chan_layers=[128, 200, 150, 150], scalefac=2, embed_dim=256,
                    num_heads=8, mlp_dim=8192, pos_encoding="sinusoidal"),                       # exp180e
    "SAUnet:XXL": _c("simple_u_net_doubleselfattn", n_chan_layers=[128, 200, 150, 150], scalefac=4, embed_dim=128,
                     num_heads=8, mlp_dim=8192, pos_encoding="sinusoidal"),                      # exp180f
    "SAUSnet:M": _c("simple_u_net_doubleselfattn_twolayers", n_chan_layers=[64, 30, 20, 10], scalefac=8,
                    embed_dim=64, num_heads=8, mlp_dim=512, pos_encoding="sinusoidal"),          # exp181b
    "SAUSnet:L": _c("simple_u_net_doubleselfattn_twolayers", n_chan_layers=[128, 80, 50, 30], scalefac=4,
                    embed_dim=128, num_heads=8, mlp_dim=4096, pos_encoding="sinusoidal"),        # exp181d
    "SAUSnet:XL": _c("simple_u_net_doubleselfattn_twolayers", n_chan_layers=[128, 200, 150, 150], scalefac=4,
                     embed_dim=128, num_heads=8, mlp_dim=8192, pos_encoding="sinusoidal"),       # exp181f
    "SAUSnet:XXL": _c("simple_u_net_doubleselfattn_twolayers", n_chan_layers=[128, 200, 150, 150], scalefac=2,
                      embed_dim=256, num_heads=8, mlp_dim=8192, pos_encoding="sinusoidal"),      # exp181e
    "BLUnet:M": _c("u_net_blstm_varlayers", n_chan_layers=[64, 30, 20, 10], scalefac=16, embed_dim=416,
                   hidden_size=208, lstm_depth=1, lstm_number=1),                                # exp186b
    "BLUnet:L": _c("u_net_blstm_varlayers", n_chan_layers=[128, 80, 50, 30], scalefac=8, embed_dim=832,
                   hidden_size=416, lstm_depth=1, lstm_number=2),                                # exp186d
    "BLUnet:XXL": _c("u_net_blstm_varlayers", n_chan_layers=[128, 200, 150, 150], scalefac=4, embed_dim=1664,
                     hidden_size=832, lstm_depth=1, lstm_number=1),                              # exp186e
    "PUnet:M": _c("simple_u_net_polyphony_classif_softmax", n_chan_layers=[128, 100, 80, 50], scalefac=8,
                  num_polyphony_steps=24),                                                       # exp195g
    "PUnet:L": _c("simple_u_net_polyphony_classif_softmax", n_chan_layers=[128, 150, 100, 80], scalefac=4,
                  num_polyphony_steps=24),                                                       # exp195e3
    "PUnet:XL": _c("simple_u_net_polyphony_classif_softmax", n_chan_layers=[128, 180, 150, 100], scalefac=2,
                   num_polyphony_steps=24),                                                      # exp195f
    # ---- build-side reductions for fast parity tests (same classes, small channel counts)
    "tiny:CNN": _c("basic_cnn_segm_sigmoid", n_chan_layers=[6, 8, 6, 4]),
    "tiny:DRCNN": _c("deep_cnn_segm_sigmoid", n_chan_layers=[6, 8, 6, 4], n_prefilt_layers=3, residual=True),
    "tiny:Unet": _c("simple_u_net_largekernels", n_chan_layers=[8, 8, 6, 4], scalefac=16),
    "tiny:SAUnet": _c("simple_u_net_doubleselfattn", n_chan_layers=[8, 8, 6, 4], scalefac=16, embed_dim=32,
                      num_heads=8, mlp_dim=64, pos_encoding="sinusoidal"),
    "tiny:SAUnet-res": _c("simple_u_net_doubleselfattn", n_chan_layers=[8, 8, 6, 4], scalefac=16, embed_dim=32,
                          num_heads=4, mlp_dim=48, pos_encoding=None, residual=True),
    "tiny:SAUnet-alt": _c("simple_u_net_doubleselfattn", n_chan_layers=[8, 8, 6, 4], scalefac=16, embed_dim=32,
                          num_heads=4, mlp_dim=48, pos_encoding=None, alt_order=True, residual=True),
    "tiny:SAUSnet": _c("simple_u_net_doubleselfattn_twolayers", n_chan_layers=[8, 8, 6, 4], scalefac=16,
                       embed_dim=32, num_heads=8, mlp_dim=64, pos_encoding="sinusoidal"),
    "tiny:BLUnet": _c("u_net_blstm_varlayers", n_chan_layers=[8, 8, 6, 4], scalefac=16, embed_dim=416,
                      hidden_size=208, lstm_depth=1, lstm_number=2),
    "tiny:PUnet": _c("simple_u_net_polyphony_classif_softmax", n_chan_layers=[8, 8, 6, 4], scalefac=16,
                     num_polyphony_steps=24),
}

# BASELINE.json "configs" -> (paper name, batch) for bench/parity sizing
BASELINE_CONFIGS = [("CNN:XS", 8), ("DRCNN:L", 64), ("Unet:L", 128), ("SAUnet:L", 256), ("BLUnet:XXL", 256)]

# algorithmic train-step GFLOP per patch at T=75 (fwd+dgrad+wgrad)
TRAIN_GFLOP_PER_PATCH = {"DRCNN:L": 436.32, "Unet:L": 88.25, "SAUnet:L": 86.67, "BLUnet:XXL": 91.01,
                         "PUnet:XL": 243.93}
# the same at T=174 (100 output frames per patch, SURVEY 8(d)'s secondary shape): 6 x forward GMAC (SURVEY's table) minus
# 2 x the first convolution's (no input gradient): SAUnet:L 6 x 36.071 - 2 x 0.8118
TRAIN_GFLOP_PER_PATCH_T174 = {"SAUnet:L": 214.80}
FWD_GMAC_PER_PATCH = {"CNN:XS": 0.458, "DRCNN:L": 73.230, "Unet:L": 14.825, "SAUnet:L": 14.562,
                      "BLUnet:XXL": 15.285, "PUnet:XL": 40.888}

# tiny instances of the U-Net variants no experiment uses (oracle/make_goldens_variants.py, tests/test_gpu_variants.py)
_VT = dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72, scalefac=16)
_VA = dict(embed_dim=32, num_heads=4, mlp_dim=24)
VARIANT_CONFIGS = {
    "simple_u_net": dict(_VT),
    "simple_u_net_selfattn": dict(_VT, **_VA),
    "simple_u_net_sixselfattn": dict(_VT, **_VA, pos_encoding="sinusoidal"),
    "simple_u_net_doubleselfattn_alllayers": dict(_VT, **_VA),
    "simple_u_net_doubleselfattn_varlayers": dict(_VT, **_VA, self_attn_depth=2, self_attn_number=2, pos_encoding="sinusoidal"),
    "simple_u_net_polyphony_classif": dict(_VT, num_polyphony_steps=24),
    "simple_u_net_doubleselfattn_polyphony": dict(_VT, **_VA),
    "simple_u_net_doubleselfattn_polyphony_classif": dict(_VT, **_VA, num_polyphony_steps=24),
    # time_embed_dim = 72 bins x n_chan_layers[1]; n_chan_layers[1] == n_chan_layers[2] (the class runs with nothing else)
    # (16 heads: the head dimension of the time layers, 288 / 16 = 18, has to stay within the attention kernels' 32)
    "simple_u_net_doubleselfattn_transenc": dict(n_chan_layers=[8, 4, 4, 2], n_bins_out=72, scalefac=16, embed_dim=32,
                                                 num_heads=16, mlp_dim=24, self_attn_depth=1, self_attn_number=1,
                                                 time_embed_dim=288, pos_encoding="sinusoidal"),
    # scalefac 16: channels 1 / 3 / 9 / 27 / 108, channels x bins = 216 on every level = embed_dim (8 heads of 27)
    "u_net_temporal_selfattn_varlayers": dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72, scalefac=16, embed_dim=216,
                                              num_heads=8, mlp_dim=24, self_attn_depth=2, self_attn_number=2,
                                              pos_encoding="sinusoidal"),
    "u_net_temporal_blstm_varlayers": dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72, scalefac=16, embed_dim=216,
                                           hidden_size=108, lstm_depth=2, lstm_number=1),
    "freq_u_net_selfattn": dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72, scalefac=4, embed_dim=24, num_heads=4, mlp_dim=20),
    "freq_u_net_doubleselfattn": dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72, scalefac=4, embed_dim=24, num_heads=4,
                                      mlp_dim=20),
    "basic_cnn": dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72),
    "basic_cnn_pool": dict(n_chan_layers=[8, 6, 5, 4], n_bins_out=72),
    "basic_cnn_segm_logsoftmax": dict(n_chan_layers=[8, 6, 5, 4], n_ch_out=3, n_bins_out=72),
    "basic_cnn_segm_blank_logsoftmax": dict(n_chan_layers=[8, 6, 5, 4], n_ch_out=3, n_bins_out=72),
}
