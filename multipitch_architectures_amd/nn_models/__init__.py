"""Drop-in for ``libdl.nn_models`` (the reference's import surface, libdl/nn_models/__init__.py:1-10).

The 7 model classes every experiment script instantiates, the building blocks they are made of and every further
variant the reference exports and can construct (the other U-Nets, ``basic_cnn``, ``basic_cnn_pool``, the two log-softmax
CNNs, and -- round 4 -- the time-axis transformer layer, the temporal and frequency U-Nets) run on hand-written gfx950
kernels: 29 of the 32 exported names.  The remaining three cannot be constructed upstream either (SURVEY.md Appendix
C.7); they are kept importable and raise the reference's own exception class on construction.
"""
from .basic_cnns import (basic_cnn, basic_cnn_pool, basic_cnn_segm_blank_logsoftmax, basic_cnn_segm_logsoftmax, basic_cnn_segm_sigmoid,
                         deep_cnn_segm_sigmoid)
from .unet_cnns import (blstm_temporal_enc_layer, double_conv, freq_u_net_doubleselfattn, freq_u_net_selfattn, simple_u_net,
                        simple_u_net_doubleselfattn,
                        simple_u_net_doubleselfattn_alllayers, simple_u_net_doubleselfattn_polyphony,
                        simple_u_net_doubleselfattn_polyphony_classif, simple_u_net_doubleselfattn_transenc,
                        simple_u_net_doubleselfattn_twolayers,
                        simple_u_net_doubleselfattn_varlayers, simple_u_net_largekernels, simple_u_net_polyphony_classif,
                        simple_u_net_polyphony_classif_softmax, simple_u_net_selfattn, simple_u_net_sixselfattn,
                        transformer_enc_layer, transformer_temporal_enc_layer, u_net_blstm_varlayers,
                        u_net_temporal_blstm_varlayers, u_net_temporal_selfattn_varlayers, unet_up_concat_padding)

BUILT = ["basic_cnn_segm_sigmoid", "deep_cnn_segm_sigmoid", "double_conv", "unet_up_concat_padding",
         "transformer_enc_layer", "blstm_temporal_enc_layer", "simple_u_net_largekernels",
         "simple_u_net_doubleselfattn", "simple_u_net_doubleselfattn_twolayers", "u_net_blstm_varlayers",
         "simple_u_net_polyphony_classif_softmax",
         # re-compositions of the same blocks that no experiment script uses (round 3; goldens tests/golden/xcls-*.npz)
         "simple_u_net", "simple_u_net_selfattn", "simple_u_net_sixselfattn", "simple_u_net_doubleselfattn_alllayers",
         "simple_u_net_doubleselfattn_varlayers", "simple_u_net_polyphony_classif",
         "simple_u_net_doubleselfattn_polyphony", "simple_u_net_doubleselfattn_polyphony_classif", "basic_cnn_pool",
         "basic_cnn_segm_logsoftmax", "basic_cnn_segm_blank_logsoftmax", "basic_cnn",
         # round 4: the time-axis transformer layer and the U-Net that reduces time with it
         "transformer_temporal_enc_layer", "simple_u_net_doubleselfattn_transenc",
         # ... and the U-Nets with (2,3) pooling / upsampling and time-axis transformer / BiLSTM layers on their skips
         "u_net_temporal_selfattn_varlayers", "u_net_temporal_blstm_varlayers",
         # ... and the frequency U-Nets (SELU, MaxUnpool2d with transferred pooling indices)
         "freq_u_net_selfattn", "freq_u_net_doubleselfattn"]

# The three exported names that cannot be constructed upstream either: the reference's
# single_conv.__init__ reads `mid_channels` before assigning it (unet_cnns.py:18 -> UnboundLocalError) and freq_u_net /
# freq_u_net_bottomstack call an undefined `single_conv_SELU` (unet_cnns.py:1558, 1628 -> NameError).  They stay importable
# and fail at construction with the exception class the reference raises.
NOT_BUILT = ["single_conv", "freq_u_net", "freq_u_net_bottomstack"]
_UPSTREAM_ERROR = {"single_conv": (UnboundLocalError, "local variable 'mid_channels' referenced before assignment"),
                   "freq_u_net": (NameError, "name 'single_conv_SELU' is not defined"),
                   "freq_u_net_bottomstack": (NameError, "name 'single_conv_SELU' is not defined")}


def _not_constructible(name):
    exc, msg = _UPSTREAM_ERROR[name]

    def __init__(self, *args, **kwargs):
        raise exc(f"{msg} -- libdl.nn_models.{name} cannot be constructed in the reference either (SURVEY.md Appendix C.7); "
                  "there is nothing to build")
    return type(name, (object,), {"__init__": __init__, "__doc__": f"the reference's {name}: raises at construction upstream too"})


for _n in NOT_BUILT:
    globals()[_n] = _not_constructible(_n)

__all__ = BUILT + NOT_BUILT
