"""Drop-in for ``libdl.nn_models`` (the reference's import surface, libdl/nn_models/__init__.py:1-10).

The 7 model classes every experiment script instantiates, the 4 building
blocks they are made of and 12 further variants the reference exports (8 U-Nets,
``basic_cnn``, ``basic_cnn_pool`` and the two log-softmax CNNs: re-compositions of the same blocks) run on hand-written gfx950 kernels.  The
remaining exported names (variants no experiment uses: the frequency U-Nets,
the temporal transformer / BiLSTM variants -- two of
them cannot even be constructed upstream) are kept
importable and raise ``NotImplementedError`` on construction.
"""
from .basic_cnns import (basic_cnn, basic_cnn_pool, basic_cnn_segm_blank_logsoftmax, basic_cnn_segm_logsoftmax, basic_cnn_segm_sigmoid,
                         deep_cnn_segm_sigmoid)
from .unet_cnns import (blstm_temporal_enc_layer, double_conv, simple_u_net, simple_u_net_doubleselfattn,
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
         "u_net_temporal_selfattn_varlayers", "u_net_temporal_blstm_varlayers"]

NOT_BUILT = ["single_conv", "freq_u_net", "freq_u_net_bottomstack",
             "freq_u_net_selfattn", "freq_u_net_doubleselfattn"]


def _not_built(name):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"libdl.nn_models.{name} is exported by the reference but used by none of its experiment scripts; "
            "it is not part of the MI355X hot path yet (see DESIGN.md, 'out of scope')")
    return type(name, (object,), {"__init__": __init__, "__doc__": f"placeholder for the reference's unused {name}"})


for _n in NOT_BUILT:
    globals()[_n] = _not_built(_n)

__all__ = BUILT + NOT_BUILT
