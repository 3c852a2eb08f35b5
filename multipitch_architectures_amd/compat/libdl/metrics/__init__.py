"""`libdl.metrics` import surface (libdl/metrics/__init__.py:1-2 of the reference), served by the HIP build."""
from multipitch_architectures_amd.metrics import calculate_single_measure, calculate_eval_measures  # noqa: F401
from multipitch_architectures_amd.metrics.monitoring import early_stopping  # noqa: F401


def calculate_mpe_measures_mireval(*a, **kw):
    raise NotImplementedError("the mir_eval based measures (eval_metrics.py:159-193) are not built: mir_eval and librosa "
                              "are third-party packages outside this path")
