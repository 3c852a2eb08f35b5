"""`libdl.data_loaders` import surface (libdl/data_loaders/__init__.py:1-2 of the reference), served by the HIP build.
The three dataset classes no experiment of the paper uses are importable and raise on construction."""
from multipitch_architectures_amd.data_loaders import dataset_context, dataset_context_segm  # noqa: F401


def _not_built(name):
    class _Stub:
        def __init__(self, *a, **kw):
            raise NotImplementedError(f"libdl.data_loaders.{name} is not built (unused by the paper's experiments)")
    _Stub.__name__ = name
    return _Stub


dataset_context_segm_pitch = _not_built("dataset_context_segm_pitch")
dataset_context_segm_widetarget = _not_built("dataset_context_segm_widetarget")
dataset_context_measuresegm = _not_built("dataset_context_measuresegm")
