"""`libdl.data_loaders` import surface (libdl/data_loaders/__init__.py:1-2 of the reference), served by the HIP build."""
from multipitch_architectures_amd.data_loaders import (dataset_context, dataset_context_measuresegm, dataset_context_segm,  # noqa: F401
                                                       dataset_context_segm_pitch, dataset_context_segm_widetarget)
