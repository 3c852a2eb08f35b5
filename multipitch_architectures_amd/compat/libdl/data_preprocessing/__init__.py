"""`libdl.data_preprocessing` import surface (libdl/data_preprocessing/__init__.py:1-2 of the reference), served by the HIP
build: the note list -> piano roll conversion runs on the GPU; the HCQT functions run on the GPU as well (parity unpinned: SURVEY 8 f4)."""
from multipitch_architectures_amd.data_preprocessing import (compute_hopsize_cqt, compute_hcqt, compute_efficient_hcqt,  # noqa: F401
                                                             compute_annotation_array_nooverlap)
