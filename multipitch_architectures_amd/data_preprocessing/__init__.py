from .hcqt import (compute_hopsize_cqt, compute_hcqt, compute_efficient_hcqt,  # noqa: F401
                   compute_annotation_array_nooverlap, annotation_array_nooverlap_device, efficient_hcqt_device,
                   estimate_tuning_device)
