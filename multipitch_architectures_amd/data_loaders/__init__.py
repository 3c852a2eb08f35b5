from .hcqt_datasets import (ContextLoader, dataset_context, dataset_context_measuresegm, dataset_context_segm,  # noqa: F401
                            dataset_context_segm_pitch, dataset_context_segm_widetarget)
