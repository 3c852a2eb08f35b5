from .hcqt_datasets import dataset_context, dataset_context_segm, ContextLoader  # noqa: F401
