"""Drop-in shim: put `<repo>/multipitch_architectures_amd/compat` (and `<repo>`) on PYTHONPATH and the reference's
`from libdl.nn_models import ...`, `from libdl.data_loaders import dataset_context` and
`from libdl.metrics import calculate_eval_measures, early_stopping` resolve to the MI355X implementation."""
