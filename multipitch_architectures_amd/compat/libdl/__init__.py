"""Drop-in shim: put `<repo>/multipitch_architectures_amd/compat` (and `<repo>`) on PYTHONPATH and the reference's
`from libdl.nn_models import ...` resolves to the MI355X implementation.  Only `libdl.nn_models` is provided."""
