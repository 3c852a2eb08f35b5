"""`libdl.nn_models` import surface (libdl/nn_models/__init__.py:1-10 of the reference), served by the HIP build."""
from multipitch_architectures_amd.nn_models import *  # noqa: F401,F403
from multipitch_architectures_amd.nn_models import __all__  # noqa: F401
