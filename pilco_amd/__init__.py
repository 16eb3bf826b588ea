"""pilco_amd: MI355X-native PILCO moment-matching rollout and GP factorisation.

Mirrors the import surface of the reference package (``pilco/__init__.py``):
``models`` (MGPR, SMGPR, PILCO), ``controllers``, ``rewards``.  All arithmetic
runs in libpilco_hip.so (hand-written gfx950 HIP kernels) through ctypes; there
is no CPU fallback -- compute calls raise if the library or a GPU is missing.
"""
from . import controllers, models, rewards  # noqa: F401
from ._lib import Context, NotPositiveDefiniteError, PilcoError, get_context, set_context  # noqa: F401
from .params import Parameter, set_trainable  # noqa: F401  (what the reference's scripts take from gpflow)
from . import training  # noqa: F401  (SciPy's optimisers load HERE, as gpflow's do when the reference's scripts import it -- not inside the first optimize_models call, where 0.2 s of import time used to be charged to the learning loop)

__all__ = ["models", "controllers", "rewards", "Context", "get_context", "set_context", "PilcoError",
           "NotPositiveDefiniteError", "Parameter", "set_trainable"]
