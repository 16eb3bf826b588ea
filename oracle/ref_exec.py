"""Execute the reference's own source (/root/reference/pilco/*.py, unmodified).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``load()`` injects the
stand-ins of ``oracle/refshim.py`` for tensorflow / tensorflow_probability /
gpflow into ``sys.modules``, imports the reference package ``pilco`` (and, on
request, ``safe_pilco_extension``) from where it lies under /root/reference,
and returns the imported modules in a namespace.  Nothing is copied: the
reference files are read by the Python import system from their own location.

/root/reference exists only in the build container.  ``available()`` says
whether it is there; the tests that execute the reference skip themselves
otherwise, and the committed fixtures under tests/golden/ (written by
``python -m oracle.gen_golden`` from these executed outputs) are what travels
to the GPU box.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PILCO_REFERENCE_ROOT", "/root/reference")

_cached = None


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "pilco", "models", "mgpr.py"))


def load(safe: bool = False):
    """-> namespace(tf, gpflow, pilco, MGPR, SMGPR, PILCO, controllers, rewards[, safe_pilco, rewards_safe])."""
    global _cached
    if _cached is not None and (not safe or hasattr(_cached, "safe_pilco")):
        return _cached
    if not available():
        raise RuntimeError(f"reference source not found under {REFERENCE_ROOT}")
    from . import refshim

    shim = refshim.make_modules()
    taken = [k for k in list(shim) + ["pilco", "safe_pilco_extension"] if k in sys.modules]
    if taken and _cached is None:
        raise RuntimeError(f"refusing to shadow already-imported modules: {taken}")
    saved_path = list(sys.path)
    names = list(shim)
    try:
        sys.modules.update(shim)
        sys.path.insert(0, REFERENCE_ROOT)
        # pandas >= 2 rejects the ambiguous option name the reference passes (pilco.py:67)
        import pandas as pd
        _orig = pd.set_option

        def _set_option(*a, **k):
            if a and a[0] == "precision":
                a = ("display.precision",) + tuple(a[1:])
            return _orig(*a, **k)
        pd.set_option = _set_option

        pilco = importlib.import_module("pilco")
        ns = types.SimpleNamespace(
            tf=shim["tensorflow"], gpflow=shim["gpflow"], shim=refshim, pilco=pilco,
            MGPR=pilco.models.MGPR, SMGPR=pilco.models.SMGPR, PILCO=pilco.models.PILCO,
            controllers=pilco.controllers, rewards=pilco.rewards, mgpr_module=pilco.models.mgpr,
        )
        if safe:
            ns.safe_pilco = importlib.import_module("safe_pilco_extension.safe_pilco")
            ns.rewards_safe = importlib.import_module("safe_pilco_extension.rewards_safe")
        for mod in list(sys.modules):
            if mod == "pilco" or mod.startswith("pilco.") or mod.startswith("safe_pilco_extension"):
                f = getattr(sys.modules[mod], "__file__", "") or ""
                assert f.startswith(REFERENCE_ROOT), f"{mod} was not imported from the reference tree: {f}"
    finally:
        sys.path[:] = saved_path
        # the reference modules keep their own bindings of tf/gpflow; do not leave fake
        # 'tensorflow' etc. importable by unrelated code in the same process
        for k in names:
            sys.modules.pop(k, None)
    _cached = ns
    return ns


def to_np(x):
    """Tensor / Parameter / ndarray -> float64 ndarray."""
    import numpy as np
    if hasattr(x, "numpy"):
        return np.array(x.numpy(), dtype=np.float64)
    return np.asarray(x, np.float64)
