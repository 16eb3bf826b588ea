"""pytest plug-in for a box WITHOUT a GPU: installs the CPU stand-in context (tests/helpers/cpu_standin_context.py) as the
process-wide default, so that the GPU parity tests that only go through the Python layer can be DRY-RUN against the oracle --
a check of the tests' own logic and of the product's host code before spending GPU minutes, never a substitute for the GPU
run (the kernels are not involved; tests that touch the C ABI directly or probe the device fail with AttributeError here).

    PYTHONPATH=tests python -m pytest tests/test_gpu_parity.py -p helpers.standin_plugin -q -k "optimize or safe or sparse"
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    from helpers.cpu_standin_context import CpuStandInContext
    from pilco_amd import _lib
    _lib.set_context(CpuStandInContext())
