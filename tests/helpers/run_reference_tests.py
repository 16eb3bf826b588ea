"""Runs ONE of the reference's own test files (/root/reference/tests/test_*.py), UNMODIFIED, against the PRODUCT:

  * `pilco`, `pilco.models`, `pilco.controllers`, `pilco.rewards` resolve to pilco_amd (the drop-in import surface);
  * `oct2py` resolves to a stand-in whose Octave session answers gp0 / gp1 / gp2 / pred / conlin / gSin / reward with
    oracle/matlab_path.py, the transliteration of the reference's tests/Matlab Code/*.m (Octave is not installed);
  * `gpflow` / `tensorflow` resolve to the two names the test files touch (config.default_float; an unused import).

    python tests/helpers/run_reference_tests.py <test file name> [--standin]

--standin installs the CPU stand-in context (a box without a GPU: a dry run of the product's Python layer against the
oracle); without it the product's device path answers (needs an MI355X).  Exit code 0 = every test function passed.
TEST INFRASTRUCTURE; runs in its own process because it aliases module names."""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
REF_TESTS = "/root/reference/tests"

import numpy as np  # noqa: E402


class Struct(dict):
    """oct2py.io.Struct: attribute access on a dict."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


class _Octave:
    def __init__(self, *a, **k):
        self.logger = types.SimpleNamespace(setLevel=lambda *a, **k: None)

    def addpath(self, path):
        pass

    @staticmethod
    def _np(v):
        return np.asarray(v.numpy() if hasattr(v, "numpy") else v, np.float64)

    def gp0(self, g, m, s, nout=3):
        from oracle import matlab_path as mp
        return mp.gp0(self._np(g.inputs), self._np(g.targets), self._np(g.hyp), self._np(m), self._np(s))

    def gp1(self, g, m, s, nout=3):
        from oracle import matlab_path as mp
        return mp.gp1(self._np(g.inputs), self._np(g.targets), self._np(g.hyp), self._np(g.induce), self._np(m), self._np(s))

    def gp2(self, g, m, s, nout=3):
        from oracle import matlab_path as mp
        return mp.gp2(self._np(g.inputs), self._np(g.targets), self._np(g.hyp), self._np(m), self._np(s))

    def pred(self, policy, plant, dyn, m, s, H, nout=2, verbose=False):
        from oracle import matlab_path as mp
        return mp.pred(self._np(m), self._np(s), int(H), self._np(dyn.inputs), self._np(dyn.targets), self._np(dyn.hyp),
                       self._np(policy.p.w), self._np(policy.p.b), self._np(policy.maxU))

    def conlin(self, policy, m, s, nout=3):
        from oracle import matlab_path as mp
        return mp.conlin(self._np(policy.p.w), self._np(policy.p.b), self._np(m), self._np(s))

    def gSin(self, m, s, e, nout=3):
        from oracle import matlab_path as mp
        return mp.gSin(self._np(m), self._np(s), e)

    def reward(self, m, s, t, W, nout=4):
        from oracle import matlab_path as mp
        mu, sr = mp.reward(self._np(m), self._np(s), self._np(t), self._np(W))
        return mu, None, None, sr


def install_aliases():
    import pilco_amd
    import pilco_amd.controllers
    import pilco_amd.models
    import pilco_amd.models.mgpr
    import pilco_amd.models.pilco
    import pilco_amd.models.smgpr
    import pilco_amd.rewards
    for name in ("", ".models", ".models.mgpr", ".models.smgpr", ".models.pilco", ".controllers", ".rewards"):
        sys.modules["pilco" + name] = sys.modules["pilco_amd" + name]
    oct2py = types.ModuleType("oct2py")
    oct2py.Oct2Py = _Octave
    oct2py.io = types.SimpleNamespace(Struct=Struct)
    oct2py.get_log = lambda name=None: types.SimpleNamespace(setLevel=lambda *a, **k: None)
    sys.modules["oct2py"] = oct2py
    gpflow = types.ModuleType("gpflow")
    gpflow.config = types.SimpleNamespace(default_float=lambda: np.float64)
    sys.modules["gpflow"] = gpflow
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")


def main():
    name = sys.argv[1]
    if "--standin" in sys.argv:
        from helpers.cpu_standin_context import CpuStandInContext
        from pilco_amd import _lib
        _lib.set_context(CpuStandInContext())
    install_aliases()
    os.chdir("/root/reference")      # the test files build the Matlab path from the working directory
    spec = importlib.util.spec_from_file_location("reference_" + name[:-3], os.path.join(REF_TESTS, name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.__file__.startswith(REF_TESTS)
    ran = 0
    for fn in sorted(n for n in dir(mod) if n.startswith("test_") and callable(getattr(mod, n))):
        getattr(mod, fn)()
        print("PASSED", name, fn, flush=True)
        ran += 1
    assert ran > 0
    return 0


if __name__ == "__main__":
    sys.exit(main())
