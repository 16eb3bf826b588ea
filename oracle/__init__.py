"""CPU oracle for the PILCO moment-matching path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU stand-in.  The product
path (``pilco_amd``) never imports this package and fails loudly when the HIP
library is missing.

What lives here:

* ``oracle.ref_exec`` + ``oracle.refshim`` -- the reference's OWN SOURCE
  (/root/reference/pilco/models/{mgpr,smgpr,pilco}.py, controllers.py,
  rewards.py) imported unmodified and executed on top of a torch-CPU-float64
  stand-in for the ~40 tensorflow ops and the slice of GPflow it uses.
  ``python -m oracle.gen_golden`` writes tests/golden/*.npz from these executed
  outputs (every fixture carries a ``provenance`` line).
* ``oracle.tf_path``     -- NumPy restatement of the same code, op for op; it
  is what the GPU parity tests and bench.py's CPU baseline call at sizes that
  have no fixture.  tests/test_reference_exec.py holds it to the executed
  reference at <= 1e-9 relative on every reference test procedure.
* ``oracle.matlab_path`` -- transliteration of the MATLAB PILCO v0.9 routines
  the reference's own tests use as ground truth (tests/Matlab Code/*.m); the
  reference's assertions (rtol 1e-4 / 2e-4) are repeated against it.
* ``oracle.mp_truth``    -- 40-digit mpmath evaluation of one moment-matching
  step, the arbiter where float64 evaluations disagree (noise floor 1e-6).
* ``oracle.torch_path``  -- torch restatement for autograd (gradient oracle),
  held to reverse mode through the executed reference's training_loss.
* ``oracle.quadrature``  -- Gauss-Hermite check of the integrals (follows no
  reference code).

Pinning status: PINNED TO THE EXECUTED REFERENCE
SOURCE for everything under /root/reference.  What remains recollection is the
third-party arithmetic that is not under /root/reference and cannot be
installed here: the meaning of the tf.* ops (one-line mappings in refshim.py)
and GPflow 2.1's SquaredExponential.K / GPR / GPRFITC objectives / bijectors /
priors (refshim.py docstring).  Of those only SquaredExponential.K is on the
prediction path; it is cross-checked by the MATLAB route (maha.m) and the
40-digit evaluation.  The GP *training* objective (GPflow internals) stays
"parity unpinned" in the strict sense: no reference test or vector pins it.
"""
