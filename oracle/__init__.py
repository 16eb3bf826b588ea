"""CPU oracle for the PILCO moment-matching path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU stand-in.  The product
path (``pilco_amd``) never imports this package and fails loudly when the HIP
library is missing.

Two independent float64 NumPy restatements live here:

* ``oracle.tf_path``     -- follows the reference's TensorFlow/GPflow code
  op-for-op (pilco/models/mgpr.py, smgpr.py, pilco.py, controllers.py,
  rewards.py).
* ``oracle.matlab_path`` -- follows the MATLAB PILCO v0.9 routines the
  reference's own tests use as ground truth (tests/Matlab Code/*.m).

Pinning status: neither TensorFlow/GPflow nor Octave can be
executed in the build image, and the reference's tests store no golden vectors
(they call Octave live).  The oracle is therefore pinned by (1) agreement of
the two independent restatements to <=1e-9 relative on every reference test
configuration, (2) an independent Gauss-Hermite quadrature check of the
moment-matching integrals (``oracle.quadrature``).  Against an *executed*
reference the parity is unpinned, and this header says so on purpose.
"""
