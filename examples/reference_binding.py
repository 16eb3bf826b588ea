"""The binding a maintainer of nrontsis/PILCO would add (as pilco/models/_hip.py) to keep the reference's own classes
and run their arithmetic in libpilco_hip.so -- INTEGRATION.md section 2, as an executable file.

Only ctypes + NumPy; it touches nothing but the attributes the reference's objects already have
(mgpr.data / .lengthscales / .variance / .noise / .num_dims / .num_outputs, pilco.controller.W / .b / .max_action,
pilco.reward.W / .t, pilco.state_dim / .control_dim).  `patch(pilco_module)` swaps the two hot methods of the imported
reference package in place; tests/test_gpu_parity.py runs the functions below on stand-in objects with exactly those
attributes (the reference itself cannot be imported on the GPU box: no TensorFlow).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = C.CDLL(os.environ.get("PILCO_LIB", os.path.join(_HERE, "..", "pilco_amd", "libpilco_hip.so")))
_lib.pilco_last_error.restype = C.c_char_p
_dp = C.POINTER(C.c_double)
_p = lambda a: a.ctypes.data_as(_dp)
_f = lambda t: np.ascontiguousarray(np.asarray(t.numpy() if hasattr(t, "numpy") else t), np.float64)


class _Policy(C.Structure):   # struct pilco_policy (include/pilco_hip.h)
    _fields_ = [("kind", C.c_int), ("state_dim", C.c_int), ("control_dim", C.c_int),
                ("W", _dp), ("b", _dp), ("max_action", _dp), ("squash", C.c_int)]


class _Reward(C.Structure):   # struct pilco_reward_term
    _fields_ = [("kind", C.c_int), ("coef", C.c_double), ("W", _dp), ("t", _dp)]


_ctx = C.c_void_p()


def _context():
    if not _ctx:
        rc = _lib.pilco_ctx_create(0, C.byref(_ctx))
        if rc:
            raise RuntimeError("pilco_ctx_create failed (%d): no MI355X visible?" % rc)
    return _ctx


def _check(rc):
    if rc:
        raise RuntimeError(_lib.pilco_last_error(_context()).decode())


def sync_model(mgpr):
    """Call after MGPR.__init__ / set_data / optimize (pilco/models/mgpr.py:38-45, 47-75, 159-190)."""
    ctx = _context()
    X, Y = [_f(t) for t in mgpr.data]
    _check(_lib.pilco_gp_set_data(ctx, 0, _p(X), _p(Y), X.shape[0], X.shape[1], Y.shape[1]))
    ls, var, nz = _f(mgpr.lengthscales), _f(mgpr.variance).reshape(-1), _f(mgpr.noise).reshape(-1)
    _check(_lib.pilco_gp_set_hyp(ctx, 0, _p(ls), _p(var), _p(nz)))
    _check(_lib.pilco_gp_factorize(ctx, 0))             # mgpr.py:81-89, cached until data / hyper-parameters change


def predict_on_noisy_inputs(mgpr, m, s):
    """Replaces MGPR.predict_on_noisy_inputs (pilco/models/mgpr.py:77-149)."""
    D, E = mgpr.num_dims, mgpr.num_outputs
    m, s = _f(m), _f(s)
    M, S, V = np.empty((1, E)), np.empty((E, E)), np.empty((D, E))
    _check(_lib.pilco_gp_predict(_context(), 0, _p(m), _p(s), _p(M), _p(S), _p(V)))
    return M, S, V


def predict(pilco, m_x, s_x, n):
    """Replaces the tf.while_loop of PILCO.predict (pilco/models/pilco.py:118-136) for a LinearController and an
    ExponentialReward (the defaults of pilco.py:26-34)."""
    E, U = pilco.state_dim, pilco.control_dim
    W, b = _f(pilco.controller.W), _f(pilco.controller.b)
    e = np.ascontiguousarray(np.broadcast_to(np.asarray(pilco.controller.max_action, np.float64).ravel(), (U,)))
    pol = _Policy(1, E, U, _p(W), _p(b), _p(e), 1)      # PILCO_POLICY_LINEAR + squash_sin (controllers.py:46-58)
    Wr, t = _f(pilco.reward.W), _f(pilco.reward.t)
    rw = _Reward(1, 1.0, _p(Wr), _p(t))                 # PILCO_REWARD_EXPONENTIAL (rewards.py:19-51)
    m_x, s_x = _f(m_x), _f(s_x)
    M, S, R = np.empty((1, E)), np.empty((E, E)), np.zeros((1, 1))
    _check(_lib.pilco_rollout(_context(), C.byref(pol), C.byref(rw), 1, _p(m_x), _p(s_x), int(n), _p(M), _p(S), _p(R), None))
    return M, S, R


def patch(pilco_pkg):
    """pilco_pkg = the imported reference package: route its two hot methods through the library."""
    MGPR, PILCO = pilco_pkg.models.MGPR, pilco_pkg.models.PILCO

    def _mgpr_predict(self, m, s):
        sync_model(self)
        return predict_on_noisy_inputs(self, m, s)

    def _pilco_predict(self, m_x, s_x, n):
        sync_model(self.mgpr)
        return predict(self, m_x, s_x, n)
    MGPR.predict_on_noisy_inputs = _mgpr_predict
    PILCO.predict = _pilco_predict
