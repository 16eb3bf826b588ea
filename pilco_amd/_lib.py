"""ctypes binding of libpilco_hip.so (include/pilco_hip.h).

No PyTorch, no TensorFlow: plain ctypes + NumPy.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C pilco_amd/csrc``.  There is NO CPU
fallback: if the library or a GPU is missing every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PILCO_LIB", os.path.join(_HERE, "libpilco_hip.so"))

PILCO_OK = 0
STATUS_NAMES = {1: "PILCO_E_SHAPE", 2: "PILCO_E_NOT_PD", 3: "PILCO_E_HIP", 4: "PILCO_E_RCCL",
                5: "PILCO_E_STATE", 6: "PILCO_E_ALLOC"}
SLOT_DYNAMICS, SLOT_POLICY = 0, 1
POLICY_NONE, POLICY_LINEAR, POLICY_RBF = 0, 1, 2
REWARD_EXPONENTIAL, REWARD_LINEAR = 1, 2
COMM_ID_BYTES = 128
PEER_HANDLE_BYTES = 64

_dp = C.POINTER(C.c_double)


class PilcoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


class NotPositiveDefiniteError(PilcoError):
    """The reference raises tf.errors.InvalidArgumentError here (tests/test_cascade.py:22)."""


class PolicyStruct(C.Structure):
    _fields_ = [("kind", C.c_int), ("state_dim", C.c_int), ("control_dim", C.c_int),
                ("W", _dp), ("b", _dp), ("max_action", _dp), ("squash", C.c_int)]


class RewardTerm(C.Structure):
    _fields_ = [("kind", C.c_int), ("coef", C.c_double), ("W", _dp), ("t", _dp)]


# every symbol include/pilco_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
# pilco_seed_fn (include/pilco_hip.h): (user, H, E, traj [H+1][E+E*E], seeds [H+1][E+E*E]) -> None
SEED_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))
SIGNATURES = {
    "pilco_abi_version": (C.c_int, []),
    "pilco_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pilco_ctx_destroy": (C.c_int, [_vp]),
    "pilco_last_error": (C.c_char_p, [_vp]),
    "pilco_last_not_pd_output": (C.c_int, [_vp]),
    "pilco_set_pair_kernel": (C.c_int, [_vp, C.c_int]),
    "pilco_set_reverse_chain": (C.c_int, [_vp, C.c_int]),
    "pilco_debug_poison": (C.c_int, [_vp, C.c_int, C.c_int]),
    "pilco_selftest": (C.c_int, [_vp]),
    "pilco_set_fused_step": (C.c_int, [_vp, C.c_int]),
    "pilco_set_grad_mode": (C.c_int, [_vp, C.c_int]),
    "pilco_set_small_step": (C.c_int, [_vp, C.c_int]),
    "pilco_set_use_graph": (C.c_int, [_vp, C.c_int]),
    "pilco_set_inline_policy": (C.c_int, [_vp, C.c_int]),
    "pilco_gp_set_data": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int]),
    "pilco_gp_set_hyp": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "pilco_gp_set_inducing": (C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    "pilco_gp_gram": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp]),
    "pilco_gp_factorize": (C.c_int, [_vp, C.c_int]),
    "pilco_gp_nlml": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "pilco_gp_fitc_nlml": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _dp, _dp]),
    "pilco_gp_num_points": (C.c_int, [_vp, C.c_int]),
    "pilco_gp_get_factors": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "pilco_gp_set_factors": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "pilco_gp_predict": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "pilco_rollout": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                _dp, _dp, _dp, _dp]),
    "pilco_gp_predict_vjp": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "pilco_rollout_tape": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                     _dp, _dp, _dp, _dp, _dp]),
    "pilco_rollout_grad": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                     _dp, _dp, _dp]),
    "pilco_rollout_grad_rbf": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                         _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "pilco_rollout_grad_seeded": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                            SEED_FN, _vp, _dp, _dp, _dp]),
    "pilco_rollout_grad_rbf_seeded": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                                _dp, _dp, _dp, _dp, C.c_int, SEED_FN, _vp, _dp, _dp, _dp, _dp]),
    "pilco_propagate": (C.c_int, [_vp, C.POINTER(PolicyStruct), _dp, _dp, _dp, _dp]),
    "pilco_rollout_batch": (C.c_int, [_vp, C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]),
    "pilco_rollout_grad_batch": (C.c_int, [_vp, C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]),
    "pilco_rollout_grad_rbf_batch": (C.c_int, [_vp, C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                               _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "pilco_rollout_grad_batch_seeded": (C.c_int, [_vp, C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                                  SEED_FN, C.POINTER(C.c_void_p), _dp, _dp, _dp]),
    "pilco_rollout_grad_rbf_batch_seeded": (C.c_int, [_vp, C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                                      _dp, _dp, _dp, _dp, C.c_int, SEED_FN, C.POINTER(C.c_void_p), _dp, _dp, _dp, _dp]),
    "pilco_policy_action": (C.c_int, [_vp, C.POINTER(PolicyStruct), _dp, _dp, _dp, _dp, _dp]),
    "pilco_reward_eval": (C.c_int, [_vp, C.POINTER(RewardTerm), C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
    "pilco_rollout_timed": (C.c_int, [_vp, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp,
                                      C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "pilco_factorize_timed": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "pilco_set_pair_timing": (C.c_int, [_vp, C.c_int]),
    "pilco_get_pair_timing": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "pilco_debug_timestamps": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "pilco_debug_blocks": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.c_int]),
    "pilco_debug_buffer": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_long]),
    "pilco_debug_sk_boundary": (C.c_int, [C.c_int] * 8),
    "pilco_debug_sk_pair_waves": (C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_int)]),
    "pilco_comm_unique_id": (C.c_int, [_vp]),
    "pilco_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "pilco_shard_set": (C.c_int, [_vp, C.c_int, C.c_int]),
    "pilco_peer_export": (C.c_int, [_vp, _vp]),
    "pilco_peer_attach": (C.c_int, [_vp, _vp, C.c_int]),
    "pilco_group_peer_attach": (C.c_int, [_vp, C.c_int]),
    "pilco_peer_detach": (C.c_int, [_vp]),
    "pilco_peer_attached": (C.c_int, [_vp]),
    "pilco_shard_owner_of_pair": (C.c_int, [_vp, C.c_int]),
    "pilco_shard_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "pilco_shard_pair_slot": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pilco_shard_output_slot": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pilco_gp_shard_pack": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "pilco_gp_shard_finish": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "pilco_rollout_grad_group": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]),
    "pilco_rollout_grad_rbf_group": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp, C.c_int,
                                               _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "pilco_rollout_group": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(PolicyStruct), C.POINTER(RewardTerm), C.c_int, _dp, _dp,
                            C.c_int, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "pilco_group_sync_model": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int]),
    "pilco_gp_beta_rows": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pilco_gp_beta_export": (C.c_int, [_vp, C.c_int, _dp]),
    "pilco_gp_beta_import": (C.c_int, [_vp, C.c_int, _dp]),
    "pilco_comm_rank": (C.c_int, [_vp]),
    "pilco_comm_size": (C.c_int, [_vp]),
    "pilco_comm_count": (C.c_int, [_vp]),
}

_lib = None


def load_library():
    """Load libpilco_hip.so and bind every declared symbol.  Raises if the HIP
    extension has not been built -- there is deliberately no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C pilco_amd/csrc). "
            "pilco_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class Context:
    """One GPU context (pilco_ctx).  All arrays in/out are host NumPy float64."""

    def __init__(self, device=None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("PILCO_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = _vp()
        rc = self.lib.pilco_ctx_create(int(device), C.byref(h))
        if rc != PILCO_OK or not h:
            raise PilcoError(rc, f"cannot create a HIP context on device {device} "
                                 "(no MI355X visible?). pilco_amd has no CPU fallback.")
        self.h = h
        self.device = int(device)
        self._keep = []
        # which Python model last pushed its data / hyper-parameters into each device slot: several MGPR / SMGPR /
        # RbfController instances may share one context, but a slot holds ONE model at a time (see MGPR._sync)
        self._slot_owner = {}
        # runtime knobs set on this context (method name -> arguments): context_for copies them onto the pooled siblings of
        # the default context; nranks / has_comm: a sharded context is never pooled
        self._settings = {}
        self.nranks, self.has_comm = 1, False

    def close(self):
        if getattr(self, "h", None):
            self.lib.pilco_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != PILCO_OK:
            msg = self.lib.pilco_last_error(self.h).decode("utf-8", "replace")
            if rc == 2:
                exc = NotPositiveDefiniteError(rc, msg)
                exc.output = int(self.lib.pilco_last_not_pd_output(self.h))   # which output's Gram matrix failed (-1: unknown)
                raise exc
            raise PilcoError(rc, msg)

    # ---- GP model
    def selftest(self):
        self._chk(self.lib.pilco_selftest(self.h))

    def set_fused_step(self, on):
        self._chk(self.lib.pilco_set_fused_step(self.h, 1 if on else 0))
        self._settings["set_fused_step"] = (on,)

    def set_small_step(self, on):
        """1 (default): models of at most 256 points run a horizon step as ONE launch (include/pilco_hip.h)."""
        self._chk(self.lib.pilco_set_small_step(self.h, 1 if on else 0))
        self._settings["set_small_step"] = (on,)

    def set_grad_mode(self, mode):
        """1 (default): Jacobian tape; 0: plain tape + per-step device adjoint (include/pilco_hip.h)."""
        self._chk(self.lib.pilco_set_grad_mode(self.h, int(mode)))
        self._settings["set_grad_mode"] = (mode,)

    def use_graph(self, on):
        self._chk(self.lib.pilco_set_use_graph(self.h, 1 if on else 0))
        self._settings["use_graph"] = (on,)

    def set_inline_policy(self, on):
        """1 (default): small RbfControllers are evaluated inside the step's serial link; 0: own launches (include/pilco_hip.h)."""
        self._chk(self.lib.pilco_set_inline_policy(self.h, 1 if on else 0))
        self._settings["set_inline_policy"] = (on,)

    def debug_poison(self, slot, which):
        """Test aid: NaN-fill a factorisation buffer of the slot (0 = L^-1, 1 = iK, 2 = beta); include/pilco_hip_dev.h."""
        self._chk(self.lib.pilco_debug_poison(self.h, int(slot), int(which)))

    def set_reverse_chain(self, on_device):
        """1 (default): a LinearController's reverse chain runs on the device; 0: the host chain (include/pilco_hip_dev.h)."""
        self._chk(self.lib.pilco_set_reverse_chain(self.h, int(bool(on_device))))
        self._settings["set_reverse_chain"] = (on_device,)

    def set_pair_kernel(self, variant):
        self._chk(self.lib.pilco_set_pair_kernel(self.h, int(variant)))
        self._settings["set_pair_kernel"] = (variant,)

    def gp_set_data(self, slot, X, Y, owner=None):
        self._slot_owner[slot] = owner     # a direct caller (owner None) invalidates whatever a model believed about the slot
        X = _f64(X)
        Y = _f64(Y)
        if X.ndim != 2 or Y.ndim != 2 or X.shape[0] != Y.shape[0]:
            raise ValueError("data must be (X (N,D), Y (N,E))")
        self._chk(self.lib.pilco_gp_set_data(self.h, slot, _ptr(X), _ptr(Y), X.shape[0], X.shape[1], Y.shape[1]))

    def gp_set_hyp(self, slot, lengthscales, variance, noise, owner=None):
        self._slot_owner[slot] = owner
        ls, var, nz = _f64(lengthscales), _f64(variance).reshape(-1), _f64(noise).reshape(-1)
        self._chk(self.lib.pilco_gp_set_hyp(self.h, slot, _ptr(ls), _ptr(var), _ptr(nz)))

    def gp_set_inducing(self, slot, Z, owner=None):
        self._slot_owner[slot] = owner
        if Z is None:
            self._chk(self.lib.pilco_gp_set_inducing(self.h, slot, None, 0))
        else:
            Z = _f64(Z)
            self._chk(self.lib.pilco_gp_set_inducing(self.h, slot, _ptr(Z), Z.shape[0]))

    def gp_gram(self, slot, X1, X2, E):
        X1 = _f64(X1)
        X2a = None if X2 is None else _f64(X2)
        n2 = X1.shape[0] if X2a is None else X2a.shape[0]
        out = np.empty((E, X1.shape[0], n2))
        self._chk(self.lib.pilco_gp_gram(self.h, slot, _ptr(X1), X1.shape[0], _ptr(X2a), n2, _ptr(out)))
        return out

    def gp_factorize(self, slot):
        self._chk(self.lib.pilco_gp_factorize(self.h, slot))

    def gp_nlml(self, slot, D, E, want_grad=True):
        nlml = np.empty(E)
        grad = np.empty((E, D + 2)) if want_grad else None
        self._chk(self.lib.pilco_gp_nlml(self.h, slot, _ptr(nlml), _ptr(grad)))
        return nlml, grad

    def gp_fitc_nlml(self, slot, Z_all, D, E, want_grad=True):
        """GPRFITC negative log marginal likelihood per output and its gradients: (nlml (E), dhyp (E, D+2), dZ (E, M, D))."""
        Z_all = _f64(Z_all)
        M = Z_all.shape[1]
        Z_all = _f64(Z_all, (E, M, D))
        nlml = np.empty(E)
        gh = np.empty((E, D + 2)) if want_grad else None
        gz = np.empty((E, M, D)) if want_grad else None
        self._chk(self.lib.pilco_gp_fitc_nlml(self.h, slot, _ptr(Z_all), M, _ptr(nlml), _ptr(gh), _ptr(gz)))
        return nlml, gh, gz

    def gp_num_points(self, slot):
        return self.lib.pilco_gp_num_points(self.h, slot)

    def gp_get_factors(self, slot, E, want_iK=True):
        n = self.gp_num_points(slot)
        iK = np.empty((E, n, n)) if want_iK else None
        beta = np.empty((E, n))
        self._chk(self.lib.pilco_gp_get_factors(self.h, slot, _ptr(iK), _ptr(beta)))
        return iK, beta

    def gp_set_factors(self, slot, iK, beta, owner=None):
        self._slot_owner[slot] = owner
        iKa = None if iK is None else _f64(iK)
        beta = _f64(beta)
        self._chk(self.lib.pilco_gp_set_factors(self.h, slot, _ptr(iKa), _ptr(beta)))

    def gp_predict(self, slot, m, s, D, E):
        m = _f64(m, (D,))
        s = _f64(s, (D, D))
        M = np.empty((1, E))
        S = np.empty((E, E))
        V = np.empty((D, E))
        self._chk(self.lib.pilco_gp_predict(self.h, slot, _ptr(m), _ptr(s), _ptr(M), _ptr(S), _ptr(V)))
        return M, S, V

    # ---- policy / reward marshalling
    def _policy(self, spec):
        """spec: dict(kind, state_dim, control_dim, W, b, max_action, squash)."""
        keep = []
        p = PolicyStruct()
        p.kind = spec["kind"]
        p.state_dim = spec["state_dim"]
        p.control_dim = spec["control_dim"]
        p.squash = 1 if spec.get("squash", True) else 0
        U, E = p.control_dim, p.state_dim
        if spec["kind"] == POLICY_LINEAR:
            W = _f64(spec["W"], (U, E))
            b = _f64(spec["b"], (U,))
            keep += [W, b]
            p.W, p.b = _ptr(W), _ptr(b)
        ma = spec.get("max_action", None)
        if ma is not None and U > 0:
            ma = _f64(np.broadcast_to(np.asarray(ma, np.float64).reshape(-1) if np.ndim(ma) else np.float64(ma), (U,)))
            keep.append(ma)
            p.max_action = _ptr(ma)
        return p, keep

    def _rewards(self, terms, E):
        """terms: list of dict(kind, coef, W, t)."""
        arr = (RewardTerm * max(len(terms), 1))()
        keep = []
        for i, t in enumerate(terms):
            arr[i].kind = t["kind"]
            arr[i].coef = float(t.get("coef", 1.0))
            if t["kind"] == REWARD_EXPONENTIAL:
                W = _f64(t["W"], (E, E))
                tg = _f64(t["t"], (E,)) if t.get("t") is not None else None
                keep += [W, tg]
                arr[i].W, arr[i].t = _ptr(W), _ptr(tg)
            else:
                W = _f64(t["W"], (E,))
                keep.append(W)
                arr[i].W = _ptr(W)
        return arr, keep

    def rollout(self, policy, rewards, m0, S0, H, want_traj=False):
        E = policy["state_dim"]
        p, k1 = self._policy(policy)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (E,))
        S0 = _f64(S0, (E, E))
        mH = np.empty((1, E))
        SH = np.empty((E, E))
        rew = np.zeros((1, 1))
        traj = np.empty((H + 1, E + E * E)) if want_traj else None
        self._chk(self.lib.pilco_rollout(self.h, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H),
                                         _ptr(mH), _ptr(SH), _ptr(rew), _ptr(traj)))
        if want_traj:
            return mH, SH, rew, traj
        return mH, SH, rew

    def rollout_batch(self, policies, rewards, m0, S0, H):
        """B independent rollouts of the same model in flight together (pilco_rollout_batch): policies: list of B policy
        specs (kind NONE / LINEAR), m0 (B, E), S0 (B, E, E) -> mH (B, E), SH (B, E, E), reward (B,).  Each lane is
        bit-identical to its solo rollout."""
        B = len(policies)
        E = policies[0]["state_dim"]
        arr, keep = self._policy_array(policies)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (B, E))
        S0 = _f64(S0, (B, E, E))
        mH = np.empty((B, E))
        SH = np.empty((B, E, E))
        rew = np.zeros(B)
        self._chk(self.lib.pilco_rollout_batch(self.h, B, arr, r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(mH), _ptr(SH), _ptr(rew)))
        return mH, SH, rew

    def _policy_array(self, policies):
        arr = (PolicyStruct * len(policies))()
        keep = []
        for i, spec in enumerate(policies):
            p, k = self._policy(spec)
            keep.append((p, k))
            arr[i] = p
        return arr, keep

    @staticmethod
    def _lane_seed_callback(seed_fns, errors):
        """One pilco_seed_fn for the B lanes of a batch: the library hands lane i's callback the user word i + 1."""
        def cb(user, H, E, traj_p, seeds_p):
            n = (H + 1) * (E + E * E)
            out = np.ctypeslib.as_array(seeds_p, shape=(n,))
            try:
                traj = np.ctypeslib.as_array(traj_p, shape=(n,)).reshape(H + 1, E + E * E).copy()
                out[:] = np.asarray(seed_fns[int(user) - 1](traj), dtype=np.float64).reshape(n)
            except BaseException as exc:   # noqa: BLE001 -- re-raised by the caller
                errors.append(exc)
                out[:] = np.nan
        return SEED_FN(cb)

    lane_seeds = True   # rollout_grad_batch / rollout_grad_rbf_batch take per-lane seed callbacks (pilco_rollout_grad*_batch_seeded)

    def rollout_grad_batch(self, policies, rewards, m0, S0, H, seed_fns=None):
        """B value-and-gradient rollouts of the same model in flight together (pilco_rollout_grad_batch; the restarts of
        optimize_policy, pilco.py:94-107): policies: B LinearController specs; m0 (B, E), S0 (B, E, E) -> reward (B,),
        dW (B, U, E), db (B, U).  Each lane is bit-identical to its solo rollout_grad."""
        B = len(policies)
        E, U = policies[0]["state_dim"], policies[0]["control_dim"]
        arr, keep = self._policy_array(policies)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (B, E)); S0 = _f64(S0, (B, E, E))
        rew = np.zeros(B); dW = np.empty((B, U, E)); db = np.empty((B, U))
        if seed_fns is None:
            self._chk(self.lib.pilco_rollout_grad_batch(self.h, B, arr, r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(rew), _ptr(dW), _ptr(db)))
            return rew, dW, db
        errors = []
        cb = self._lane_seed_callback(seed_fns, errors)    # (seed_fns[i](traj) -> seeds of lane i: pilco_rollout_grad_batch_seeded)
        users = (C.c_void_p * B)(*[i + 1 for i in range(B)])
        rc = self.lib.pilco_rollout_grad_batch_seeded(self.h, B, arr, r, len(rewards), _ptr(m0), _ptr(S0), int(H), cb, users,
                                                      _ptr(rew), _ptr(dW), _ptr(db))
        if errors:
            raise errors[0]
        self._chk(rc)
        return rew, dW, db

    def rollout_grad_rbf_batch(self, policies, rewards, m0, S0, H, Xp, Yp, lsp, noisep, seed_fns=None):
        """The same for B RbfControllers: Xp (B, bf, E), Yp (B, bf, U), lsp (B, U, E), noisep (B, U) -> reward (B,), dX, dY, dls.
        The call uploads lane i's policy GP into the policy slot of lane i's context -- this context's slot holds lane 0's
        controller afterwards (whoever believed to own the slot must push its parameters again)."""
        B = len(policies)
        E, U = policies[0]["state_dim"], policies[0]["control_dim"]
        arr, keep = self._policy_array(policies)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (B, E)); S0 = _f64(S0, (B, E, E))
        Xp = _f64(Xp); bf = Xp.shape[1]
        Xp = _f64(Xp, (B, bf, E)); Yp = _f64(Yp, (B, bf, U)); lsp = _f64(lsp, (B, U, E)); noisep = _f64(noisep, (B, U))
        rew = np.zeros(B); dX = np.empty((B, bf, E)); dY = np.empty((B, bf, U)); dls = np.empty((B, U, E))
        self._slot_owner[SLOT_POLICY] = None   # (the call overwrites the slot)
        if seed_fns is None:
            self._chk(self.lib.pilco_rollout_grad_rbf_batch(self.h, B, arr, r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(Xp), _ptr(Yp),
                                                            _ptr(lsp), _ptr(noisep), bf, _ptr(rew), _ptr(dX), _ptr(dY), _ptr(dls)))
            return rew, dX, dY, dls
        errors = []
        cb = self._lane_seed_callback(seed_fns, errors)
        users = (C.c_void_p * B)(*[i + 1 for i in range(B)])
        rc = self.lib.pilco_rollout_grad_rbf_batch_seeded(self.h, B, arr, r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(Xp), _ptr(Yp),
                                                          _ptr(lsp), _ptr(noisep), bf, cb, users, _ptr(rew), _ptr(dX), _ptr(dY), _ptr(dls))
        if errors:
            raise errors[0]
        self._chk(rc)
        return rew, dX, dY, dls

    def gp_predict_vjp(self, slot, m, s, Mbar, Sbar, Vbar, D, E):
        m = _f64(m, (D,)); s = _f64(s, (D, D))
        Mb = _f64(Mbar, (E,)); Sb = _f64(Sbar, (E, E)); Vb = _f64(Vbar, (D, E))
        mbar = np.empty((1, D)); sbar = np.empty((D, D))
        self._chk(self.lib.pilco_gp_predict_vjp(self.h, slot, _ptr(m), _ptr(s), _ptr(Mb), _ptr(Sb), _ptr(Vb), _ptr(mbar), _ptr(sbar)))
        return mbar, sbar

    def rollout_tape(self, policy, rewards, m0, S0, H):
        E = policy["state_dim"]; D = E + policy["control_dim"]
        p, k1 = self._policy(policy)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (E,)); S0 = _f64(S0, (E, E))
        mH = np.empty((1, E)); SH = np.empty((E, E)); rew = np.zeros((1, 1))
        traj = np.empty((H + 1, E + E * E))
        TS = D + D * D + E * D + E + E * E + D * E
        tape = np.zeros((max(H, 1), TS))
        self._chk(self.lib.pilco_rollout_tape(self.h, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H),
                                              _ptr(mH), _ptr(SH), _ptr(rew), _ptr(traj), _ptr(tape)))
        return mH, SH, rew, traj, tape[:H]

    @staticmethod
    def _seed_callback(seed_fn, errors):
        """Wrap seed_fn(traj (H+1, E+E*E)) -> seeds of the same shape as a pilco_seed_fn; an exception inside the callback
        is kept in `errors` (it must not unwind through the C frames) and poisons the seeds so that the call fails."""
        def cb(_user, H, E, traj_p, seeds_p):
            n = (H + 1) * (E + E * E)
            out = np.ctypeslib.as_array(seeds_p, shape=(n,))
            try:
                traj = np.ctypeslib.as_array(traj_p, shape=(n,)).reshape(H + 1, E + E * E).copy()
                out[:] = np.asarray(seed_fn(traj), dtype=np.float64).reshape(n)
            except BaseException as exc:   # noqa: BLE001 -- re-raised by the caller
                errors.append(exc)
                out[:] = np.nan
        return SEED_FN(cb)

    def rollout_grad(self, policy, rewards, m0, S0, H, seed_fn=None):
        """(reward, dW (U,E), db (U)) for a squashed linear policy: the native reverse sweep (pilco_rollout_grad).
        seed_fn(traj) -> d objective / d (m_t, s_t) for an objective beyond the additive reward (pilco_rollout_grad_seeded)."""
        E = policy["state_dim"]; U = policy["control_dim"]
        p, k1 = self._policy(policy)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (E,)); S0 = _f64(S0, (E, E))
        rew = np.zeros((1, 1)); dW = np.empty((U, E)); db = np.empty((U,))
        errors = []
        cb = self._seed_callback(seed_fn, errors) if seed_fn is not None else SEED_FN()
        rc = self.lib.pilco_rollout_grad_seeded(self.h, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H), cb, None,
                                                _ptr(rew), _ptr(dW), _ptr(db))
        if errors:
            raise errors[0]
        self._chk(rc)
        return float(rew[0, 0]), dW, db

    def rollout_grad_rbf(self, policy, rewards, m0, S0, H, Xp, Yp, lsp, noisep, seed_fn=None):
        """(reward, dX (bf,E), dY (bf,U), dls (U,E)) for an RBF policy: the native reverse sweep (pilco_rollout_grad_rbf)."""
        E = policy["state_dim"]; U = policy["control_dim"]
        p, k1 = self._policy(policy)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (E,)); S0 = _f64(S0, (E, E))
        Xp = _f64(Xp); bf = Xp.shape[0]
        Xp = _f64(Xp, (bf, E)); Yp = _f64(Yp, (bf, U)); lsp = _f64(lsp, (U, E)); noisep = _f64(noisep, (U,))
        rew = np.zeros((1, 1)); dX = np.empty((bf, E)); dY = np.empty((bf, U)); dls = np.empty((U, E))
        errors = []
        cb = self._seed_callback(seed_fn, errors) if seed_fn is not None else SEED_FN()
        rc = self.lib.pilco_rollout_grad_rbf_seeded(self.h, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H),
                                                    _ptr(Xp), _ptr(Yp), _ptr(lsp), _ptr(noisep), bf, cb, None,
                                                    _ptr(rew), _ptr(dX), _ptr(dY), _ptr(dls))
        if errors:
            raise errors[0]
        self._chk(rc)
        return float(rew[0, 0]), dX, dY, dls

    def propagate(self, policy, m_x, s_x):
        E = policy["state_dim"]
        p, k1 = self._policy(policy)
        m_x = _f64(m_x, (E,))
        s_x = _f64(s_x, (E, E))
        M = np.empty((1, E))
        S = np.empty((E, E))
        self._chk(self.lib.pilco_propagate(self.h, C.byref(p), _ptr(m_x), _ptr(s_x), _ptr(M), _ptr(S)))
        return M, S

    def policy_action(self, policy, m, s):
        E, U = policy["state_dim"], policy["control_dim"]
        p, k1 = self._policy(policy)
        m = _f64(m, (E,))
        s = _f64(s, (E, E))
        M = np.empty((1, U))
        S = np.empty((U, U))
        V = np.empty((E, U))
        self._chk(self.lib.pilco_policy_action(self.h, C.byref(p), _ptr(m), _ptr(s), _ptr(M), _ptr(S), _ptr(V)))
        return M, S, V

    def reward_eval(self, rewards, E, m, s):
        r, k2 = self._rewards(rewards, E)
        m = _f64(m, (E,))
        s = _f64(s, (E, E))
        mu = C.c_double()
        var = C.c_double()
        self._chk(self.lib.pilco_reward_eval(self.h, r, len(rewards), E, _ptr(m), _ptr(s), C.byref(mu), C.byref(var)))
        return np.array([[mu.value]]), np.array([[var.value]])

    def rollout_timed(self, policy, rewards, m0, S0, H, reps, time_pair=True):
        E = policy["state_dim"]
        p, k1 = self._policy(policy)
        r, k2 = self._rewards(rewards, E)
        m0 = _f64(m0, (E,))
        S0 = _f64(S0, (E, E))
        mH = np.empty((1, E))
        SH = np.empty((E, E))
        rew = np.zeros((1, 1))
        ms_total = C.c_float()
        ms_pair = C.c_float()
        npair = C.c_int()
        self._chk(self.lib.pilco_rollout_timed(self.h, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H),
                                               int(reps), _ptr(mH), _ptr(SH), _ptr(rew), C.byref(ms_total),
                                               C.byref(ms_pair) if time_pair else None, C.byref(npair)))
        return dict(mH=mH, SH=SH, reward=rew, ms_total=ms_total.value, ms_pair=ms_pair.value,
                    n_pair_launches=npair.value)

    def factorize_timed(self, slot, reps):
        ms = C.c_float()
        self._chk(self.lib.pilco_factorize_timed(self.h, slot, int(reps), C.byref(ms)))
        return ms.value

    def set_pair_timing(self, on=True):
        self._chk(self.lib.pilco_set_pair_timing(self.h, 1 if on else 0))

    def get_pair_timing(self):
        """(summed ms, number of launches) of the O(N^2) kernel in the last rollout made while pair timing was on."""
        ms = C.c_float()
        n = C.c_int()
        self._chk(self.lib.pilco_get_pair_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_buffer(self, slot, which, n):
        """Developer aid: raw copy of a device work buffer (include/pilco_hip.h: pilco_debug_buffer)."""
        out = np.empty(int(n))
        self._chk(self.lib.pilco_debug_buffer(self.h, int(slot), int(which), _ptr(out), int(n)))
        return out

    def debug_timestamps(self, read=True):
        buf = (C.c_ulonglong * 64)()
        self._chk(self.lib.pilco_debug_timestamps(self.h, buf if read else None))
        return list(buf)

    def debug_blocks(self, n):
        buf = (C.c_ulonglong * n)()
        self._chk(self.lib.pilco_debug_blocks(self.h, buf, n))
        return list(buf)

    # ---- sharding
    def comm_unique_id(self):
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = self.lib.pilco_comm_unique_id(buf)
        if rc != PILCO_OK:
            raise PilcoError(rc, "ncclGetUniqueId failed")
        return bytes(buf.raw)

    def comm_init(self, id_bytes, rank, nranks):
        buf = C.create_string_buffer(bytes(id_bytes), COMM_ID_BYTES)
        self._chk(self.lib.pilco_comm_init(self.h, buf, int(rank), int(nranks)))
        self.nranks, self.has_comm = int(nranks), True

    def shard_pack(self, slot, m, s, D, E, nranks, rank):
        plan = shard_plan(E, D, nranks, rank)
        m = _f64(m, (D,))
        s = _f64(s, (D, D))
        seg = np.zeros(plan["SEG"])
        self._chk(self.lib.pilco_gp_shard_pack(self.h, slot, _ptr(m), _ptr(s), _ptr(seg)))
        return seg

    def shard_finish(self, slot, gathered, D, E):
        g = _f64(gathered).reshape(-1)
        M = np.empty((1, E))
        S = np.empty((E, E))
        V = np.empty((D, E))
        self._chk(self.lib.pilco_gp_shard_finish(self.h, slot, _ptr(g), _ptr(M), _ptr(S), _ptr(V)))
        return M, S, V

    def shard_set(self, rank, nranks):
        self._chk(self.lib.pilco_shard_set(self.h, int(rank), int(nranks)))
        self.nranks = int(nranks)

    # beta rows of a sharded factorisation over any host transport (ranks in different processes, no communicator)
    def beta_export(self, slot=0):
        el, npad = C.c_int(0), C.c_int(0)
        self._chk(self.lib.pilco_gp_beta_rows(self.h, slot, C.byref(el), C.byref(npad)))
        rows = np.zeros((el.value, npad.value))
        self._chk(self.lib.pilco_gp_beta_export(self.h, slot, _ptr(rows)))
        return rows

    def beta_import(self, all_rows, slot=0):
        """all_rows: the exported blocks of all ranks in rank order, (nranks, elcap, npad)."""
        all_rows = np.ascontiguousarray(all_rows, dtype=np.float64)
        self._chk(self.lib.pilco_gp_beta_import(self.h, slot, _ptr(all_rows)))

    # peer exchange (include/pilco_hip.h): the per-step all-gather as direct stores into the other ranks' memory
    def peer_export(self):
        """-> the 64 opaque bytes of this rank's exchange area (all-gather them over any host transport)."""
        buf = C.create_string_buffer(PEER_HANDLE_BYTES)
        self._chk(self.lib.pilco_peer_export(self.h, buf))
        return bytes(buf.raw)

    def peer_attach(self, handles, share_gpu=False):
        """handles: the nranks exported handles in rank order.  share_gpu: ranks share a GPU (oversubscribed tests)."""
        blob = b"".join(bytes(h) for h in handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._chk(self.lib.pilco_peer_attach(self.h, buf, 1 if share_gpu else 0))

    def comm_count(self):
        """Ranks RCCL itself reports for the attached communicator (0: none)."""
        return int(self.lib.pilco_comm_count(self.h))

    def peer_detach(self):
        self._chk(self.lib.pilco_peer_detach(self.h))

    def peer_attached(self):
        return bool(self.lib.pilco_peer_attached(self.h))


def rollout_grad_group(ctxs, policy, rewards, m0, S0, H):
    """Value and gradient of one sharded rollout over the contexts of this process (pilco_rollout_grad_group):
    (reward (n,), dW (n, U, E), db (n, U)) -- every rank's results."""
    c0 = ctxs[0]
    n = len(ctxs)
    E, U = policy["state_dim"], policy["control_dim"]
    p, k1 = c0._policy(policy)
    r, k2 = c0._rewards(rewards, E)
    m0 = _f64(m0, (E,)); S0 = _f64(S0, (E, E))
    arr = (_vp * n)(*[c.h for c in ctxs])
    rew, dW, db = np.zeros(n), np.empty((n, U, E)), np.empty((n, U))
    c0._chk(c0.lib.pilco_rollout_grad_group(arr, n, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(rew), _ptr(dW), _ptr(db)))
    return rew, dW, db


def rollout_grad_rbf_group(ctxs, policy, rewards, m0, S0, H, Xp, Yp, lsp, noisep):
    """Value and gradient of one sharded rollout with an RbfController over the contexts of this process
    (pilco_rollout_grad_rbf_group): (reward (n,), dX (n, bf, E), dY (n, bf, U), dls (n, U, E)) -- every rank's results."""
    c0 = ctxs[0]
    n = len(ctxs)
    E, U = policy["state_dim"], policy["control_dim"]
    p, k1 = c0._policy(policy)
    r, k2 = c0._rewards(rewards, E)
    m0 = _f64(m0, (E,)); S0 = _f64(S0, (E, E))
    Xp = _f64(Xp); bf = Xp.shape[0]
    Xp = _f64(Xp, (bf, E)); Yp = _f64(Yp, (bf, U)); lsp = _f64(lsp, (U, E)); noisep = _f64(noisep, (U,))
    arr = (_vp * n)(*[c.h for c in ctxs])
    rew, dX, dY, dls = np.zeros(n), np.empty((n, bf, E)), np.empty((n, bf, U)), np.empty((n, U, E))
    c0._chk(c0.lib.pilco_rollout_grad_rbf_group(arr, n, C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(Xp), _ptr(Yp), _ptr(lsp),
                                                _ptr(noisep), bf, _ptr(rew), _ptr(dX), _ptr(dY), _ptr(dls)))
    return rew, dX, dY, dls


def group_nlml(ctxs, slot, D, E):
    """pilco_gp_nlml over the contexts of one process that shard a model by output: every rank evaluates the outputs it owns
    (the others come back NaN), this combines them -- what the ncclAllGather inside pilco_gp_nlml does between processes."""
    nlml, grad = np.full(E, np.nan), np.full((E, D + 2), np.nan)
    for c in ctxs:
        n, g = c.gp_nlml(slot, D, E)
        own = ~np.isnan(n)
        assert not np.any(own & ~np.isnan(nlml)), "two ranks claim the same output"
        nlml[own], grad[own] = n[own], g[own]
    return nlml, grad


def group_fitc_nlml(ctxs, slot, Z_all, D, E):
    """pilco_gp_fitc_nlml over the contexts of one process that shard a sparse model by output (see group_nlml)."""
    Z_all = np.asarray(Z_all, np.float64)
    nlml, gh, gz = np.full(E, np.nan), np.full((E, D + 2), np.nan), np.full(Z_all.shape, np.nan)
    for c in ctxs:
        n, h, z = c.gp_fitc_nlml(slot, Z_all, D, E)
        own = ~np.isnan(n)
        assert not np.any(own & ~np.isnan(nlml)), "two ranks claim the same output"
        nlml[own], gh[own], gz[own] = n[own], h[own], z[own]
    return nlml, gh, gz


def group_sync_model(ctxs, slot=0):
    """Exchange the beta rows of the contexts of this process after each has factorised its own outputs."""
    arr = (_vp * len(ctxs))(*[c.h for c in ctxs])
    ctxs[0]._chk(ctxs[0].lib.pilco_group_sync_model(arr, len(ctxs), int(slot)))


def group_peer_attach(ctxs):
    """Attach the peer exchange between the contexts of this process (context i = rank i)."""
    arr = (_vp * len(ctxs))(*[c.h for c in ctxs])
    ctxs[0]._chk(ctxs[0].lib.pilco_group_peer_attach(arr, len(ctxs)))


def rollout_group(ctxs, policy, rewards, m0, S0, H, want_traj=False):
    """One rollout sharded over the contexts of this process (context i = rank i; each must be shard_set(i, n) and hold
    the same factorised model).  -> (mH, SH, reward[, traj], mismatch)."""
    c0 = ctxs[0]
    E = policy["state_dim"]
    p, k1 = c0._policy(policy)
    r, k2 = c0._rewards(rewards, E)
    m0 = _f64(m0, (E,))
    S0 = _f64(S0, (E, E))
    mH, SH, rew = np.empty((1, E)), np.empty((E, E)), np.zeros((1, 1))
    traj = np.empty((H + 1, E + E * E)) if want_traj else None
    arr = (_vp * len(ctxs))(*[c.h for c in ctxs])
    mm = C.c_int(0)
    c0._chk(c0.lib.pilco_rollout_group(arr, len(ctxs), C.byref(p), r, len(rewards), _ptr(m0), _ptr(S0), int(H), _ptr(mH), _ptr(SH),
                                       _ptr(rew), _ptr(traj), C.byref(mm)))
    return (mH, SH, rew, traj, mm.value) if want_traj else (mH, SH, rew, mm.value)


def shard_plan(E, D, nranks, rank):
    """Ownership / gather-buffer layout of the sharded step (pure host function)."""
    lib = load_library()
    out = (C.c_int * 5)()
    rc = lib.pilco_shard_plan(int(E), int(D), int(nranks), int(rank), out)
    if rc != PILCO_OK:
        raise PilcoError(rc, "bad shard plan arguments")
    return dict(PL=out[0], EL=out[1], SEG=out[2], OUTOFF=out[3], P=out[4])


def shard_pair_slot(E, D, nranks, a, b):
    return load_library().pilco_shard_pair_slot(int(E), int(D), int(nranks), int(a), int(b))


def shard_output_slot(E, D, nranks, a):
    return load_library().pilco_shard_output_slot(int(E), int(D), int(nranks), int(a))


_default_ctx = None


def get_context():
    """Process-wide default context (device = $PILCO_DEVICE or $LOCAL_RANK or 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_context(ctx):
    global _default_ctx
    _default_ctx = ctx
    _ctx_pool[:] = []


# A context holds ONE dynamics model and ONE policy GP on the device.  Two PILCO objects alive at the same time therefore
# get a context each (round 2 put both on the default context, where every switch from one to the other re-uploaded and
# re-factorised the model: a silent 1.7 ms at N = 1000).  Contexts whose owner has gone are handed out again.
# PILCO_CTX_POOL=0: everything on the default context, as before.
_ctx_pool = []   # [[context, weakref to its owner or None]]; entry 0 is the default context


def resolve_ctx(obj):
    """The context of a model-layer object that has none yet: its PILCO object's (which asks context_for on ITS first
    use), or the default context for an object that stands alone.  Nothing touches the device before this is called."""
    ref = getattr(obj, "_ctx_owner", None)
    owner = ref() if ref is not None else None
    return owner.ctx if owner is not None else get_context()


def context_adopted(ctx, owner):
    """A PILCO object took `ctx` from one of its components: if it is a pooled context without a live holder, the object
    becomes its holder (otherwise context_for would hand the same context to the next object as well)."""
    import weakref
    for ent in _ctx_pool:
        if ent[0] is ctx and (ent[1] is None or ent[1]() is None):
            ent[1] = weakref.ref(owner)


def context_for(owner):
    import weakref
    d = get_context()
    if os.environ.get("PILCO_CTX_POOL", "1") == "0":
        return d
    # a sharded default context (several ranks, a communicator or a peer group) is THE context of this process: a pooled
    # sibling would run the full unsharded model on every rank and ignore what was configured on the default one
    if getattr(d, "nranks", 1) != 1 or getattr(d, "has_comm", False):
        return d
    if not _ctx_pool or _ctx_pool[0][0] is not d:
        _ctx_pool[:] = [[d, None]]
    for ent in _ctx_pool:
        holder = ent[1]() if ent[1] is not None else None
        if holder is None or holder is owner:
            ent[1] = weakref.ref(owner)
            return ent[0]
    try:
        c = type(d)(device=getattr(d, "device", None))
    except TypeError:   # a stand-in installed by set_context() (tests) that takes no device
        c = type(d)()
    for name, arg in getattr(d, "_settings", {}).items():   # what was configured on the default context applies to its siblings
        getattr(c, name)(*arg)
    _ctx_pool.append([c, weakref.ref(owner)])
    return c
