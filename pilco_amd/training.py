"""Optimiser glue around the device kernels.

``optimize_mgpr``   -- MGPR.optimize (pilco/models/mgpr.py:47-75): MAP fit of the per-output GP
hyper-parameters.  The objective follows GPflow's GPR.training_loss as recalled in SURVEY.md
Appendix C: negative log marginal likelihood minus the Gamma log-priors on the lengthscales
(shape 1.1, rate 0.1) and the kernel variance (shape 1.5, rate 0.5), optimised in softplus space
(noise variance >= 1e-6) with SciPy L-BFGS-B.  NLML and its gradient come from the device
(pilco_gp_nlml); the E outputs are independent problems, each solved by its own L-BFGS-B run as in the
reference, the runs evaluated in lockstep (one batched factorisation per round, lockstep_minimize).
The end points are pinned against the executed reference (tests/golden/models_optimisation.npz,
sparse_models_optimisation.npz).

``optimize_policy`` -- PILCO.optimize_policy (pilco/models/pilco.py:75-113): L-BFGS-B over the
controller parameters with the GP frozen, restarts via controller.randomize().  The gradient of the
rollout reward is the hand-derived adjoint (native reverse sweep, csrc/grad.hip) for the
linear and RBF controllers with exponential / linear / combined rewards; other cases (a PILCO subclass that
overrides predict such as SafePILCO, D beyond the reverse pass's limit) fall back to central finite differences
of device rollouts.  Both are deterministic and bitwise repeatable.
"""
from __future__ import annotations

import time

import numpy as np
from scipy.optimize import minimize
from scipy.special import gammaln

NOISE_LOWER = 1e-6
# L-BFGS-B iteration cap of the model fits.  The reference passes no options to SciPy for them (mgpr.py:52,55,66; the
# maxiter argument of PILCO.optimize_models is ignored, pilco.py:52-56), i.e. SciPy's default.
MODEL_FIT_MAXITER = 15000


def _softplus(u):
    return np.logaddexp(0.0, u)


def _softplus_inv(x):
    x = np.maximum(x, 1e-300)
    return np.where(x > 30.0, x, np.log(np.expm1(np.minimum(x, 30.0))))


def _dsoftplus(u):
    return 1.0 / (1.0 + np.exp(-u))


def _gamma_logpdf_and_grad(x, shape, rate):
    lp = shape * np.log(rate) - gammaln(shape) + (shape - 1.0) * np.log(x) - rate * x
    return lp, (shape - 1.0) / x - rate


def _mgpr_pack(mgpr):
    ls, var, nz = mgpr.lengthscales, mgpr.variance, mgpr.noise
    return np.concatenate([_softplus_inv(ls).ravel(), _softplus_inv(var), _softplus_inv(np.maximum(nz - NOISE_LOWER, 1e-12))])


def _mgpr_unpack(mgpr, u):
    E, D = mgpr.num_outputs, mgpr.num_dims
    ls = _softplus(u[:E * D]).reshape(E, D)
    var = _softplus(u[E * D:E * D + E])
    nz = NOISE_LOWER + _softplus(u[E * D + E:])
    return ls, var, nz


def _eval_isolating(eval_all, u, parts, last_good, safe, wall):
    """eval_all(u) with a wall exception confined to the output that caused it.  The reference fits one optimiser per
    output (mgpr.py:47-56): a Gram matrix that is not positive definite at output a's trial point is a failed evaluation of
    output a's line search and of nobody else's.  The batched device call reports ONE failing output per call (exception
    attribute `output`, from pilco_last_not_pd_output); that output is set back to the last point at which it evaluated
    (then to `safe`, if that one fails too) and the batch is evaluated again, until it goes through.
    Returns (values, gradient, walled): the outputs in `walled` get no value from this round."""
    u_eval = np.array(u, dtype=np.float64)
    walled, tried_safe = set(), set()
    while True:
        try:
            vals, grad = eval_all(u_eval)
            return vals, grad, walled
        except wall as exc:
            bad = getattr(exc, "output", None)
            if bad is None or not (0 <= int(bad) < len(parts)):
                raise                       # the failing output is unknown: the caller walls the whole round
            bad = int(bad)
            if bad not in walled:
                walled.add(bad)
                u_eval[parts[bad]] = last_good[bad]
            elif safe is not None and bad not in tried_safe:
                tried_safe.add(bad)         # its last evaluated point fails as well (a start that is not positive definite)
                u_eval[parts[bad]] = np.asarray(safe, dtype=np.float64)[parts[bad]]
            else:
                raise


def lockstep_minimize(eval_all, u0, parts, maxiter=15000, wall=(RuntimeError,), safe=None):
    """E independent L-BFGS-B problems (problem a owns the entries parts[a] of the packed vector u) solved exactly as E
    separate scipy.optimize.minimize runs would solve them -- what the reference does, one optimiser per output
    (mgpr.py:47-56) -- while every round of function evaluations costs ONE call of eval_all(u) -> (values (E,), gradient):
    each problem runs in its own thread, its objective posts the point it wants evaluated and waits; when every problem
    that is still running has posted, the coordinator evaluates them all together (one batched device call) and hands
    the values back.  A joint run on the sum of the losses is not the same thing: its line search and stopping rule couple
    the outputs, and it can end in another local optimum (found against the executed reference).
    An evaluation that raises one of `wall` (a Gram matrix that is not positive definite) is a wall (1e25, zero gradient)
    for the output that caused it ONLY (exception attribute `output`); the others get their true values from a second
    evaluation of the batch with that output set back (_eval_isolating; `safe`: a packed vector every output can be
    evaluated at, the last resort).  Without the attribute the whole round is walled.
    Returns (u_end, values_end); an output whose END point cannot be evaluated reports 1e25."""
    import threading
    E = len(parts)
    u = np.array(u0, dtype=np.float64)
    cv = threading.Condition()
    pending, results, final, errors = {}, {}, {}, []
    active = set(range(E))
    state = {"abort": None}
    last_good = [u[parts[a]].copy() for a in range(E)]

    class _Abort(Exception):
        pass

    def worker(a):
        def fun(ua):
            with cv:
                if state["abort"] is not None:
                    raise _Abort()
                pending[a] = np.array(ua, dtype=np.float64)
                cv.notify_all()
                while a not in results and state["abort"] is None:
                    cv.wait()
                if state["abort"] is not None:
                    raise _Abort()
                return results.pop(a)
        x = None
        try:
            x = minimize(fun, u[parts[a]].copy(), jac=True, method="L-BFGS-B", options=dict(maxiter=maxiter)).x
        except _Abort:
            pass
        except BaseException as exc:   # noqa: BLE001 -- re-raised by the coordinator
            errors.append(exc)
        with cv:
            if x is not None:
                final[a] = x
            active.discard(a)
            pending.pop(a, None)
            cv.notify_all()

    def evaluate(req):
        """{a: (value, gradient)} for the outputs of `req` at the current u."""
        try:
            vals, grad, walled = _eval_isolating(eval_all, u, parts, last_good, safe, wall)
        except wall:   # cannot be pinned on one output: a wall for every problem of this round
            return {a: (1e25, np.zeros(len(parts[a]))) for a in req}
        for a in range(E):
            if a not in walled:
                last_good[a] = u[parts[a]].copy()
        return {a: ((1e25, np.zeros(len(parts[a]))) if a in walled else (float(vals[a]), np.array(grad[parts[a]], dtype=np.float64)))
                for a in req}

    threads = [threading.Thread(target=worker, args=(a,), daemon=True) for a in range(E)]
    for t in threads:
        t.start()
    while True:
        with cv:
            while active and not all(a in pending for a in active) and not errors:
                cv.wait()
            if errors and state["abort"] is None:
                state["abort"] = errors[0]          # one run failed: the others stop at their next evaluation
                cv.notify_all()
            for a, x in final.items():
                u[parts[a]] = x
            if not active:
                break
            if state["abort"] is not None:
                cv.wait(0.05)
                continue
            req = [a for a in pending if a in active]
            for a in req:
                u[parts[a]] = pending.pop(a)
        try:
            out = evaluate(req)
        except BaseException as exc:   # noqa: BLE001 -- a device error: release every waiting run, then re-raise here
            with cv:
                state["abort"] = exc
                cv.notify_all()
            continue
        with cv:
            results.update(out)
            cv.notify_all()
    for t in threads:
        t.join()
    if state["abort"] is not None:
        raise state["abort"]
    end = evaluate(list(range(E)))     # (an end point that is not positive definite -- only a start that was not -- reports the wall value)
    return u, np.array([end[a][0] for a in range(E)], dtype=np.float64)


def _trainable_masks(models):
    """Per output: is the lengthscale vector / kernel variance / likelihood variance in the trainable set?  The reference
    hands model.trainable_variables to its optimiser model by model (mgpr.py:51-56,66), so every output has its own set."""
    return (np.array([bool(m.kernel.lengthscales.trainable) for m in models]),
            np.array([bool(m.kernel.variance.trainable) for m in models]),
            np.array([bool(m.likelihood.variance.trainable) for m in models]))


def _as_mask(flag, E):
    return np.full(E, bool(flag)) if np.ndim(flag) == 0 else np.asarray(flag, bool)


def _require_complete(values, what, ctx=None):
    """A context sharded by output evaluates only the outputs it owns; without a communicator the others come back NaN and
    the ranks have to be combined by the caller (pilco_amd._lib.group_nlml / group_fitc_nlml).  Feeding NaN losses and
    gradients to L-BFGS-B instead would fail silently, so the optimiser entry points refuse.  (Only there: a non-finite
    objective on an unsharded context -- overflowing hyper-parameters -- is the optimiser's business, which backs off.)"""
    sharded_alone = ctx is None or (getattr(ctx, "nranks", 1) > 1 and not getattr(ctx, "has_comm", False))
    if sharded_alone and not np.all(np.isfinite(values)):
        raise RuntimeError("%s.optimize: the training objective came back incomplete (NaN for outputs another rank owns): this "
                           "context is sharded by output and has no communicator -- attach one (Context.comm_init) or train on an "
                           "unsharded context" % what)


def mgpr_objective(mgpr, u, noise_trainable=True, ls_trainable=True, var_trainable=True):
    """Per-output GPflow training loss and its gradient in the unconstrained space.  The *_trainable arguments are
    per-output masks (or one bool for all): a parameter outside an output's trainable set keeps its value (zero gradient
    component: L-BFGS-B never moves it, and its inner products and stopping test do not see it)."""
    E, D = mgpr.num_outputs, mgpr.num_dims
    tn, tl, tv = _as_mask(noise_trainable, E), _as_mask(ls_trainable, E), _as_mask(var_trainable, E)
    ls, var, nz = _mgpr_unpack(mgpr, u)
    for i, m in enumerate(mgpr.models):
        m.kernel.lengthscales.assign(ls[i])
        m.kernel.variance.assign(var[i])
        if tn[i]:                                # a fixed likelihood variance is not touched (no transform round trip)
            m.likelihood.variance.assign(nz[i])
    mgpr._sync()
    nlml, g = mgpr.ctx.gp_nlml(mgpr._slot, D, E)
    _require_complete(nlml, "MGPR", mgpr.ctx)
    lp_l, dlp_l = _gamma_logpdf_and_grad(ls, 1.1, 0.1)          # mgpr.py:33
    lp_v, dlp_v = _gamma_logpdf_and_grad(var, 1.5, 0.5)         # mgpr.py:34
    # GPflow's training_loss adds log_prior_density of the TRAINABLE parameters only: a frozen parameter's prior is not in
    # the loss (a constant there, but it would shift scipy's relative ftol test and the reported value)
    per_output = nlml - (lp_l * tl[:, None]).sum(1) - lp_v * tv
    g_ls = (g[:, :D] - dlp_l) * _dsoftplus(u[:E * D]).reshape(E, D) * tl[:, None]
    g_var = (g[:, D] - dlp_v) * _dsoftplus(u[E * D:E * D + E]) * tv
    g_nz = g[:, D + 1] * _dsoftplus(u[E * D + E:]) * tn
    return per_output, np.concatenate([g_ls.ravel(), g_var, g_nz])


def _restart_draws(E, D, restarts, noise_trainable):
    """randomize() (mgpr.py:8-15) for every (model, restart) in the order the reference draws them from NumPy's global
    generator: model by model (mgpr.py:58-66), within a model restart by restart; per draw lengthscales (D) and kernel
    variance (always, trainable or not), and the likelihood variance only if THAT model's is trainable.  The lockstep driver
    needs restart r of ALL outputs at once, so the draws are taken up front; with the same np.random.seed the starts are the
    reference's."""
    tn = _as_mask(noise_trainable, E)
    ls, var, nz = np.empty((restarts, E, D)), np.empty((restarts, E)), np.full((restarts, E), np.nan)
    for a in range(E):
        for r in range(restarts):
            ls[r, a] = 1 + 0.01 * np.random.normal(size=(D,))
            var[r, a] = 1 + 0.01 * np.random.normal(size=())
            if tn[a]:
                nz[r, a] = 1 + 0.01 * np.random.normal()
    return ls, var, nz


def _check_keep(keep):
    if keep not in ("best", "last"):
        raise ValueError("keep: 'last' (what the reference ends with) or 'best' (per output the fit with the lowest loss)")


def optimize_mgpr(mgpr, restarts=1, maxiter=None, verbose=False, keep="last"):
    """MGPR.optimize (mgpr.py:47-75).  keep='last' (default) is what the reference ends with: its `best_params` hold the
    live Parameter objects, not copies (mgpr.py:59-62,69-71), so the final assign (mgpr.py:73-75) assigns every parameter
    to itself and the model keeps the LAST restart's fit whether or not it was better -- with the same np.random.seed the
    product's models end where the reference's end.  keep='best': every output ends with the better of its fits, what that
    bookkeeping sets out to do (an extension; not the reference's behaviour)."""
    from .models.smgpr import SMGPR
    from . import _lib
    _check_keep(keep)
    maxiter = MODEL_FIT_MAXITER if maxiter is None else maxiter
    if isinstance(mgpr, SMGPR):
        raise TypeError("optimize_mgpr fits the exact GP objective; SMGPR.optimize uses optimize_smgpr (GPRFITC objective)")
    tl, tv, tn = _trainable_masks(mgpr.models)
    E, D = mgpr.num_outputs, mgpr.num_dims
    parts = [np.concatenate([np.arange(a * D, (a + 1) * D), [E * D + a], [E * D + E + a]]) for a in range(E)]
    # a point every output can be evaluated at (unit lengthscales / variance / noise), the isolation's last resort
    safe = np.concatenate([_softplus_inv(np.ones(E * D)), _softplus_inv(np.ones(E)), _softplus_inv(np.ones(E) - NOISE_LOWER)])

    def objective(u):
        return mgpr_objective(mgpr, u, tn, tl, tv)

    def run(u0):
        return lockstep_minimize(objective, u0, parts, maxiter, wall=(_lib.NotPositiveDefiniteError,), safe=safe)

    u_best, per_best = run(_mgpr_pack(mgpr))
    ls_r, var_r, nz_r = _restart_draws(E, D, restarts, tn)
    for r in range(restarts):
        nz0 = np.where(tn, nz_r[r], mgpr.noise)
        u0 = np.concatenate([_softplus_inv(ls_r[r]).ravel(), _softplus_inv(var_r[r]), _softplus_inv(np.maximum(nz0 - NOISE_LOWER, 1e-12))])
        u, per = run(u0)
        better = per < per_best if keep == "best" else np.ones(E, bool)
        if verbose:
            print("restart: per-output losses", per, "kept", better)
        ub = u_best.copy()
        for a in np.nonzero(better)[0]:
            ub[parts[a]] = u[parts[a]]
        u_best, per_best = ub, np.where(better, per, per_best)
    try:
        objective(u_best)   # leaves the kept parameters assigned
    except _lib.NotPositiveDefiniteError:
        pass                # (assigned all the same; only a start that was not positive definite ends here, and the
                            # reference's model would hold such parameters too)
    mgpr._sync()
    return per_best


def smgpr_objective(smgpr, u, noise_trainable=True, ls_trainable=True, var_trainable=True):
    """Per-output gpflow GPRFITC training loss (no priors: smgpr.py:16-22 sets none) and its gradient in the
    unconstrained space: softplus for lengthscales / variances (noise floor 1e-6), identity for the inducing inputs.
    *_trainable: per-output masks as in mgpr_objective."""
    E, D, M = smgpr.num_outputs, smgpr.num_dims, smgpr.num_induced_points
    tn, tl, tv = _as_mask(noise_trainable, E), _as_mask(ls_trainable, E), _as_mask(var_trainable, E)
    nk = E * D + 2 * E
    ls, var, nz = _mgpr_unpack(smgpr, u[:nk])
    Z = u[nk:].reshape(E, M, D)
    for i, m in enumerate(smgpr.models):
        m.kernel.lengthscales.assign(ls[i])
        m.kernel.variance.assign(var[i])
        if tn[i]:
            m.likelihood.variance.assign(nz[i])
    smgpr._sync()
    nlml, gh, gz = smgpr.ctx.gp_fitc_nlml(smgpr._slot, Z, D, E)
    _require_complete(nlml, "SMGPR", smgpr.ctx)
    g_ls = gh[:, :D] * _dsoftplus(u[:E * D]).reshape(E, D) * tl[:, None]
    g_var = gh[:, D] * _dsoftplus(u[E * D:E * D + E]) * tv
    g_nz = gh[:, D + 1] * _dsoftplus(u[E * D + E:nk]) * tn
    return nlml, np.concatenate([g_ls.ravel(), g_var, g_nz, gz.ravel()])


def optimize_smgpr(smgpr, restarts=1, maxiter=None, keep="last"):
    """MGPR.optimize applied to GPRFITC models (mgpr.py:47-75 with smgpr.py:16-22): every output's kernel
    hyper-parameters, noise variance and OWN inducing inputs by L-BFGS-B on the device objective (pilco_gp_fitc_nlml);
    the outputs are independent problems: one L-BFGS-B run each as in the reference, evaluated in lockstep
    (lockstep_minimize).  `restarts` extra fits start from randomize() (mgpr.py:8-15), which leaves the inducing inputs
    alone: every restart starts from the inducing inputs the previous fit of that output ended with.  keep: as in
    optimize_mgpr."""
    from . import _lib
    _check_keep(keep)
    maxiter = MODEL_FIT_MAXITER if maxiter is None else maxiter
    E, D, M = smgpr.num_outputs, smgpr.num_dims, smgpr.num_induced_points
    Z0 = np.stack([np.asarray(m.inducing_variable.Z.numpy(), np.float64) for m in smgpr.models])
    nk = E * D + 2 * E
    parts = [np.concatenate([np.arange(a * D, (a + 1) * D), [E * D + a], [E * D + E + a],
                             nk + np.arange(a * M * D, (a + 1) * M * D)]) for a in range(E)]
    tl, tv, tn = _trainable_masks(smgpr.models)
    safe = np.concatenate([_softplus_inv(np.ones(E * D)), _softplus_inv(np.ones(E)), _softplus_inv(np.ones(E) - NOISE_LOWER), Z0.ravel()])

    def objective(u):
        return smgpr_objective(smgpr, u, tn, tl, tv)

    def run(u0):
        return lockstep_minimize(objective, u0, parts, maxiter, wall=(_lib.NotPositiveDefiniteError,), safe=safe)

    u_best, per_best = run(np.concatenate([_mgpr_pack(smgpr), Z0.ravel()]))
    u_prev = u_best
    ls_r, var_r, nz_r = _restart_draws(E, D, restarts, tn)
    for r in range(restarts):
        nz0 = np.where(tn, nz_r[r], smgpr.noise)
        u0 = np.concatenate([_softplus_inv(ls_r[r]).ravel(), _softplus_inv(var_r[r]), _softplus_inv(np.maximum(nz0 - NOISE_LOWER, 1e-12)),
                             u_prev[nk:]])
        u_prev, per = run(u0)
        better = per < per_best if keep == "best" else np.ones(E, bool)
        ub = u_best.copy()
        for a in np.nonzero(better)[0]:
            ub[parts[a]] = u_prev[parts[a]]
        u_best, per_best = ub, np.where(better, per, per_best)
    try:
        objective(u_best)      # leaves the kept kernel parameters assigned
    except _lib.NotPositiveDefiniteError:
        pass
    Zf = u_best[nk:].reshape(E, M, D)
    for i, m in enumerate(smgpr.models):
        m.inducing_variable.Z.assign(Zf[i])
    smgpr._sync()
    return per_best


# --------------------------------------------------------------------------- policy
def _policy_params(controller):
    """(get, set) over a flat unconstrained vector for LinearController / RbfController."""
    from .controllers import LinearController, RbfController
    if isinstance(controller, LinearController):
        shapes = [controller.W.shape, controller.b.shape]

        def get():
            return np.concatenate([controller.W.numpy().ravel(), controller.b.numpy().ravel()])

        def put(u):
            n0 = int(np.prod(shapes[0]))
            controller.W.assign(u[:n0].reshape(shapes[0]))
            controller.b.assign(u[n0:].reshape(shapes[1]))
        return get, put
    if isinstance(controller, RbfController):
        gp = controller._gp
        bf, d, k = gp.num_datapoints, gp.num_dims, gp.num_outputs
        lower = 1e-3                                              # positive(lower=1e-3), controllers.py:100

        def get():
            return np.concatenate([gp.X.ravel(), gp.Y.ravel(), _softplus_inv(gp.lengthscales - lower).ravel()])

        def put(u):
            X = u[:bf * d].reshape(bf, d)
            Y = u[bf * d:bf * d + bf * k].reshape(bf, k)
            ls = lower + _softplus(u[bf * d + bf * k:]).reshape(k, d)
            gp.set_data((X, Y))
            for i, m in enumerate(gp.models):
                m.kernel.lengthscales.assign(ls[i])
        return get, put
    raise TypeError("optimize_policy supports LinearController and RbfController")


def policy_loss_and_grad(pilco, u, put, eps=1e-6):
    """-reward and its gradient: the hand-derived adjoint (PILCO.value_and_gradient) for linear and RBF controllers
    with exponential / linear / combined rewards, central differences of device rollouts (2n+1 of them) otherwise."""
    from . import _lib
    from .controllers import LinearController, RbfController
    put(u)
    ctl = pilco.controller
    from .models.pilco import PILCO
    # the adjoint differentiates PILCO.predict's reward (pilco.py:118-136); a subclass that overrides predict (SafePILCO's
    # multiplicative risk term, safe_pilco.py:29-50) is differentiated by finite differences of ITS training_loss instead
    plain = type(pilco).predict is PILCO.predict
    # ... unless it says what predict() adds to the additive reward as a function of the state trajectory
    # (trajectory_objective -> value, cotangent seeds): the native sweep takes the seeds (pilco_rollout_grad_seeded)
    # ... and so does a plain PILCO whose reward has terms the device does not evaluate (host reward terms, e.g. Safe-PILCO
    # constraints inside a CombinedRewards: examples/safe_swimmer_run.py:59-64)
    # (a subclass that overrides predict must bring its OWN trajectory_objective to take this path)
    own_objective = getattr(type(pilco), "trajectory_objective", None) is not PILCO.trajectory_objective
    seeded = (plain and bool(pilco._host_reward_terms())) or ((not plain) and own_objective and hasattr(pilco, "trajectory_objective"))
    analytic = ((plain or seeded) and pilco.control_dim > 0 and pilco.state_dim + pilco.control_dim <= 32   # the forward path's limit; the Jacobian tape serves D <= 14, the per-step device adjoint the rest
                and all(t["kind"] in (_lib.REWARD_EXPONENTIAL, _lib.REWARD_LINEAR) for t in pilco._reward_terms()))
    extra = {"v": 0.0, "ok": True}

    def seed_fn(traj):
        out = pilco.trajectory_objective(traj)
        if out is None:
            extra["ok"] = False
            return np.zeros_like(traj)
        extra["v"] = float(out[0])
        return out[1]

    if analytic and isinstance(ctl, (LinearController, RbfController)):
        r, grads = pilco.value_and_gradient(seed_fn if seeded else None)
        if extra["ok"]:
            r += extra["v"]
            if isinstance(ctl, LinearController):
                Wb, bb = grads
                return -r, -np.concatenate([Wb.ravel(), bb.ravel()])
            Xb, Yb, lb = grads
            nu = lb.size                                         # ls = lower + softplus(u): d ls / du = sigmoid(u)
            return -r, -np.concatenate([Xb.ravel(), Yb.ravel(), (lb * _dsoftplus(u[-nu:]).reshape(lb.shape)).ravel()])
    f0 = float(pilco.training_loss()[0, 0])
    g = np.empty_like(u)
    for i in range(u.size):
        h = eps * max(1.0, abs(u[i]))
        up = u.copy()
        up[i] += h
        put(up)
        fp = float(pilco.training_loss()[0, 0])
        up[i] -= 2 * h
        put(up)
        fm = float(pilco.training_loss()[0, 0])
        g[i] = (fp - fm) / (2 * h)
    put(u)
    return f0, g


MAX_LANES = 64   # PILCO_MAX_LANES of include/pilco_hip.h


def _restart_lanes_apply(pilco, restarts=2):
    """Can the restarts of optimize_policy run as lanes of ONE batched value-and-gradient call per round?  The plain additive
    reward through the native sweep on one rank (what policy_loss_and_grad calls `analytic` and not `seeded`)."""
    import os
    from . import _lib
    from .controllers import LinearController, RbfController
    from .models.pilco import PILCO
    if os.environ.get("PILCO_RESTART_LANES", "1") == "0":
        return False
    ctx = pilco.ctx
    if getattr(ctx, "nranks", 1) != 1 or getattr(ctx, "has_comm", False) or not hasattr(ctx, "rollout_grad_batch"):
        return False
    if restarts > MAX_LANES:   # (the batch entry points take at most 64 lanes: more restarts run one after the other, as before)
        return False
    if not isinstance(pilco.controller, (LinearController, RbfController)):
        return False
    if not (pilco.control_dim > 0 and pilco.state_dim + pilco.control_dim <= 32
            and all(t["kind"] in (_lib.REWARD_EXPONENTIAL, _lib.REWARD_LINEAR) for t in pilco._reward_terms())):
        return False
    # the same decision policy_loss_and_grad makes per evaluation: the plain additive reward, or an objective that says what
    # it adds as a function of the state trajectory (Safe-PILCO's risk term, host-evaluated reward terms: cotangent seeds)
    plain = type(pilco).predict is PILCO.predict
    own_objective = getattr(type(pilco), "trajectory_objective", None) is not PILCO.trajectory_objective
    seeded = (plain and bool(pilco._host_reward_terms())) or ((not plain) and own_objective and hasattr(pilco, "trajectory_objective"))
    if not plain and not seeded:
        return False
    if seeded:
        if not getattr(ctx, "lane_seeds", False):   # (a context whose batched calls take per-lane seed callbacks)
            return False
        traj = PILCO.predict_trajectory(pilco, pilco.m_init, pilco.S_init, pilco.horizon)[3]
        if pilco.trajectory_objective(np.asarray(traj)) is None:   # (a term without compute_reward_grad: finite differences, one walk at a time)
            return False
        return "seeded"
    return "plain"


def _optimize_policy_lanes(pilco, maxiter, restarts, verbose, seeded=False):
    """pilco.py:75-113 with its restarts side by side: restart i's start is drawn exactly where the reference draws it (the
    L-BFGS-B walks draw nothing), every walk is the scipy run of the sequential loop (lockstep_minimize: one thread per walk,
    a round of evaluations = ONE pilco_rollout_grad[_rbf]_batch call whose lanes are bit-identical to the solo calls), and the
    end points are compared as the reference compares them (compute_reward, first best wins)."""
    from .controllers import LinearController
    ctl = pilco.controller
    get, put = _policy_params(ctl)
    start = time.time()
    starts = [get()]
    for _ in range(restarts - 1):                                  # pilco.py:99
        ctl.randomize()
        starts.append(get())
    B, n = len(starts), starts[0].size
    parts = [np.arange(i * n, (i + 1) * n) for i in range(B)]
    linear = isinstance(ctl, LinearController)
    pilco.mgpr._user_factors = None
    pilco.mgpr._ensure_factorized()
    ctx, rw, H = pilco.ctx, pilco._reward_terms(), pilco.horizon
    m0, S0 = np.asarray(pilco.m_init, np.float64).reshape(-1), np.asarray(pilco.S_init, np.float64)
    cache = {}   # lane -> (u_i, value, gradient): a walk that has ended (or waits) is not evaluated again

    def eval_all(u):
        todo = [i for i in range(B) if i not in cache or not np.array_equal(cache[i][0], u[parts[i]])]
        if todo:
            us = [np.array(u[parts[i]], dtype=np.float64) for i in todo]
            nb = len(todo)
            mm, SS = np.tile(m0, (nb, 1)), np.tile(S0, (nb, 1, 1))
            extra = [0.0] * nb          # what the objective adds to lane k's additive reward (seeded objectives)
            no_seeds = set()            # lanes whose objective gave no cotangent seeds at this point: evaluated like the sequential loop does
            seed_kw = {}
            if seeded:
                def lane_seeds(k):
                    def fn(traj):
                        out = pilco.trajectory_objective(traj)
                        if out is None:   # (policy_loss_and_grad: this evaluation by finite differences; the lane's batch result is dropped)
                            no_seeds.add(k)
                            return np.zeros_like(traj)
                        extra[k] = float(out[0])
                        return out[1]
                    return fn
                seed_kw = dict(seed_fns=[lane_seeds(k) for k in range(nb)])
            if linear:
                base = ctl.policy_spec(True)
                nW = ctl.W.numpy().size
                specs = [dict(base, W=ui[:nW].reshape(ctl.W.shape), b=ui[nW:].reshape(-1)) for ui in us]
                r, dW, db = ctx.rollout_grad_batch(specs, rw, mm, SS, H, **seed_kw)
                for k, i in enumerate(todo):
                    cache[i] = (us[k], -(float(r[k]) + extra[k]), -np.concatenate([dW[k].ravel(), db[k].ravel()]))
            else:
                gp = ctl._gp
                bf, d, U = gp.num_datapoints, gp.num_dims, gp.num_outputs
                X = np.stack([ui[:bf * d].reshape(bf, d) for ui in us])
                Y = np.stack([ui[bf * d:bf * d + bf * U].reshape(bf, U) for ui in us])
                ls = np.stack([1e-3 + _softplus(ui[bf * d + bf * U:]).reshape(U, d) for ui in us])   # positive(lower=1e-3), as _policy_params
                nz = np.tile(np.asarray(ctl.noise, np.float64).reshape(-1), (nb, 1))
                spec = dict(kind=base_kind, state_dim=ctl.state_dim, control_dim=ctl.control_dim, max_action=ctl.max_action, squash=True)
                r, dX, dY, dl = ctx.rollout_grad_rbf_batch([spec] * nb, rw, mm, SS, H, X, Y, ls, nz, **seed_kw)
                for k, i in enumerate(todo):
                    ui = us[k]
                    g = np.concatenate([dX[k].ravel(), dY[k].ravel(), (dl[k] * _dsoftplus(ui[-dl[k].size:]).reshape(dl[k].shape)).ravel()])
                    cache[i] = (ui, -(float(r[k]) + extra[k]), -g)
            for k in sorted(no_seeds):   # the sequential loop's own evaluation of that point (finite differences), then the controller as it was
                keep = get()
                f, g = policy_loss_and_grad(pilco, us[k], put)
                put(keep)
                cache[todo[k]] = (us[k], f, g)
        vals = np.array([cache[i][1] for i in range(B)])
        grad = np.concatenate([cache[i][2] for i in range(B)])
        return vals, grad

    from . import _lib
    base_kind = _lib.POLICY_RBF
    u_end, _ = lockstep_minimize(eval_all, np.concatenate(starts), parts, maxiter=maxiter, wall=())
    best_u, best_r = None, None
    for i in range(B):
        put(u_end[parts[i]])
        r = float(pilco.compute_reward()[0, 0])
        if verbose:
            print("Controller's optimization: done in %.1f seconds with reward=%.3f." % (time.time() - start, r))
        if best_r is None or r > best_r:                           # pilco.py:104-106
            best_u, best_r = u_end[parts[i]].copy(), r
    put(best_u)
    return best_r


def optimize_policy(pilco, maxiter=50, restarts=1, verbose=True):
    if pilco.controller is None:
        raise ValueError("optimize_policy: the model has no controller (control_dim == 0)")
    lanes = _restart_lanes_apply(pilco, restarts) if restarts >= 2 else False
    if lanes:
        return _optimize_policy_lanes(pilco, maxiter, restarts, verbose, seeded=(lanes == "seeded"))
    get, put = _policy_params(pilco.controller)

    def run():
        start = time.time()
        res = minimize(lambda u: policy_loss_and_grad(pilco, u, put), get(), jac=True, method="L-BFGS-B",
                       options=dict(maxiter=maxiter))
        put(res.x)
        r = float(pilco.compute_reward()[0, 0])
        if verbose:
            print("Controller's optimization: done in %.1f seconds with reward=%.3f." % (time.time() - start, r))
        return res.x, r

    best_u, best_r = run()
    for _ in range(restarts - 1):                                  # pilco.py:94-107
        pilco.controller.randomize()
        u, r = run()
        if r > best_r:
            best_u, best_r = u, r
    put(best_u)
    return best_r
