"""Optimiser glue.  Built in a later milestone."""


def optimize_mgpr(mgpr, restarts=1):
    raise NotImplementedError("GP hyper-parameter training on the device is not built yet; "
                              "set hyper-parameters through model.kernel.*.assign()")


def optimize_policy(pilco, maxiter=50, restarts=1, verbose=True):
    raise NotImplementedError("policy optimisation is not built yet")
