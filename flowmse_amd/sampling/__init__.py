"""Samplers (reference: flowmse/sampling/__init__.py:27-62).

``get_white_box_solver`` keeps the reference signature and semantics: prior sample, ``torch.linspace(T_rev,
t_eps, N)`` time grid, step sizes ``t_i - t_{i+1}`` with the LAST step equal to ``t_{N-1}`` (so the trajectory
ends at t = 0), N solver updates, returns ``(x, N)``.

When ``VF_fn`` is a HIP-backed :class:`flowmse_amd.model.VFModel` and the solver is one the library implements
(``'euler'``, the reference's; ``'heun'`` / ``'rk4'``, the fixed-step plugins) the whole loop runs as one C-ABI call
(``flowse_rk_sample``): N x stages x (NCSN++ forward + solver update fused into the head kernel) enqueued on the
current stream with no host synchronisation.  Any other callable ``VF_fn`` / registered solver goes through the
generic plugin loop, exactly like the reference; that loop keeps its time grid on the host, so it never reads a
device tensor back either.
"""
import contextlib

import torch

from .odesolvers import ODEsolver, ODEsolverRegistry

__all__ = ["ODEsolverRegistry", "ODEsolver", "get_white_box_solver", "get_black_box_solver", "time_grid"]


def time_grid(T_rev, t_eps, N, device="cpu"):
    """(timesteps, stepsizes) exactly as the reference loop builds them (sampling/__init__.py:45-53)."""
    timesteps = torch.linspace(T_rev, t_eps, N, device=device)
    steps = []
    for i in range(N):
        steps.append(timesteps[i] - timesteps[i + 1] if i != N - 1 else timesteps[-1])
    return timesteps, torch.stack(steps)


def _frozen(VF_fn):
    """``VF_fn.weights_frozen()`` when the field is a HIP-backed model (its per-call weight-version scan then runs once
    per sampler loop instead of once per network evaluation), else a no-op context."""
    f = getattr(VF_fn, "weights_frozen", None)
    return f() if callable(f) else contextlib.nullcontext()


def _fused_tableau(odesolver_cls):
    """Name of the library's fused implementation of this solver class, or None (plugin loop).

    Only a class that opts in ITSELF qualifies: ``fused_tableau`` must be set in the class's own ``__dict__`` and its
    ``update_fn`` must be the one defined next to it.  A plugin that subclasses a built-in solver
    (``class My(EulerODEsolver)``) and overrides ``update_fn`` inherits the attribute but not the claim -- it goes
    through the generic loop, where its ``update_fn`` is what runs."""
    name = odesolver_cls.__dict__.get("fused_tableau")
    if name is None or "update_fn" not in odesolver_cls.__dict__:
        return None
    return name


def get_white_box_solver(odesolver_name, ode, VF_fn, Y, Y_prior=None, T_rev=1.0, t_eps=0.03, N=30, z=None,
                         **kwargs):
    """Returns ``ode_solver() -> (x_result, N)``.  Extra keyword ``z``: explicit prior noise (reproducibility)."""
    odesolver_cls = ODEsolverRegistry.get_by_name(odesolver_name)
    odesolver = odesolver_cls(ode, VF_fn)
    fused = _fused_tableau(odesolver_cls) is not None and hasattr(VF_fn, "rk_sample_") and Y.is_cuda

    def ode_solver(Y_prior=Y_prior):
        with torch.no_grad():
            if Y_prior is None:
                Y_prior = Y
            if z is not None:
                xt, _ = ode.prior_sampling(Y_prior.shape, Y_prior, z)
            else:
                xt, _ = ode.prior_sampling(Y_prior.shape, Y_prior)
            xt = xt.to(Y_prior.device)
            # host copy of the grid: the values equal torch.linspace(..., device=Y.device) of the reference
            timesteps, stepsizes = time_grid(T_rev, t_eps, N)
            if fused:
                xt = VF_fn.rk_sample_(xt.contiguous(), Y.contiguous(), timesteps.tolist(), stepsizes.tolist(),
                                      _fused_tableau(odesolver_cls))
                return xt, N
            try:
                with _frozen(VF_fn):
                    for i in range(N):
                        # t and the step size stay host values (0-d CPU tensors): `ones(B) * t` is a fill kernel with
                        # a scalar argument, and a solver may inspect the step ("does it land on t = 0?") with no
                        # readback
                        t = timesteps[i]
                        stepsize = stepsizes[i]
                        vec_t = torch.ones(Y.shape[0], device=Y.device) * float(t)
                        odesolver.step_start_time = float(t)
                        xt = odesolver.update_fn(xt, vec_t, Y, stepsize)
            finally:
                odesolver.step_start_time = None
            return xt, N

    return ode_solver


def to_flattened_numpy(x):
    """Flatten a torch tensor and convert it to numpy (sampling/__init__.py:17-19)."""
    return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
    """Form a torch tensor with the given shape from a flattened numpy array (sampling/__init__.py:22-24)."""
    return torch.from_numpy(x.reshape(shape))


def get_black_box_solver(ode, VF_fn, y, rtol=1e-5, atol=1e-5, T_rev=1.0, t_eps=0.03, N=30, method="RK45",
                         device="cuda", z=None, **kwargs):
    """Adaptive black-box sampler (reference: flowmse/sampling/__init__.py:64-114): scipy ``solve_ivp`` on the
    flattened complex state from T_rev down to t_eps (NOT to 0), each right-hand side evaluation being one call
    of ``VF_fn`` (host <-> device round trip per evaluation, as in the reference).  Returns ``(x, nfe)``.
    ``evaluate.py`` imports but never calls it; provided for API completeness."""
    from scipy import integrate

    def ode_solver(**solver_kwargs):
        with torch.no_grad():
            x = (ode.prior_sampling(y.shape, y, z)[0] if z is not None else ode.prior_sampling(y.shape, y)[0]).to(device)

            def ode_func(t, xf):
                xt = from_flattened_numpy(xf, y.shape).to(device).type(torch.complex64)
                vec_t = torch.ones(y.shape[0], device=xt.device) * t
                return to_flattened_numpy(VF_fn(xt, vec_t, y))

            with _frozen(VF_fn):
                solution = integrate.solve_ivp(ode_func, (T_rev, t_eps), to_flattened_numpy(x), rtol=rtol, atol=atol,
                                               method=method, **solver_kwargs)
            x = torch.tensor(solution.y[:, -1]).reshape(y.shape).to(device).type(torch.complex64)
            return x, solution.nfev

    return ode_solver
