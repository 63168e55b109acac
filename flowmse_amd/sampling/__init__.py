"""Samplers (reference: flowmse/sampling/__init__.py:27-62).

``get_white_box_solver`` keeps the reference signature and semantics: prior sample, ``torch.linspace(T_rev,
t_eps, N)`` time grid, step sizes ``t_i - t_{i+1}`` with the LAST step equal to ``t_{N-1}`` (so the trajectory
ends at t = 0), N solver updates, returns ``(x, N)``.

When ``VF_fn`` is a HIP-backed :class:`flowmse_amd.model.VFModel` and the solver is ``'euler'`` the whole loop
runs as one C-ABI call (``flowse_euler_sample``): N x (NCSN++ forward + fused Euler update) enqueued on the
current stream with no host synchronisation.  Any other callable ``VF_fn`` / registered solver goes through
the generic plugin loop, exactly like the reference.
"""
import torch

from .odesolvers import ODEsolver, ODEsolverRegistry

__all__ = ["ODEsolverRegistry", "ODEsolver", "get_white_box_solver", "time_grid"]


def time_grid(T_rev, t_eps, N, device="cpu"):
    """(timesteps, stepsizes) exactly as the reference loop builds them (sampling/__init__.py:45-53)."""
    timesteps = torch.linspace(T_rev, t_eps, N, device=device)
    steps = []
    for i in range(N):
        steps.append(timesteps[i] - timesteps[i + 1] if i != N - 1 else timesteps[-1])
    return timesteps, torch.stack(steps)


def get_white_box_solver(odesolver_name, ode, VF_fn, Y, Y_prior=None, T_rev=1.0, t_eps=0.03, N=30, z=None,
                         **kwargs):
    """Returns ``ode_solver() -> (x_result, N)``.  Extra keyword ``z``: explicit prior noise (reproducibility)."""
    odesolver_cls = ODEsolverRegistry.get_by_name(odesolver_name)
    odesolver = odesolver_cls(ode, VF_fn)
    fused = (odesolver_name == "euler" and hasattr(VF_fn, "euler_sample_") and Y.is_cuda)

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
                xt = VF_fn.euler_sample_(xt.contiguous(), Y.contiguous(), timesteps.tolist(), stepsizes.tolist())
                return xt, N
            timesteps = timesteps.to(Y.device)
            for i in range(N):
                t = timesteps[i]
                stepsize = stepsizes[i].to(Y.device)
                vec_t = torch.ones(Y.shape[0], device=Y.device) * t
                xt = odesolver.update_fn(xt, vec_t, Y, stepsize)
            return xt, N

    return ode_solver
