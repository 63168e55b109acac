"""ODE-solver plugins (reference: flowmse/sampling/odesolvers.py:9-47).

``'euler'`` is the reference's only white-box solver.  ``'heun'`` and ``'rk4'`` are fixed-step higher-order
solvers registered through the same plugin point (the reference has none: its only Runge-Kutta is the adaptive
scipy RK45 black box of flowmse/sampling/__init__.py:64-114); they integrate the same dx/dt = VF(x,t,y)
backwards in time with the reference's step rule.

The network is only defined for t in [t_eps, T]: it divides by t (ncsnpp.py:398) and embeds log t.  The reference's
step rule makes the LAST step as long as the last grid time (sampling/__init__.py:53), i.e. it lands on t = 0, so a
higher-order stage evaluated at t + dt (or t + dt/2) of that step would query the field at / next to its
singularity.  The higher-order solvers therefore take a plain Euler update whenever a step reaches t = 0 (exactly
what the reference's own solver does there) and never evaluate the field at a time below the step's start ... end
range of an interior step, all of which are >= t_eps.
"""
import abc

import torch

from flowmse_amd.util.registry import Registry

ODEsolverRegistry = Registry("ODEsolver")


def axpy(x, k, dt):
    """x + dt * k.  On the GPU this is the library's ``flowse_axpy`` kernel (complex64 tensors on 'cuda'); for
    anything else (the plugin loop accepts arbitrary callables / devices, like the reference) plain tensor math."""
    if x.is_cuda and k.is_cuda and x.dtype == torch.complex64 and k.dtype == torch.complex64 and x.shape == k.shape:
        from flowmse_amd import _lib
        x, k = x.contiguous(), k.contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.flowse_axpy(_lib.ptr(x), _lib.ptr(k), float(dt), _lib.ptr(out), x.numel(),
                                            _lib.current_stream()))
        return out
    return x + k * dt


def _step_value(stepsize):
    """The step size as arithmetic operand: a Python float when it lives on the host (0-d CPU tensor or float; mixing
    a CPU tensor into device arithmetic would be an error for non-scalar shapes and a hidden copy otherwise), the
    device tensor itself when the caller keeps the grid on the device, like the reference's loop."""
    h = _host_scalar(stepsize)
    return stepsize if h is None else h


def _host_scalar(v):
    """Python float of a step size / time given as a float or a CPU tensor; None for a device tensor (reading it
    would block the host on the GPU queue)."""
    if torch.is_tensor(v):
        return None if v.is_cuda else float(v.flatten()[0])
    return float(v)


def _lands_on_zero(t, stepsize, t_start=None):
    """True when the step [t, t - stepsize] ends at (or numerically below) t = 0: the final step of the reference's
    grid, whose length equals the last grid time.  Decided on the host: ``get_white_box_solver`` hands the step size
    (and the solvers keep the step's start time) as host values.  Only a caller of ``update_fn`` that passes device
    tensors for BOTH pays a readback."""
    h = _host_scalar(stepsize)
    t0 = _host_scalar(t) if t_start is None else float(t_start)
    if h is None:
        h = float(stepsize)
    if t0 is None:
        t0 = float(t.flatten()[0])
    return t0 - h <= 1e-6 * max(1.0, abs(t0))


class ODEsolver(abc.ABC):
    nfe_per_step = 1
    # name of the library's fused implementation of this solver (flowse_rk_sample tableau), or None: plugin loop only
    fused_tableau = None
    # set by the sampler loop around each update_fn call (host float, None outside a loop): the start time of the step,
    # so that a solver can reason about the step's position on the grid without reading the device tensor `t` back
    step_start_time = None

    def __init__(self, ode, VF_fn):
        super().__init__()
        self.ode = ode
        self.VF_fn = VF_fn

    @abc.abstractmethod
    def update_fn(self, x, t, *args):
        pass


@ODEsolverRegistry.register("euler")
class EulerODEsolver(ODEsolver):
    fused_tableau = "euler"

    def update_fn(self, x, t, y, stepsize, *args):
        dt = -_step_value(stepsize)
        vectorfield = self.VF_fn(x, t, y)
        return axpy(x, vectorfield, dt)


@ODEsolverRegistry.register("heun")
class HeunODEsolver(ODEsolver):
    """Explicit trapezoid (RK2): k1 = f(x,t), k2 = f(x + dt k1, t + dt), x += dt/2 (k1 + k2)."""
    nfe_per_step = 2
    fused_tableau = "heun"

    def update_fn(self, x, t, y, stepsize, *args):
        dt = -_step_value(stepsize)
        k1 = self.VF_fn(x, t, y)
        if _lands_on_zero(t, stepsize, self.step_start_time):              # final step: Euler (the field is singular at its end point)
            return axpy(x, k1, dt)
        k2 = self.VF_fn(axpy(x, k1, dt), t + dt, y)
        return axpy(axpy(x, k1, 0.5 * dt), k2, 0.5 * dt)


@ODEsolverRegistry.register("rk4")
class RK4ODEsolver(ODEsolver):
    """Classical Runge-Kutta 4."""
    nfe_per_step = 4
    fused_tableau = "rk4"

    def update_fn(self, x, t, y, stepsize, *args):
        dt = -_step_value(stepsize)
        k1 = self.VF_fn(x, t, y)
        if _lands_on_zero(t, stepsize, self.step_start_time):              # final step: Euler (the field is singular at its end point)
            return axpy(x, k1, dt)
        th, te = t + 0.5 * dt, t + dt
        k2 = self.VF_fn(axpy(x, k1, 0.5 * dt), th, y)
        k3 = self.VF_fn(axpy(x, k2, 0.5 * dt), th, y)
        k4 = self.VF_fn(axpy(x, k3, dt), te, y)
        out = axpy(x, k1, dt / 6.0)
        out = axpy(out, k2, dt / 3.0)
        out = axpy(out, k3, dt / 3.0)
        return axpy(out, k4, dt / 6.0)
