"""CPU oracle: flow-matching prior + Euler white-box sampler.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: flowmse/odes.py:86-100 (prior), flowmse/sampling/__init__.py:27-62
(time grid + loop), flowmse/sampling/odesolvers.py:42-47 (Euler update).
"""
import torch

from .ncsnpp_oracle import vf_forward


def prior_sampling(y, z, sigma_min=0.0, sigma_max=0.487):
    """odes.py:93-100 with the noise `z` made explicit; std = _std(1) (odes.py:86-88)."""
    t1 = torch.ones((y.shape[0],))
    std = (1 - t1) * sigma_min + t1 * sigma_max
    return y + z * std[:, None, None, None]


def time_grid(T_rev, t_eps, N):
    """sampling/__init__.py:45-53: linspace grid, step sizes, last step = t_eps."""
    ts = torch.linspace(T_rev, t_eps, N)
    steps = []
    for i in range(N):
        steps.append(ts[i] - ts[i + 1] if i != N - 1 else ts[-1])
    return ts, torch.stack(steps)


def euler_sample(VF_fn, Y, z, T_rev=1.0, t_eps=0.03, N=30, sigma_min=0.0, sigma_max=0.487):
    """ode_solver() -- sampling/__init__.py:36-60 with EulerODEsolver.update_fn."""
    with torch.no_grad():
        xt = prior_sampling(Y, z, sigma_min, sigma_max)
        ts, steps = time_grid(T_rev, t_eps, N)
        for i in range(N):
            vec_t = torch.ones(Y.shape[0]) * ts[i]
            xt = xt + VF_fn(xt, vec_t, Y) * (-steps[i])      # odesolvers.py:43-45
        return xt, N


def euler_sample_net(weights, cfg, Y, z, **kw):
    return euler_sample(lambda x, t, y: vf_forward(weights, cfg, x, t, y), Y, z, **kw)


def rk_sample(VF_fn, Y, z, tableau="rk4", T_rev=1.0, t_eps=0.03, N=30, sigma_min=0.0, sigma_max=0.487):
    """Fixed-step Heun / classical RK4 over the reference's time grid and step rule.  The reference has no
    fixed-step Runge-Kutta (its only RK is scipy's adaptive RK45, sampling/__init__.py:64-114), so this is pinned
    by COMPOSITION: the reference-pinned vector field inside the textbook tableau.  The last step of the grid
    lands on t = 0 where the field (h / t, log t) is singular; like the reference's own solver it is an Euler step."""
    with torch.no_grad():
        xt = prior_sampling(Y, z, sigma_min, sigma_max)
        ts, steps = time_grid(T_rev, t_eps, N)
        for i in range(N):
            t = torch.ones(Y.shape[0]) * ts[i]
            dt = -steps[i]
            k1 = VF_fn(xt, t, Y)
            if i == N - 1:
                xt = xt + k1 * dt
            elif tableau == "heun":
                k2 = VF_fn(xt + k1 * dt, t + dt, Y)
                xt = xt + (k1 + k2) * (0.5 * dt)
            elif tableau == "rk4":
                k2 = VF_fn(xt + k1 * (0.5 * dt), t + 0.5 * dt, Y)
                k3 = VF_fn(xt + k2 * (0.5 * dt), t + 0.5 * dt, Y)
                k4 = VF_fn(xt + k3 * dt, t + dt, Y)
                xt = xt + (k1 + 2 * k2 + 2 * k3 + k4) * (dt / 6.0)
            else:
                raise ValueError(tableau)
        return xt
