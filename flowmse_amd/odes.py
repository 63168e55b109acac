"""Flow-matching probability path used by the sampler.

API mirror of the reference's ``ODERegistry`` / ``FLOWMATCHING`` (flowmse/odes.py:17-107) -- same registry name
``"flowmatching"``, constructor keywords and method names -- written for this package:

    mean(t)  = (1 - t) * x0 + t * y
    sigma(t) = (1 - t) * sigma_min + t * sigma_max          (defaults 0.0 / 0.487)
    prior    : x_T = y + sigma(1) * z,  z ~ CN(0, 1)         (odes.py:93-100)

Only ``prior_sampling`` is on the sampling hot path; on a GPU it is one ``flowse_prior_sample`` launch.
"""
import warnings

import torch

from flowmse_amd.util.registry import Registry

ODERegistry = Registry("ODE")


class ODE:
    """Minimal base: what the sampler needs from a probability path."""

    def marginal_prob(self, x0, t, y):
        raise NotImplementedError

    def prior_sampling(self, shape, y, z=None):
        raise NotImplementedError

    def copy(self):
        raise NotImplementedError


def _bcast(t):
    return t.reshape(-1, 1, 1, 1)


@ODERegistry.register("flowmatching")
class FLOWMATCHING(ODE):
    def __init__(self, sigma_min=0.0, sigma_max=0.487, **_unused):
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    @staticmethod
    def add_argparse_args(parser):
        for name, default in (("--sigma_min", 0.0), ("--sigma_max", 0.487)):
            parser.add_argument(name, type=float, default=default)
        return parser

    def copy(self):
        return type(self)(self.sigma_min, self.sigma_max)

    # the reference leaves the drift itself unimplemented as well (odes.py:81)
    def ode(self, x, t, *args):
        return None

    # ---- closed forms of the path -------------------------------------------------------------
    def _std(self, t):
        return self.sigma_min * (1 - t) + self.sigma_max * t

    def _mean(self, x0, t, y):
        return _bcast(1 - t) * x0 + _bcast(t) * y

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)

    def der_mean(self, x0, t, y):
        return y - x0

    def der_std(self, t):
        return self.sigma_max - self.sigma_min

    # ---- prior --------------------------------------------------------------------------------
    def prior_std(self):
        """sigma(1) as the float32 number the reference evaluates for a batch of ones (odes.py:96)."""
        return float(self._std(torch.ones(1, dtype=torch.float32)))

    def prior_sampling(self, shape, y, z=None):
        """Returns ``(x_T, z)``.  ``z`` may be passed in for reproducible trajectories."""
        if tuple(shape) != tuple(y.shape):
            warnings.warn(f"prior_sampling: requested shape {tuple(shape)} differs from y {tuple(y.shape)}; using y's")
        z = torch.randn_like(y) if z is None else z
        if not y.is_cuda:
            return y + z * _bcast(self._std(torch.ones(y.shape[0], device=y.device))), z
        from flowmse_amd import _lib
        y, zc = y.contiguous(), z.contiguous()
        x_T = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _lib.check(_lib.lib.flowse_prior_sample(_lib.ptr(y), _lib.ptr(zc), self.prior_std(), _lib.ptr(x_T),
                                                    y.numel(), _lib.current_stream()))
        return x_T, z
