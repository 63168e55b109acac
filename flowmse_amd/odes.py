"""Flow-matching probability path (reference: flowmse/odes.py:17-107).

mu_t = (1 - t) x0 + t y,  sigma_t = (1 - t) sigma_min + t sigma_max,  prior x_T = y + sigma(1) z.
Only ``prior_sampling`` / ``_std`` are on the sampling hot path; the remaining methods are the tiny closed
forms the reference exposes (kept so that code written against ``model.ode`` keeps working).
"""
import abc
import warnings

import torch

from flowmse_amd.util.registry import Registry

ODERegistry = Registry("ODE")


class ODE(abc.ABC):
    @abc.abstractmethod
    def marginal_prob(self, x, t, *args):
        pass

    @abc.abstractmethod
    def prior_sampling(self, shape, *args):
        pass

    @abc.abstractmethod
    def copy(self):
        pass


@ODERegistry.register("flowmatching")
class FLOWMATCHING(ODE):
    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--sigma_min", type=float, default=0.00)
        parser.add_argument("--sigma_max", type=float, default=0.487)
        return parser

    def __init__(self, sigma_min=0.00, sigma_max=0.487, **ignored_kwargs):
        super().__init__()
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max

    def copy(self):
        return FLOWMATCHING(self.sigma_min, self.sigma_max)

    def ode(self, x, t, *args):
        pass

    def _mean(self, x0, t, y):
        return (1 - t)[:, None, None, None] * x0 + t[:, None, None, None] * y

    def _std(self, t):
        return (1 - t) * self.sigma_min + t * self.sigma_max

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)

    def prior_std(self):
        """sigma(t=1) as the float32 value the reference computes (odes.py:96)."""
        return float(self._std(torch.ones((1,), dtype=torch.float32))[0])

    def prior_sampling(self, shape, y, z=None):
        """x_T = y + z * sigma(1) (odes.py:93-100).  `z` may be supplied for reproducible sampling."""
        if tuple(shape) != tuple(y.shape):
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        if z is None:
            z = torch.randn_like(y)
        if y.is_cuda:
            from flowmse_amd import _lib
            y = y.contiguous()
            z = z.contiguous()
            x_T = torch.empty_like(y)
            with torch.cuda.device(y.device):
                _lib.check(_lib.lib.flowse_prior_sample(_lib.ptr(y), _lib.ptr(z), self.prior_std(), _lib.ptr(x_T),
                                                        y.numel(), _lib.current_stream()))
            return x_T, z
        std = self._std(torch.ones((y.shape[0],), device=y.device))
        return y + z * std[:, None, None, None], z

    def der_mean(self, x0, t, y):
        return y - x0

    def der_std(self, t):
        return self.sigma_max - self.sigma_min
