"""Schedules of the reflected variance-exploding SDE and the reverse-time wrapper (host side).

API counterpart of Reflected-Diffusion/sde_lib.py ("RD/sde_lib.py"): `SDE` (abstract), `SDE.reverse(score_fn,
probability_flow)` and `RVESDE`.  Everything here is a per-sample scalar schedule ([B] tensors); the per-element
sampler arithmetic that consumes sigma(t) and g(t) runs in librdmi (rdmi_em_update / rdmi_pc_sample evaluate them
in-kernel with the same fp32 operation order).
"""
import abc
import math

import torch


def _bcast(v):
    """[B] -> [B,1,1,1] so that per-sample scalars multiply image-shaped tensors."""
    return v[:, None, None, None]


class SDE(abc.ABC):
    """Interface of RD/sde_lib.py:7-111: N discretisation steps, horizon T, forward coefficients, marginals, prior."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    # -- what a concrete SDE supplies ------------------------------------------------------------------------------
    @property
    @abc.abstractmethod
    def T(self): ...

    @abc.abstractmethod
    def sde(self, x, t):
        """-> (drift like x, diffusion [B])"""

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        """-> (mean like x, std [B]) of the perturbation kernel p_t(x | x_0)"""

    @abc.abstractmethod
    def prior_sampling(self, shape): ...

    @abc.abstractmethod
    def prior_logp(self, z): ...

    # -- derived ---------------------------------------------------------------------------------------------------
    def discretize(self, x, t):
        """One Euler-Maruyama step's (f, G) with step 1/N (RD/sde_lib.py:53-69)."""
        h = 1 / self.N
        f, g = self.sde(x, t)
        return f * h, g * torch.sqrt(torch.tensor(h, device=t.device))

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE, or the probability-flow ODE when `probability_flow` (RD/sde_lib.py:71-111).  As in the
        reference the result is an instance of a subclass of type(self) (so isinstance checks on the forward type hold),
        carrying N, T and `probability_flow`; only `sde` and `discretize` are overridden."""
        forward = self
        half = 0.5 if probability_flow else 1.0

        def _reverse_coeffs(f, g, x, t):
            f = f - _bcast(g) ** 2 * score_fn(x, t) * half
            return f, (torch.zeros_like(g) if probability_flow else g)

        def _init(rs):
            rs.N = forward.N
            rs.probability_flow = probability_flow

        members = {
            '__init__': _init,
            'T': property(lambda rs: forward.T),
            'sde': lambda rs, x, t: _reverse_coeffs(*forward.sde(x, t), x, t),
            'discretize': lambda rs, x, t: _reverse_coeffs(*forward.discretize(x, t), x, t),
        }
        return type('RSDE', (type(self),), members)()


class RVESDE(SDE):
    """Variance-exploding SDE reflected on the unit cube (RD/sde_lib.py:114-161):
    sigma(t) = sigma_min * (sigma_max / sigma_min) ** t, no drift, g(t) = sigma(t) * sqrt(2 ln(sigma_max / sigma_min)),
    prior U[0,1]^d (log-density 0)."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, T=1):
        super().__init__(N)
        self.sigma_min, self.sigma_max, self.T_val = sigma_min, sigma_max, T
        self.N = N
        # geometric ladder of the SMLD discretisation
        self.discrete_sigmas = torch.exp(torch.linspace(math.log(sigma_min), math.log(sigma_max), N))

    @property
    def T(self):
        return self.T_val

    def _sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def marginal_prob(self, x, t):
        return x, self._sigma(t)

    def sde(self, x, t):
        two_log_ratio = torch.tensor(2 * (math.log(self.sigma_max) - math.log(self.sigma_min)), device=t.device,
                                     dtype=torch.float32)
        return torch.zeros_like(x), self._sigma(t) * torch.sqrt(two_log_ratio)

    def prior_sampling(self, shape):
        return torch.rand(*shape)

    def prior_logp(self, z):
        return torch.zeros_like(z)

    def discretize(self, x, t):
        """SMLD ladder step (RD/sde_lib.py:153-161): G_i = sqrt(sigma_i^2 - sigma_{i-1}^2), sigma_{-1} = 0."""
        ladder = self.discrete_sigmas.to(t.device)
        i = (t * (self.N - 1) / self.T).long()
        below = torch.where(i == 0, torch.zeros_like(t), ladder[i - 1])
        return torch.zeros_like(x), torch.sqrt(ladder[i] ** 2 - below ** 2)
