"""SDE classes: abstract SDE, its reverse, and the reflected variance-exploding SDE.

Host mirror of Reflected-Diffusion/sde_lib.py ("RD/sde_lib.py").  These are scalar-per-sample schedule
formulas ([B] tensors); the per-element sampler updates that consume them run in librdmi
(rdmi_em_update / rdmi_pc_sample evaluate sigma(t), g(t) in-kernel with the same fp32 operation order).
"""
import abc

import numpy as np
import torch


class SDE(abc.ABC):
    """RD/sde_lib.py:7-111."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        """End time of the SDE."""

    @abc.abstractmethod
    def sde(self, x, t):
        """(drift, diffusion) at (x, t)."""

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        """(mean, std) of p_t(x | x_0)."""

    @abc.abstractmethod
    def prior_sampling(self, shape):
        """One draw from p_T."""

    @abc.abstractmethod
    def prior_logp(self, z):
        """log p_T(z)."""

    def discretize(self, x, t):
        """Euler-Maruyama discretisation x_{i+1} = x_i + f + G z (RD/sde_lib.py:53-69)."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt, device=t.device))

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE / probability-flow ODE (RD/sde_lib.py:71-111)."""
        N, T = self.N, self.T
        fwd_sde, fwd_disc = self.sde, self.discretize

        class RSDE(self.__class__):
            def __init__(self):
                self.N = N
                self.probability_flow = probability_flow

            @property
            def T(self):
                return T

            def sde(self, x, t):
                drift, diffusion = fwd_sde(x, t)
                score = score_fn(x, t)
                drift = drift - diffusion[:, None, None, None] ** 2 * score * (0.5 if self.probability_flow else 1.)
                diffusion = torch.zeros_like(diffusion) if self.probability_flow else diffusion
                return drift, diffusion

            def discretize(self, x, t):
                f, G = fwd_disc(x, t)
                rev_f = f - G[:, None, None, None] ** 2 * score_fn(x, t) * (0.5 if self.probability_flow else 1.)
                rev_G = torch.zeros_like(G) if self.probability_flow else G
                return rev_f, rev_G

        return RSDE()


class RVESDE(SDE):
    """Reflected VE SDE on the unit cube, RD/sde_lib.py:114-161: sigma(t) = sigma_min (sigma_max/sigma_min)^t,
    zero drift, g(t) = sigma(t) sqrt(2 ln(sigma_max/sigma_min)), uniform prior."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, T=1):
        super().__init__(N)
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))
        self.N = N
        self.T_val = T

    @property
    def T(self):
        return self.T_val

    def _sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):
        g2 = torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)), device=t.device, dtype=torch.float32)
        return torch.zeros_like(x), self._sigma(t) * torch.sqrt(g2)

    def marginal_prob(self, x, t):
        return x, self._sigma(t)

    def prior_sampling(self, shape):
        return torch.rand(*shape)

    def prior_logp(self, z):
        return torch.zeros_like(z)

    def discretize(self, x, t):
        """SMLD discretisation (RD/sde_lib.py:153-161)."""
        timestep = (t * (self.N - 1) / self.T).long()
        sig = self.discrete_sigmas.to(t.device)
        sigma = sig[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), sig[timestep - 1])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent ** 2)
