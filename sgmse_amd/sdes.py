"""SDE library for the enhancement path: the SDE base (discretize / reverse) and OUVESDE.

Host-side mirror of reference sgmse/sdes.py:16-232 (SDERegistry, SDE, OUVESDE).  These classes only produce
*scalars* and small closures: on the native path every per-step constant is evaluated here with the same fp32 torch
expressions the reference evaluates (so the constants are bit-identical) and shipped to the HIP sampler as a table;
the tensor arithmetic of the loop runs in HIP kernels.  The tensor-valued methods (``sde``, ``marginal_prob``,
``prior_sampling``, ``discretize``, ``reverse``) are kept for API parity and for the generic Python sampler loop
used with predictors/correctors that have no fused kernel.  SBVESDE (Schroedinger bridge) is out of scope.
"""
import abc
import warnings

import numpy as np
import torch

from .util.registry import Registry

SDERegistry = Registry("SDE")


def _bshape(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return v.reshape(-1, *([1] * (like.dim() - 1)))


class SDE(abc.ABC):
    """Abstract SDE; N = number of discretisation steps (reference sdes.py:19-30)."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        ...

    @abc.abstractmethod
    def sde(self, x, y, t, *args):
        ...

    @abc.abstractmethod
    def marginal_prob(self, x, y, t, *args):
        ...

    @abc.abstractmethod
    def prior_sampling(self, shape, *args):
        ...

    @staticmethod
    @abc.abstractmethod
    def add_argparse_args(parent_parser):
        ...

    @abc.abstractmethod
    def copy(self):
        ...

    def discretize(self, x, y, t, stepsize):
        """Euler-Maruyama discretisation f = drift*dt, G = diffusion*sqrt(dt) (reference sdes.py:72-89)."""
        dt = stepsize
        drift, diffusion = self.sde(x, y, t)
        return drift * dt, diffusion * torch.sqrt(dt)

    def reverse(oself, score_model, probability_flow=False):
        """The reverse-time SDE / probability-flow ODE as an object with ``sde``, ``rsde_parts`` and ``discretize``
        (reference sdes.py:91-137)."""
        N, T = oself.N, oself.T
        fwd_sde, fwd_disc = oself.sde, oself.discretize

        class RSDE(oself.__class__):
            def __init__(self):
                self.N = N
                self.probability_flow = probability_flow

            @property
            def T(self):
                return T

            def rsde_parts(self, x, y, t, *args):
                sde_drift, sde_diffusion = fwd_sde(x, y, t, *args)
                score = score_model(x, y, t, *args)
                score_drift = -_bshape(sde_diffusion, x) ** 2 * score * (0.5 if self.probability_flow else 1.0)
                diffusion = torch.zeros_like(sde_diffusion) if self.probability_flow else sde_diffusion
                return {"total_drift": sde_drift + score_drift, "diffusion": diffusion, "sde_drift": sde_drift,
                        "sde_diffusion": sde_diffusion, "score_drift": score_drift, "score": score}

            def sde(self, x, y, t, *args):
                parts = self.rsde_parts(x, y, t, *args)
                return parts["total_drift"], parts["diffusion"]

            def discretize(self, x, y, t, stepsize):
                f, G = fwd_disc(x, y, t, stepsize)
                rev_f = f - _bshape(G, x) ** 2 * score_model(x, y, t) * (0.5 if self.probability_flow else 1.0)
                rev_G = torch.zeros_like(G) if self.probability_flow else G
                return rev_f, rev_G

        return RSDE()


@SDERegistry.register("ouve")
class OUVESDE(SDE):
    """Ornstein-Uhlenbeck variance-exploding SDE  dx = theta (y - x) dt + sigma(t) dw,
    sigma(t) = sigma_min (sigma_max/sigma_min)^t sqrt(2 log(sigma_max/sigma_min))  (reference sdes.py:144-232)."""

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--theta", type=float, default=1.5, help="Stiffness of the Ornstein-Uhlenbeck process (1.5).")
        parser.add_argument("--sigma-min", type=float, default=0.05, help="Smallest sigma (0.05).")
        parser.add_argument("--sigma-max", type=float, default=0.5, help="Largest sigma (0.5).")
        parser.add_argument("--N", type=int, default=30, help="Number of discretisation steps (30).")
        parser.add_argument("--sampler_type", type=str, default="pc", help="Sampler used by ScoreModel.enhance ('pc').")
        return parser

    def __init__(self, theta, sigma_min, sigma_max, N=30, sampler_type="pc", **ignored_kwargs):
        super().__init__(N)
        self.theta = theta
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.logsig = np.log(self.sigma_max / self.sigma_min)
        self.N = N
        self.sampler_type = sampler_type

    def copy(self):
        return OUVESDE(self.theta, self.sigma_min, self.sigma_max, N=self.N, sampler_type=self.sampler_type)

    @property
    def T(self):
        return 1

    def sde(self, x, y, t):
        drift = self.theta * (y - x)
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        diffusion = sigma * np.sqrt(2 * self.logsig)
        return drift, diffusion

    def _mean(self, x0, y, t):
        w = _bshape(torch.exp(-self.theta * t), x0)
        return w * x0 + (1 - w) * y

    def _std(self, t):
        smin, theta, logsig = self.sigma_min, self.theta, self.logsig
        return torch.sqrt((smin ** 2 * torch.exp(-2 * theta * t) * (torch.exp(2 * (theta + logsig) * t) - 1) * logsig)
                          / (theta + logsig))

    def marginal_prob(self, x0, y, t):
        return self._mean(x0, y, t), self._std(t)

    def prior_sampling(self, shape, y):
        if shape != y.shape:
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        std = self._std(torch.ones((y.shape[0],), device=y.device))
        return y + torch.randn_like(y) * _bshape(std, y)

    def prior_logp(self, z):
        raise NotImplementedError("prior_logp for OU SDE not yet implemented!")

    # -- per-step constants of the discretised sampler (what the HIP loop consumes) --------------------------
    def step_table(self, eps: float, snr: float, N: int = None):
        """fp32 tensors of length N, evaluated with the expressions of reference sampling/__init__.py:56-62,
        correctors.py:72-79 and sdes.py:72-89: t, dt, std, ald_eps, ald_noise, G, G2."""
        N = self.N if N is None else N
        ts = torch.linspace(self.T, eps, N)
        dt = torch.empty_like(ts)
        dt[:-1] = ts[:-1] - ts[1:]
        dt[-1] = ts[-1]
        std = self._std(ts)
        ald_eps = (snr * std) ** 2 * 2
        ald_noise = torch.sqrt(ald_eps * 2)
        g = self.sde(torch.zeros(1), torch.zeros(1), ts)[1]
        G = g * torch.sqrt(dt)
        return dict(t=ts, dt=dt, std=std, ald_eps=ald_eps, ald_noise=ald_noise, g=g, G=G, G2=G ** 2)
