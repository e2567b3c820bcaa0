"""SDE library for the enhancement path: the SDE base (discretize / reverse) and OUVESDE.

Host-side mirror of reference sgmse/sdes.py:16-232 (SDERegistry, SDE, OUVESDE).  These classes only produce
*scalars* and small closures: on the native path every per-step constant is evaluated here with the same fp32 torch
expressions the reference evaluates (so the constants are bit-identical) and shipped to the HIP sampler as a table;
the tensor arithmetic of the loop runs in HIP kernels.  The tensor-valued methods (``sde``, ``marginal_prob``,
``prior_sampling``, ``discretize``, ``reverse``) are kept for API parity and for the generic Python sampler loop
used with predictors/correctors that have no fused kernel.
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


@SDERegistry.register("sbve")
class SBVESDE(SDE):
    """Schroedinger bridge with a variance-exploding noise schedule (reference sdes.py:235-313; Jukic et al. 2024):
    f = 0, g(t) = sqrt(c) k^t.  Only scalars are produced here; the sampler loop runs in the HIP library."""

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--N", type=int, default=50, help="Number of discretisation steps (50).")
        parser.add_argument("--k", type=float, default=2.6, help="Base of the diffusion coefficient (2.6).")
        parser.add_argument("--c", type=float, default=0.4, help="Scale of the diffusion coefficient (0.4).")
        parser.add_argument("--eps", type=float, default=1e-8, help="Numerical-stability constant (1e-8).")
        parser.add_argument("--sampler_type", type=str, default="ode")
        return parser

    def __init__(self, k, c, N=50, eps=1e-8, sampler_type="ode", **ignored_kwargs):
        super().__init__(N)
        self.k, self.c, self.N, self.eps, self.sampler_type = k, c, N, eps, sampler_type

    def copy(self):
        return SBVESDE(self.k, self.c, N=self.N)       # like the reference (sdes.py:263-264): eps / sampler_type reset to defaults

    @property
    def T(self):
        return 1

    def sde(self, x, y, t):
        return 0.0, torch.sqrt(torch.tensor(self.c)) * self.k ** t

    def _sigmas_alphas(self, t):
        alpha_t = torch.ones_like(t)
        alpha_T = torch.ones_like(t)
        logk2 = 2 * torch.log(torch.tensor(self.k))
        sigma_t = torch.sqrt((self.c * (self.k ** (2 * t) - 1.0)) / logk2)
        sigma_T = torch.sqrt((self.c * (self.k ** (2 * self.T) - 1.0)) / logk2)
        alpha_bart = alpha_t / (alpha_T + self.eps)
        sigma_bart = torch.sqrt(sigma_T ** 2 - sigma_t ** 2 + self.eps)
        return sigma_t, sigma_T, sigma_bart, alpha_t, alpha_T, alpha_bart

    def _mean(self, x0, y, t):
        sigma_t, sigma_T, sigma_bart, alpha_t, alpha_T, alpha_bart = self._sigmas_alphas(t)
        w_xt = alpha_t * sigma_bart ** 2 / (sigma_T ** 2 + self.eps)
        w_yt = alpha_bart * sigma_t ** 2 / (sigma_T ** 2 + self.eps)
        return _bshape(w_xt, x0) * x0 + _bshape(w_yt, y) * y

    def _std(self, t):
        sigma_t, sigma_T, sigma_bart, alpha_t, alpha_T, alpha_bart = self._sigmas_alphas(t)
        return (alpha_t * sigma_bart * sigma_t) / (sigma_T + self.eps)

    def marginal_prob(self, x0, y, t):
        return self._mean(x0, y, t), self._std(t)

    def prior_sampling(self, shape, y):
        if shape != y.shape:
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        return y

    def prior_logp(self, z):
        raise NotImplementedError("prior_logp for SBVE SDE not yet implemented!")

    def sb_step_table(self, eps: float, sampler_type: str, N: int = None):
        """Per-step weights of get_sb_sampler (sampling/__init__.py:145-249), fp32 tensors of length N evaluated with the
        reference's expressions on t = linspace(T, eps, N+1):  x <- w_prev x + w_est model(x, y, t) + w_y y + w_z z."""
        N = self.N if N is None else N
        ts = torch.linspace(self.T, eps, N + 1)
        s, sT, sbar, a, aT, abar = self._sigmas_alphas(ts)
        sp, sbp, ap = s[:-1], sbar[:-1], a[:-1]            # values at the previous time step
        st, sbt, at, sTt, aTt = s[1:], sbar[1:], a[1:], sT, aT[1:]   # sigma_T is a 0-dim tensor (sdes.py:281-282)
        if sampler_type == "sde":
            w_prev = at * st ** 2 / (ap * sp ** 2 + self.eps)
            tmp = 1 - st ** 2 / (sp ** 2 + self.eps)
            w_est = at * tmp
            w_z = at * st * torch.sqrt(tmp)
            w_z[-1] = 0.0                                   # no noise on the last step (sampling/__init__.py:183-184)
            w_y = torch.zeros_like(w_prev)
        elif sampler_type == "ode":
            w_prev = at * st * sbt / (ap * sp * sbp + self.eps)
            w_est = at / (sTt ** 2 + self.eps) * (sbt ** 2 - sbp * st * sbt / (sp + self.eps))
            w_y = at / (aTt * sTt ** 2 + self.eps) * (st ** 2 - sp * st * sbt / (sbp + self.eps))
            w_z = torch.zeros_like(w_prev)
        else:
            raise ValueError("Invalid type. Choose 'ode' or 'sde'.")
        return dict(t=ts[1:].clone(), w_prev=w_prev, w_est=w_est, w_y=w_y, w_z=w_z)
