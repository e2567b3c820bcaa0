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
    """A forward SDE dx = f(x, y, t) dt + g(t) dw on t in [0, T], discretised in N steps (reference sdes.py:19-30).
    Subclasses provide the coefficients (``sde``), the perturbation kernel (``marginal_prob``), the prior and ``copy``;
    the base supplies the Euler-Maruyama discretisation and the time reversal."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        """End time."""

    @abc.abstractmethod
    def sde(self, x, y, t, *args):
        """(drift [B,...], diffusion [B])."""

    @abc.abstractmethod
    def marginal_prob(self, x, y, t, *args):
        """(mean, std) of p_t(x | x0, y)."""

    @abc.abstractmethod
    def prior_sampling(self, shape, *args):
        """A sample of p_T."""

    @staticmethod
    @abc.abstractmethod
    def add_argparse_args(parent_parser):
        """Register the SDE's command-line flags."""

    @abc.abstractmethod
    def copy(self):
        """An independent instance with the same parameters."""

    def discretize(self, x, y, t, stepsize):
        """One Euler-Maruyama step of size ``stepsize``: (f, G) = (drift * dt, diffusion * sqrt(dt)), so that
        x_{i+1} = x_i + f + G z (reference sdes.py:72-89)."""
        drift, diffusion = self.sde(x, y, t)
        return drift * stepsize, diffusion * torch.sqrt(stepsize)

    def reverse(self, score_model, probability_flow=False):
        """Time reversal driven by ``score_model(x, y, t)`` (reference sdes.py:91-137): the reverse SDE, or with
        ``probability_flow`` the deterministic ODE with the same marginals."""
        return ReverseSDE(self, score_model, probability_flow)


class ReverseSDE:
    """Reverse-time counterpart of a forward SDE (Anderson 1982; Song et al. 2021):

        dx = [f(x, y, t) - w g(t)^2 score(x, y, t)] dt + g~(t) dw~,      (w, g~) = (1, g)  or, probability flow, (1/2, 0)

    The reference builds this object as a class nested in ``SDE.reverse`` and derived from the forward SDE's own class
    (sdes.py:98-137); here it is a plain wrapper with the same public methods (``N``, ``T``, ``sde``, ``rsde_parts``,
    ``discretize``) and the same arithmetic per method."""

    def __init__(self, forward, score_model, probability_flow=False):
        self.forward, self.score_model = forward, score_model
        self.probability_flow = probability_flow
        self.N = forward.N

    @property
    def T(self):
        return self.forward.T

    @property
    def _score_weight(self):
        return 0.5 if self.probability_flow else 1.0

    def rsde_parts(self, x, y, t, *args):
        """Every term of the reverse drift by name (used by diagnostics in the reference)."""
        f, g = self.forward.sde(x, y, t, *args)
        score = self.score_model(x, y, t, *args)
        from_score = -_bshape(g, x) ** 2 * score * self._score_weight
        return {"total_drift": f + from_score, "diffusion": torch.zeros_like(g) if self.probability_flow else g,
                "sde_drift": f, "sde_diffusion": g, "score_drift": from_score, "score": score}

    def sde(self, x, y, t, *args):
        parts = self.rsde_parts(x, y, t, *args)
        return parts["total_drift"], parts["diffusion"]

    def discretize(self, x, y, t, stepsize):
        """(rev_f, rev_G) of one step: x_{i-1} = x_i - rev_f + rev_G z."""
        f, G = self.forward.discretize(x, y, t, stepsize)
        rev_f = f - _bshape(G, x) ** 2 * self.score_model(x, y, t) * self._score_weight
        return rev_f, (torch.zeros_like(G) if self.probability_flow else G)


@SDERegistry.register("ouve")
class OUVESDE(SDE):
    """Ornstein-Uhlenbeck drift towards the noisy spectrogram y with a variance-exploding diffusion (reference
    sdes.py:144-232; Richter et al. 2023):

        dx = theta (y - x) dt + g(t) dw,     g(t) = sigma_min (sigma_max / sigma_min)^t sqrt(2 lambda),   lambda = ln(sigma_max / sigma_min)

    so that x_t | x_0, y is Gaussian with mean e^{-theta t} x_0 + (1 - e^{-theta t}) y and variance
    sigma_min^2 e^{-2 theta t} (e^{2 (theta + lambda) t} - 1) lambda / (theta + lambda)."""

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--theta", type=float, default=1.5, help="Stiffness of the OU drift (default 1.5).")
        parser.add_argument("--sigma-min", type=float, default=0.05, help="Diffusion scale at t = 0 (default 0.05).")
        parser.add_argument("--sigma-max", type=float, default=0.5, help="Diffusion scale at t = 1 (default 0.5).")
        parser.add_argument("--N", type=int, default=30, help="Discretisation steps of the reverse process (default 30).")
        parser.add_argument("--sampler_type", type=str, default="pc", help="Sampler ScoreModel.enhance uses ('pc' or 'ode').")
        return parser

    def __init__(self, theta, sigma_min, sigma_max, N=30, sampler_type="pc", **ignored_kwargs):
        super().__init__(N)
        self.theta, self.sigma_min, self.sigma_max = theta, sigma_min, sigma_max
        self.logsig = np.log(self.sigma_max / self.sigma_min)         # lambda (a NumPy float64, as in the reference)
        self.sampler_type = sampler_type

    def copy(self):
        return OUVESDE(self.theta, self.sigma_min, self.sigma_max, N=self.N, sampler_type=self.sampler_type)

    @property
    def T(self):
        return 1

    def _g(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t * np.sqrt(2 * self.logsig)

    def sde(self, x, y, t):
        return self.theta * (y - x), self._g(t)

    def _mean(self, x0, y, t):
        decay = _bshape(torch.exp(-self.theta * t), x0)
        return decay * x0 + (1 - decay) * y

    def _std(self, t):
        lam, th = self.logsig, self.theta
        var = (self.sigma_min ** 2 * torch.exp(-2 * th * t) * (torch.exp(2 * (th + lam) * t) - 1) * lam) / (th + lam)
        return torch.sqrt(var)

    def marginal_prob(self, x0, y, t):
        return self._mean(x0, y, t), self._std(t)

    def prior_sampling(self, shape, y):
        """x_T = y + std(T) z (the mean term e^{-theta T} x_0 is dropped: x_0 is unknown at inference time)."""
        if shape != y.shape:
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        std_T = self._std(torch.ones((y.shape[0],), device=y.device))
        return y + torch.randn_like(y) * _bshape(std_T, y)

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
