"""Oracle restatement of the OUVE SDE and the PC / fixed-step samplers
(TEST INFRASTRUCTURE ONLY).

Follows sgmse/sdes.py:72-89 (discretize), :91-137 (reverse / RSDE),
:144-232 (OUVESDE); sgmse/sampling/__init__.py:26-70 (pc_sampler);
sgmse/sampling/predictors.py:41-76; sgmse/sampling/correctors.py:37-94.

Noise is *replayed*: every sampler takes a ``noise`` callable ``noise(like)``
returning a complex64 standard-normal tensor (what ``torch.randn_like`` gives
for complex input: Re, Im ~ N(0, 1/2)), so that the HIP path can consume the
identical draws.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Tuple

import numpy as np
import torch


@dataclass
class OUVE:
    """sdes.py:155-179.  VoiceBank-DEMAND checkpoint values by default (:148-150)."""
    theta: float = 1.5
    sigma_min: float = 0.05
    sigma_max: float = 0.5
    N: int = 30

    @property
    def logsig(self):
        return np.log(self.sigma_max / self.sigma_min)                 # :177 (numpy float64)

    T = 1                                                              # :184-186

    def diffusion(self, t: torch.Tensor) -> torch.Tensor:
        """g(t), sdes.py:188-196."""
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return sigma * np.sqrt(2 * self.logsig)

    def drift(self, x, y):
        return self.theta * (y - x)                                    # :189

    def std(self, t: torch.Tensor) -> torch.Tensor:
        """_std(t), sdes.py:206-219."""
        smin, th, ls = self.sigma_min, self.theta, self.logsig
        return torch.sqrt(
            (smin ** 2 * torch.exp(-2 * th * t) * (torch.exp(2 * (th + ls) * t) - 1) * ls) / (th + ls))

    def prior(self, y, noise):
        """prior_sampling, sdes.py:224-229."""
        std = self.std(torch.ones((y.shape[0],)))
        return y + noise(y) * std[:, None, None, None]


def timesteps(sde: OUVE, eps: float) -> torch.Tensor:
    return torch.linspace(sde.T, eps, sde.N)                           # sampling/__init__.py:56


def step_table(sde: OUVE, eps: float, snr: float) -> Dict[str, torch.Tensor]:
    """All per-step scalars of the PC loop as fp32 tensors of length N, computed with
    the same torch expressions the reference evaluates."""
    ts = timesteps(sde, eps)
    dt = torch.empty_like(ts)
    dt[:-1] = ts[:-1] - ts[1:]                                         # :59-60
    dt[-1] = ts[-1]                                                    # :61-62
    std = sde.std(ts)
    ald_eps = (snr * std) ** 2 * 2                                     # correctors.py:77
    ald_noise = torch.sqrt(ald_eps * 2)                                # :79
    g = sde.diffusion(ts)
    G = g * torch.sqrt(dt)                                             # sdes.py:88
    return dict(t=ts, dt=dt, std=std, ald_eps=ald_eps, ald_noise=ald_noise, g=g, G=G, G2=G ** 2)


def ald_update(sde, score, x, y, t, snr, noise, n_steps=1):
    """AnnealedLangevinDynamics.update_fn, correctors.py:69-81."""
    std = sde.std(t)
    x_mean = x
    for _ in range(n_steps):
        grad = score(x, y, t)
        z = noise(x)
        step = (snr * std) ** 2 * 2
        x_mean = x + step[:, None, None, None] * grad
        x = x_mean + z * torch.sqrt(step * 2)[:, None, None, None]
    return x, x_mean


def langevin_update(sde, score, x, y, t, snr, noise, n_steps=1):
    """LangevinCorrector.update_fn, correctors.py:44-56 (batch-mean norms)."""
    x_mean = x
    for _ in range(n_steps):
        grad = score(x, y, t)
        z = noise(x)
        gn = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
        zn = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
        step = ((snr * zn / gn) ** 2 * 2).unsqueeze(0)
        x_mean = x + step[:, None, None, None] * grad
        x = x_mean + z * torch.sqrt(step * 2)[:, None, None, None]
    return x, x_mean


def revdiff_update(sde, score, x, y, t, stepsize, noise, probability_flow=False):
    """ReverseDiffusionPredictor.update_fn (predictors.py:60-65) over
    RSDE.discretize (sdes.py:130-135) over SDE.discretize (:72-89).
    probability_flow=True gives the fixed-step PF-ODE Euler step of SURVEY 8-a9
    (score weight 1/2, no noise)."""
    f = sde.drift(x, y) * stepsize
    G = sde.diffusion(t) * torch.sqrt(stepsize)
    rev_f = f - G[:, None, None, None] ** 2 * score(x, y, t) * (0.5 if probability_flow else 1.0)
    rev_G = torch.zeros_like(G) if probability_flow else G
    z = noise(x)
    x_mean = x - rev_f
    x = x_mean + rev_G[:, None, None, None] * z
    return x, x_mean


def euler_maruyama_update(sde, score, x, y, t, noise):
    """EulerMaruyamaPredictor.update_fn, predictors.py:46-52 with RSDE.sde (:114-128)."""
    dt = -1.0 / sde.N
    z = noise(x)
    g = sde.diffusion(t)
    f = sde.drift(x, y) - g[:, None, None, None] ** 2 * score(x, y, t)
    x_mean = x + f * dt
    x = x_mean + g[:, None, None, None] * np.sqrt(-dt) * z
    return x, x_mean


def pc_sample(sde: OUVE, score: Callable, y: torch.Tensor, noise: Callable, *, eps=0.03, snr=0.5,
              corrector="ald", corrector_steps=1, predictor="reverse_diffusion", denoise=True,
              probability_flow=False, trace: List = None) -> Tuple[torch.Tensor, int]:
    """pc_sampler, sampling/__init__.py:52-68.  ``probability_flow`` here selects the
    fixed-step PF-ODE predictor (the reference's Predictor ignores the flag,
    predictors.py:18)."""
    with torch.no_grad():
        xt = sde.prior(y, noise)
        ts = timesteps(sde, eps)
        xm = xt
        for i in range(sde.N):
            t = ts[i]
            stepsize = t - ts[i + 1] if i != sde.N - 1 else ts[-1]
            vec_t = torch.ones(y.shape[0]) * t
            if corrector == "ald":
                xt, xm = ald_update(sde, score, xt, y, vec_t, snr, noise, corrector_steps)
            elif corrector == "langevin":
                xt, xm = langevin_update(sde, score, xt, y, vec_t, snr, noise, corrector_steps)
            elif corrector != "none":
                raise ValueError(corrector)
            if predictor == "reverse_diffusion":
                xt, xm = revdiff_update(sde, score, xt, y, vec_t, stepsize, noise, probability_flow)
            elif predictor == "euler_maruyama":
                xt, xm = euler_maruyama_update(sde, score, xt, y, vec_t, noise)
            elif predictor == "none":
                xm = xt                                                # NonePredictor.update_fn returns (x, x), predictors.py:69-76
            else:
                raise ValueError(predictor)
            if trace is not None:
                trace.append(xt.clone())
        n_corr = 0 if corrector == "none" else corrector_steps
        return (xm if denoise else xt), sde.N * (n_corr + 1)


def ode_sample_adaptive(sde: OUVE, score: Callable, y: torch.Tensor, noise: Callable, *, eps=0.03, rtol=1e-5, atol=1e-5,
                        method="RK45") -> Tuple[torch.Tensor, int]:
    """The reference's black-box probability-flow sampler with ``denoise=False`` (sampling/__init__.py:96-143; with the default
    ``denoise=True`` the reference raises TypeError at :101).  The solver is the third-party scipy.integrate.solve_ivp
    (reference pin scipy==1.10.1, requirements_version.txt:14; Dormand-Prince RK45 with scipy's step control -- not restated here,
    called as the reference calls it at :128-131); restated are the drift handed to it -- rsde.sde(...)[0] with
    probability_flow=True: theta (y - x) - g(t)^2 score / 2 (sdes.py:113-128, :188-196) -- and the flattening of the complex
    state (utils/other.py: to_flattened_numpy / from_flattened_numpy)."""
    from scipy import integrate
    with torch.no_grad():
        x = sde.prior(y, noise)

        def ode_func(t, xflat):
            xt = torch.from_numpy(xflat.reshape(tuple(y.shape))).type(torch.complex64)
            vec_t = torch.ones(y.shape[0]) * t
            g = sde.diffusion(vec_t)
            drift = sde.drift(xt, y) - g[:, None, None, None] ** 2 * score(xt, y, vec_t) * 0.5
            return drift.detach().cpu().numpy().reshape((-1,))

        sol = integrate.solve_ivp(ode_func, (sde.T, eps), x.detach().cpu().numpy().reshape((-1,)), rtol=rtol, atol=atol, method=method)
        return torch.tensor(sol.y[:, -1]).reshape(y.shape).type(torch.complex64), int(sol.nfev)


class NoiseReplay:
    """Deterministic complex standard-normal stream (SURVEY 8-d 'Noise').
    Draw k has shape ``like.shape``; values come from a dedicated CPU generator so
    the sequence can be materialised up front for the HIP path."""

    def __init__(self, seed: int = 7):
        self.gen = torch.Generator().manual_seed(seed)
        self.draws: List[torch.Tensor] = []

    def __call__(self, like: torch.Tensor) -> torch.Tensor:
        z = torch.randn(like.shape, dtype=like.dtype, generator=self.gen)
        self.draws.append(z)
        return z


@dataclass
class SBVE:
    """SBVESDE scalars (sdes.py:246-288)."""
    k: float = 2.6
    c: float = 0.4
    N: int = 50
    eps: float = 1e-8
    T = 1

    def sigmas_alphas(self, t):
        alpha_t = torch.ones_like(t)
        alpha_T = torch.ones_like(t)
        sigma_t = torch.sqrt((self.c * (self.k ** (2 * t) - 1.0)) / (2 * torch.log(torch.tensor(self.k))))
        sigma_T = torch.sqrt((self.c * (self.k ** (2 * self.T) - 1.0)) / (2 * torch.log(torch.tensor(self.k))))
        alpha_bart = alpha_t / (alpha_T + self.eps)
        sigma_bart = torch.sqrt(sigma_T ** 2 - sigma_t ** 2 + self.eps)
        return sigma_t, sigma_T, sigma_bart, alpha_t, alpha_T, alpha_bart

    def std(self, t):
        sigma_t, sigma_T, sigma_bart, alpha_t, alpha_T, alpha_bart = self.sigmas_alphas(t)
        return (alpha_t * sigma_bart * sigma_t) / (sigma_T + self.eps)


def sb_sample(sde: SBVE, model: Callable, y: torch.Tensor, noise: Callable, *, eps=1e-4, sampler_type="ode"):
    """get_sb_sampler, sampling/__init__.py:145-249 (both variants); ``model(x, y, t)`` is the data-prediction network."""
    b4 = lambda v: v[:, None, None, None]
    with torch.no_grad():
        xt = y[:, [0]]
        ts = torch.linspace(sde.T, eps, sde.N + 1)
        sp, _, sbp, ap, _, _ = sde.sigmas_alphas(ts[0] * torch.ones(xt.shape[0]))
        for t in ts[1:]:
            time = t * torch.ones(xt.shape[0])
            st, sT, sbt, at, aT, _ = sde.sigmas_alphas(time)
            est = model(xt, y, time)
            if sampler_type == "sde":
                w_prev = at * st ** 2 / (ap * sp ** 2 + sde.eps)
                tmp = 1 - st ** 2 / (sp ** 2 + sde.eps)
                w_est = at * tmp
                w_z = b4(at * st * torch.sqrt(tmp))
                z = noise(xt)
                if t == ts[-1]:
                    w_z = 0.0
                xt = b4(w_prev) * xt + b4(w_est) * est + w_z * z
            else:
                w_prev = at * st * sbt / (ap * sp * sbp + sde.eps)
                w_est = at / (sT ** 2 + sde.eps) * (sbt ** 2 - sbp * st * sbt / (sp + sde.eps))
                w_y = at / (aT * sT ** 2 + sde.eps) * (st ** 2 - sp * st * sbt / (sbp + sde.eps))
                xt = b4(w_prev) * xt + b4(w_est) * est + b4(w_y) * y
            sp, sbp, ap = st, sbt, at
        return xt, sde.N
