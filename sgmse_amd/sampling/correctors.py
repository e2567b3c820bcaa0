"""Correctors for the generic (Python-loop) sampler; registry names and update rules of reference
sgmse/sampling/correctors.py:9-94.  'ald' and 'none' also exist as fused HIP kernels (native path)."""
import abc

import torch

from ..util.registry import Registry

CorrectorRegistry = Registry("Corrector")


def _b(v, like):
    return v.reshape(-1, *([1] * (like.dim() - 1)))


class Corrector(abc.ABC):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.rsde = sde.reverse(score_fn)
        self.score_fn = score_fn
        self.snr = snr
        self.n_steps = n_steps

    @abc.abstractmethod
    def update_fn(self, x, y, t, *args):
        ...


@CorrectorRegistry.register(name="langevin")
class LangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        self.score_fn = score_fn
        self.n_steps = n_steps
        self.snr = snr

    def update_fn(self, x, y, t, *args):
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, y, t)
            noise = torch.randn_like(x)
            grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
            noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
            step_size = ((self.snr * noise_norm / grad_norm) ** 2 * 2).unsqueeze(0)
            x_mean = x + _b(step_size, x) * grad
            x = x_mean + noise * _b(torch.sqrt(step_size * 2), x)
        return x, x_mean


@CorrectorRegistry.register(name="ald")
class AnnealedLangevinDynamics(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        self.sde = sde
        self.score_fn = score_fn
        self.snr = snr
        self.n_steps = n_steps

    def update_fn(self, x, y, t, *args):
        x_mean = x
        std = self.sde.marginal_prob(x, y, t)[1]
        for _ in range(self.n_steps):
            grad = self.score_fn(x, y, t)
            noise = torch.randn_like(x)
            step_size = (self.snr * std) ** 2 * 2
            x_mean = x + _b(step_size, x) * grad
            x = x_mean + noise * _b(torch.sqrt(step_size * 2), x)
        return x, x_mean


@CorrectorRegistry.register(name="none")
class NoneCorrector(Corrector):
    def __init__(self, *args, **kwargs):
        self.snr = 0
        self.n_steps = 0

    def update_fn(self, x, y, t, *args):
        return x, x
