"""Corrector half of the generic (Python-loop) predictor-corrector sampler.

Registry names and the ``update_fn(x, y, t) -> (x, x_mean)`` contract follow reference sgmse/sampling/correctors.py
(:9 registry, :37-56 'langevin', :60-81 'ald', :85-94 'none').  All three are also fused HIP kernels inside
``sgmse_pc_sample`` (the default path); these classes serve ``force_python_loop=True``, non-HIP score functions and
API parity.

Both Langevin variants are the same iteration

    x_mean = x + eps * score(x, y, t)          x = x_mean + sqrt(2 eps) * z,   z ~ CN(0, I)

and differ only in the step size eps, so the loop lives once in ``_LangevinLoop`` and a variant supplies ``_eps``:
'ald' anneals with the SDE's marginal standard deviation (one eps per utterance), 'langevin' balances the batch-mean
norms of noise and score (one eps for the whole batch).
"""
import abc

import torch

from ..util.registry import Registry

CorrectorRegistry = Registry("Corrector")


def _per_utterance(v, like):
    """[B] (or [1]) coefficients -> broadcastable against ``like`` [B, ...]."""
    return v.reshape(-1, *([1] * (like.dim() - 1)))


class Corrector(abc.ABC):
    """What ``get_pc_sampler`` builds: ``Corrector(sde, score_fn, snr=..., n_steps=...)``; ``n_steps`` counts score
    evaluations per sampler step (it enters the sampler's nfe)."""

    def __init__(self, sde, score_fn, snr, n_steps):
        self.sde, self.score_fn = sde, score_fn
        self.rsde = sde.reverse(score_fn)
        self.snr, self.n_steps = snr, n_steps

    @abc.abstractmethod
    def update_fn(self, x, y, t, *args):
        """One corrector pass at time ``t`` ([B]); returns the noisy iterate and its noise-free mean."""


class _LangevinLoop(Corrector):
    @abc.abstractmethod
    def _eps(self, x, y, t, score, z):
        """Step size as a [B] or [1] tensor."""

    def update_fn(self, x, y, t, *args):
        mean = x
        for _ in range(self.n_steps):
            score = self.score_fn(x, y, t)
            z = torch.randn_like(x)
            eps = self._eps(x, y, t, score, z)
            mean = x + _per_utterance(eps, x) * score
            x = mean + z * _per_utterance(torch.sqrt(eps * 2), x)
        return x, mean


@CorrectorRegistry.register(name="langevin")
class LangevinCorrector(_LangevinLoop):
    """eps = 2 (snr * mean_b ||z_b|| / mean_b ||score_b||)^2: the only place of the path where utterances of a batch
    interact (correctors.py:50-52)."""

    def _eps(self, x, y, t, score, z):
        flat_norm = lambda v: torch.norm(v.reshape(v.shape[0], -1), dim=-1).mean()
        return ((self.snr * flat_norm(z) / flat_norm(score)) ** 2 * 2).unsqueeze(0)


@CorrectorRegistry.register(name="ald")
class AnnealedLangevinDynamics(_LangevinLoop):
    """eps = 2 (snr * std(t))^2 with std(t) the marginal standard deviation of the forward SDE (correctors.py:72-77)."""

    def _eps(self, x, y, t, score, z):
        return (self.snr * self.sde.marginal_prob(x, y, t)[1]) ** 2 * 2


@CorrectorRegistry.register(name="none")
class NoneCorrector(Corrector):
    """Identity; contributes no score evaluations (n_steps = 0)."""

    def __init__(self, *args, **kwargs):
        self.snr, self.n_steps = 0, 0

    def update_fn(self, x, y, t, *args):
        return x, x
