"""Predictor half of the generic (Python-loop) predictor-corrector sampler.

Registry names and the ``update_fn(x, y, t, stepsize) -> (x, x_mean)`` contract follow reference
sgmse/sampling/predictors.py (:9 registry, :41-52 'euler_maruyama', :56-65 'reverse_diffusion', :69-76 'none').
'reverse_diffusion' and 'none' run fused inside ``sgmse_pc_sample``; these classes serve ``force_python_loop=True``,
non-HIP score functions and API parity.

A predictor asks the reverse-time SDE for a mean displacement and a noise amplitude and applies them:

    x_mean = x + shift          x = x_mean + amp * z,   z ~ CN(0, I)

'reverse_diffusion' takes both from the step's discretisation (shift = -rev_f, amp = rev_G), 'euler_maruyama' from the
continuous coefficients with the fixed step -1/N.  NB 'euler_maruyama' cannot run inside the PC loop -- neither here
nor in the reference: the loop passes ``stepsize`` on to ``update_fn``, which forwards it into ``RSDE.sde`` and from
there into ``OUVESDE.sde(x, y, t)`` (predictors.py:49 -> sdes.py:114-120): TypeError.  The behaviour is kept (and
pinned by a test) rather than silently repaired; call ``update_fn(x, y, t)`` directly to use the class.
"""
import abc

import numpy as np
import torch

from ..util.registry import Registry

PredictorRegistry = Registry("Predictor")


def _amp(g, like):
    """Noise amplitude ([B] tensor or scalar) -> broadcastable against ``like`` [B, ...]."""
    if torch.is_tensor(g) and g.dim() > 0:
        return g.reshape(-1, *([1] * (like.dim() - 1)))
    return g


class Predictor(abc.ABC):
    """What ``get_pc_sampler`` builds: ``Predictor(sde, score_fn, probability_flow=...)``.  Like the reference
    (predictors.py:18) the flag is stored but NOT handed to ``sde.reverse``: the PC sampler always integrates the
    reverse SDE, the probability-flow ODE has its own sampler."""

    def __init__(self, sde, score_fn, probability_flow=False):
        self.sde, self.score_fn = sde, score_fn
        self.rsde = sde.reverse(score_fn)
        self.probability_flow = probability_flow

    @abc.abstractmethod
    def update_fn(self, x, y, t, *args):
        """One predictor step at time ``t`` ([B]); returns the noisy iterate and its noise-free mean."""

    def debug_update_fn(self, x, y, t, *args):
        raise NotImplementedError(f"Debug update function not implemented for predictor {self}.")

    @staticmethod
    def _apply(x, shift, amp):
        mean = x + shift
        return mean + _amp(amp, x) * torch.randn_like(x), mean


@PredictorRegistry.register("euler_maruyama")
class EulerMaruyamaPredictor(Predictor):
    def update_fn(self, x, y, t, *args):
        h = -1.0 / self.rsde.N
        z = torch.randn_like(x)                      # drawn before the score evaluation, like the reference
        drift, g = self.rsde.sde(x, y, t, *args)
        mean = x + drift * h
        return mean + _amp(g, x) * np.sqrt(-h) * z, mean


@PredictorRegistry.register("reverse_diffusion")
class ReverseDiffusionPredictor(Predictor):
    def update_fn(self, x, y, t, stepsize):
        rev_f, rev_g = self.rsde.discretize(x, y, t, stepsize)
        return self._apply(x, -rev_f, rev_g)


@PredictorRegistry.register("none")
class NonePredictor(Predictor):
    """Identity."""

    def __init__(self, *args, **kwargs):
        pass

    def update_fn(self, x, y, t, *args):
        return x, x
