"""Predictors for the generic (Python-loop) sampler; same registry names and update rules as reference
sgmse/sampling/predictors.py:9-76.  The fused HIP sampler implements 'reverse_diffusion' and 'none' natively
(sampling/__init__.py here); these classes serve the remaining combinations and API parity."""
import abc

import numpy as np
import torch

from ..util.registry import Registry

PredictorRegistry = Registry("Predictor")


def _b(v, like):
    return v.reshape(-1, *([1] * (like.dim() - 1))) if torch.is_tensor(v) and v.dim() > 0 else v


class Predictor(abc.ABC):
    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn)     # NB: like the reference (predictors.py:18), probability_flow is not forwarded
        self.score_fn = score_fn
        self.probability_flow = probability_flow

    @abc.abstractmethod
    def update_fn(self, x, y, t, *args):
        ...

    def debug_update_fn(self, x, y, t, *args):
        raise NotImplementedError(f"Debug update function not implemented for predictor {self}.")


@PredictorRegistry.register("euler_maruyama")
class EulerMaruyamaPredictor(Predictor):
    def update_fn(self, x, y, t, *args):
        dt = -1.0 / self.rsde.N
        z = torch.randn_like(x)
        f, g = self.rsde.sde(x, y, t)
        x_mean = x + f * dt
        x = x_mean + _b(g, x) * np.sqrt(-dt) * z
        return x, x_mean


@PredictorRegistry.register("reverse_diffusion")
class ReverseDiffusionPredictor(Predictor):
    def update_fn(self, x, y, t, stepsize):
        f, g = self.rsde.discretize(x, y, t, stepsize)
        z = torch.randn_like(x)
        x_mean = x - f
        x = x_mean + _b(g, x) * z
        return x, x_mean


@PredictorRegistry.register("none")
class NonePredictor(Predictor):
    def __init__(self, *args, **kwargs):
        pass

    def update_fn(self, x, y, t, *args):
        return x, x
