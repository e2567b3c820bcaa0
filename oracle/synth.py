"""Deterministic synthetic weights and inputs (TEST INFRASTRUCTURE ONLY).

No checkpoint is reachable offline, so parity and timing use
synthetic parameters.  They are drawn from a dedicated CPU ``torch.Generator`` in
``param_shapes`` order, so the build container (where the real reference is run
to make golden vectors) and the GPU box (same image, same torch) get bit-identical
tensors without shipping 262 MB of weights.

Unlike the reference constructor, which initialises 56 conv weights at scale
1e-10 (layers.py:90,122), every weight here is O(1/sqrt(fan_in))
so that deep layers contribute to the output and parity is not vacuous.
"""
from __future__ import annotations

from typing import Dict

import torch

from .ncsnpp_oracle import NetCfg, param_shapes


def synth_params(cfg: NetCfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in param_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if name == "all_modules.0.W":                       # GaussianFourierProjection.W, layerspp.py:37
            v = torch.randn(shp, generator=g) * cfg.fourier_scale
        elif "GroupNorm" in name or (leaf in ("weight", "bias") and len(shp) == 1 and _is_gn(name, cfg)):
            v = (1.0 + 0.1 * torch.randn(shp, generator=g)) if leaf == "weight" else 0.1 * torch.randn(shp, generator=g)
        elif leaf in ("bias", "b"):
            v = 0.1 * torch.randn(shp, generator=g)
        elif leaf == "W":                                   # NIN.W [C_in, C_out], layers.py:549
            v = torch.randn(shp, generator=g) / (shp[0] ** 0.5)
        else:                                               # conv / linear weight [C_out, C_in, ...]
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) / (fan_in ** 0.5)
        out[name] = v.to(torch.float32).contiguous()
    return out


_GN_CACHE: Dict[str, set] = {}


def _is_gn(name: str, cfg: NetCfg) -> bool:
    """Stand-alone nn.GroupNorm entries of all_modules (ncsnpp.py:218,230,249)."""
    from .ncsnpp_oracle import build_layout
    # keyed by the configuration's VALUE (round 5: it was keyed by id(cfg) -- ids are reused once an object is collected, so a later,
    # different configuration could inherit another one's GroupNorm set and draw its parameters from the wrong distributions: a rare,
    # order-dependent failure of whichever oracle test came next)
    key = repr(cfg)
    if key not in _GN_CACHE:
        _GN_CACHE[key] = {f"all_modules.{m.idx}" for m in build_layout(cfg) if m.kind == "gn"}
    return name.rsplit(".", 1)[0] in _GN_CACHE[key]


def synth_waveform(length: int = 64000, seed: int = 0, batch: int = 1) -> torch.Tensor:
    """SURVEY 8-d synthetic input: y ~ N(0,1) fp32 [batch, length]."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, length, generator=g)


def synth_spec(B: int, F_: int, T: int, seed: int = 0, scale: float = 0.25) -> torch.Tensor:
    """A complex64 [B,1,F,T] 'noisy spectrogram' with the magnitude statistics the
    spec_fwd transform yields (abs mean ~0.24)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 1, F_, T, dtype=torch.complex64, generator=g) * scale)
