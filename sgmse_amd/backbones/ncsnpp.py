"""NCSN++ score network, host side (reference sgmse/backbones/ncsnpp.py:36-419 and ncsnpp_48k.py).

The ``nn.Module`` below only *holds parameters* under the reference's state_dict names
(``output_layer.{weight,bias}``, ``all_modules.{i}.…`` -- ncsnpp.py:105,253) so that checkpoints load with
``load_state_dict(strict=True)``; ``forward`` hands the tensors to the HIP engine (sgmse_ncsnpp_forward), which runs
the whole network as hand-written gfx950 kernels.  There is no PyTorch implementation of the layers here.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _lib
from .shared import BackboneRegistry


def _res_entries(cin, cout, temb, up, down):
    e = [("GroupNorm_0.weight", (cin,)), ("GroupNorm_0.bias", (cin,)),
         ("Conv_0.weight", (cout, cin, 3, 3)), ("Conv_0.bias", (cout,)),
         ("Dense_0.weight", (cout, temb)), ("Dense_0.bias", (cout,)),
         ("GroupNorm_1.weight", (cout,)), ("GroupNorm_1.bias", (cout,)),
         ("Conv_1.weight", (cout, cout, 3, 3)), ("Conv_1.bias", (cout,))]
    if cin != cout or up or down:
        e += [("Conv_2.weight", (cout, cin, 1, 1)), ("Conv_2.bias", (cout,))]
    return e


def _attn_entries(c):
    e = [("GroupNorm_0.weight", (c,)), ("GroupNorm_0.bias", (c,))]
    for i in range(4):
        e += [(f"NIN_{i}.W", (c, c)), (f"NIN_{i}.b", (c,))]
    return e


def module_manifest(nf: int, ch_mult: Sequence[int], num_res_blocks: int, attn_resolutions: Sequence[int],
                    image_size: int, progressive: str, progressive_input: str) -> List[List[Tuple[str, Tuple[int, ...]]]]:
    """Parameter names/shapes of every entry of ``all_modules`` in constructor order (ncsnpp.py:107-253)."""
    L = len(ch_mult)
    all_res = [image_size // (2 ** i) for i in range(L)]
    temb, channels = nf * 4, 4
    mods: List[List[Tuple[str, Tuple[int, ...]]]] = []
    mods.append([("W", (nf,))])
    mods.append([("weight", (temb, 2 * nf)), ("bias", (temb,))])
    mods.append([("weight", (temb, temb)), ("bias", (temb,))])
    mods.append([("weight", (nf, channels, 3, 3)), ("bias", (nf,))])
    hs_c, in_ch = [nf], nf
    for lvl in range(L):
        for _ in range(num_res_blocks):
            out_ch = nf * ch_mult[lvl]
            mods.append(_res_entries(in_ch, out_ch, temb, False, False))
            in_ch = out_ch
            if all_res[lvl] in attn_resolutions:
                mods.append(_attn_entries(in_ch))
            hs_c.append(in_ch)
        if lvl != L - 1:
            mods.append(_res_entries(in_ch, in_ch, temb, False, True))
            if progressive_input == "input_skip":
                mods.append([("Conv_0.weight", (in_ch, channels, 1, 1)), ("Conv_0.bias", (in_ch,))])
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods.append(_res_entries(in_ch, in_ch, temb, False, False))
    mods.append(_attn_entries(in_ch))
    mods.append(_res_entries(in_ch, in_ch, temb, False, False))
    for lvl in reversed(range(L)):
        for _ in range(num_res_blocks + 1):
            out_ch = nf * ch_mult[lvl]
            mods.append(_res_entries(in_ch + hs_c.pop(), out_ch, temb, False, False))
            in_ch = out_ch
        if all_res[lvl] in attn_resolutions:
            mods.append(_attn_entries(in_ch))
        if progressive == "output_skip":
            mods.append([("weight", (in_ch,)), ("bias", (in_ch,))])
            mods.append([("weight", (channels, in_ch, 3, 3)), ("bias", (channels,))])
        if lvl != 0:
            mods.append(_res_entries(in_ch, in_ch, temb, True, False))
    assert not hs_c
    if progressive != "output_skip":
        mods.append([("weight", (in_ch,)), ("bias", (in_ch,))])
        mods.append([("weight", (channels, in_ch, 3, 3)), ("bias", (channels,))])
    return mods


class _Params(nn.Module):
    """A parameter container whose attribute tree reproduces dotted reference names ('Conv_0.weight')."""

    def __init__(self, entries, gen: torch.Generator, fourier_scale: float):
        super().__init__()
        for name, shape in entries:
            head, _, tail = name.partition(".")
            if tail:
                if not hasattr(self, head):
                    sub = [(n.partition(".")[2], s) for n, s in entries if n.partition(".")[0] == head]
                    setattr(self, head, _Params(sub, gen, fourier_scale))
                continue
            leaf = name
            if leaf == "W" and len(shape) == 1:          # GaussianFourierProjection.W: fixed random frequencies
                p = nn.Parameter(torch.randn(shape, generator=gen) * fourier_scale, requires_grad=False)
            elif leaf in ("bias", "b"):
                p = nn.Parameter(torch.zeros(shape))
            elif len(shape) == 1:                        # GroupNorm weight
                p = nn.Parameter(torch.ones(shape))
            else:                                        # conv / dense / NIN kernels: variance-scaling (fan-in) normal
                fan_in = shape[0] if leaf == "W" else int(torch.tensor(shape[1:]).prod())
                p = nn.Parameter(torch.randn(shape, generator=gen) / max(fan_in, 1) ** 0.5)
            setattr(self, leaf, p)


@BackboneRegistry.register("ncsnpp")
class NCSNpp(nn.Module):
    """NCSN++ (reference ncsnpp.py:36-419).  ``forward(x, time_cond)``: x complex64 [B,2,F,T] (x_t and y stacked on
    dim 1), time_cond float32 [B]  ->  complex64 [B,1,F,T]."""

    VARIANT = "ncsnpp"
    DEFAULTS = dict(attn_resolutions=(16,), progressive="output_skip", progressive_input="input_skip")

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--ch_mult", type=int, nargs="+", default=[1, 1, 2, 2, 2, 2, 2])
        parser.add_argument("--num_res_blocks", type=int, default=2)
        parser.add_argument("--attn_resolutions", type=int, nargs="+", default=[16])
        parser.add_argument("--no-centered", dest="centered", action="store_false", help="The data is not centered [-1, 1]")
        parser.add_argument("--centered", dest="centered", action="store_true", help="The data is centered [-1, 1]")
        parser.set_defaults(centered=True)
        return parser

    def __init__(self, scale_by_sigma=True, nonlinearity="swish", nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                 attn_resolutions=None, resamp_with_conv=True, conditional=True, fir=True, fir_kernel=(1, 3, 3, 1),
                 skip_rescale=True, resblock_type="biggan", progressive=None, progressive_input=None,
                 progressive_combine="sum", init_scale=0.0, fourier_scale=16, image_size=256, embedding_type="fourier",
                 dropout=0.0, centered=True, **unused_kwargs):
        super().__init__()
        attn_resolutions = self.DEFAULTS["attn_resolutions"] if attn_resolutions is None else attn_resolutions
        progressive = self.DEFAULTS["progressive"] if progressive is None else progressive
        progressive_input = self.DEFAULTS["progressive_input"] if progressive_input is None else progressive_input
        # the HIP network implements the configuration every published checkpoint uses; refuse anything else loudly
        fixed = dict(nonlinearity=(nonlinearity, "swish"), conditional=(conditional, True), fir=(fir, True),
                     fir_kernel=(tuple(fir_kernel), (1, 3, 3, 1)), skip_rescale=(skip_rescale, True),
                     resblock_type=(resblock_type, "biggan"), progressive_combine=(progressive_combine, "sum"),
                     embedding_type=(embedding_type, "fourier"), centered=(centered, True))
        for k, (got, want) in fixed.items():
            if got != want:
                raise NotImplementedError(f"{type(self).__name__}: {k}={got!r} is not implemented by the HIP backbone (only {want!r})")
        if dropout not in (0, 0.0):
            raise NotImplementedError("dropout > 0 is a training feature; the HIP backbone is inference-only")
        assert progressive in ("none", "output_skip", "residual") and progressive_input in ("none", "input_skip", "residual")
        if progressive == "residual" or progressive_input == "residual":
            raise NotImplementedError("progressive='residual' is not implemented by the HIP backbone")
        self.nf, self.ch_mult, self.num_res_blocks = int(nf), tuple(int(c) for c in ch_mult), int(num_res_blocks)
        self.attn_resolutions = tuple(int(a) for a in attn_resolutions)
        self.image_size, self.progressive, self.progressive_input = int(image_size), progressive, progressive_input
        self.scale_by_sigma = bool(scale_by_sigma)
        gen = torch.Generator().manual_seed(0)
        self.output_layer = _Params([("weight", (2, 4, 1, 1)), ("bias", (2,))], gen, fourier_scale)
        man = module_manifest(self.nf, self.ch_mult, self.num_res_blocks, self.attn_resolutions, self.image_size,
                              progressive, progressive_input)
        self.all_modules = nn.ModuleList([_Params(e, gen, fourier_scale) for e in man])
        self._ctx = None
        self._dirty = True

    # -- weight hand-off to the engine -----------------------------------------------------------------------
    def _load_from_state_dict(self, *a, **k):
        self._dirty = True
        return super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        self._dirty = True
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def mark_weights_changed(self):
        """Call after modifying parameters in place (e.g. an EMA swap) so the engine re-reads them."""
        self._dirty = True

    def engine(self, device: torch.device) -> "_lib.Context":
        dev = torch.device(device)
        if self._ctx is None or self._ctx.device != (dev if dev.type == "cpu" else torch.device("cuda", dev.index or 0)):
            self._ctx = _lib.Context(dev)
            self._ctx.configure(variant=self.VARIANT, nf=self.nf, ch_mult=self.ch_mult, num_res_blocks=self.num_res_blocks,
                                attn_resolutions=self.attn_resolutions, image_size=self.image_size,
                                progressive=self.progressive, progressive_input=self.progressive_input,
                                scale_by_sigma=self.scale_by_sigma)
            self._dirty = True
        if self._dirty:
            self._ctx.load_weights({k: v for k, v in self.state_dict().items()})
            self._dirty = False
        return self._ctx

    def forward(self, x: torch.Tensor, time_cond: torch.Tensor) -> torch.Tensor:
        ctx = self.engine(x.device)
        return ctx.forward(x, time_cond.to(torch.float32))


@BackboneRegistry.register("ncsnpp_48k")
class NCSNpp_48k(NCSNpp):
    """48 kHz variant (reference ncsnpp_48k.py): no attention except the bottleneck, no pyramids, and
    ``output_layer`` applied before the division by t (ncsnpp_48k.py:59,66-67,414-421)."""

    VARIANT = "ncsnpp_48k"
    DEFAULTS = dict(attn_resolutions=(), progressive="none", progressive_input="none")


@BackboneRegistry.register("ncsnpp_v2")
class NCSNpp_v2(NCSNpp):
    """Backbone of the ICASSP-2025 / Schroedinger-bridge checkpoints (reference ncsnpp_v2.py:36-395): the ncsnpp graph with
    ``forward(x, y, t)`` (x_t and y passed separately, ncsnpp_v2.py:241-247) and no division by t at the output
    (ncsnpp_v2.py:388-394).  All output scaling lives in ScoreModel.forward (model.py:284-304)."""

    VARIANT = "ncsnpp"

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--nf", type=int, default=128)
        parser.add_argument("--ch_mult", type=int, nargs="+", default=[1, 1, 2, 2, 2, 2, 2])
        parser.add_argument("--num_res_blocks", type=int, default=2)
        parser.add_argument("--attn_resolutions", type=int, nargs="+", default=[16])
        return parser

    def __init__(self, **kwargs):
        kwargs.pop("scale_by_sigma", None)
        kwargs.pop("conditional", None)
        kwargs.pop("centered", None)
        super().__init__(scale_by_sigma=False, **kwargs)

    def forward(self, x: torch.Tensor, y: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        return super().forward(torch.cat([x, y], dim=1), t)
