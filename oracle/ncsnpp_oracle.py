"""Oracle restatement of the NCSN++ score network (TEST INFRASTRUCTURE ONLY).

Functional PyTorch-fp32 CPU restatement of
  sgmse/backbones/ncsnpp.py:50-253 (module list)  and  :256-419 (forward),
  sgmse/backbones/ncsnpp_48k.py (same, defaults/ordering deltas at :59,66-67,414-421),
  sgmse/backbones/ncsnpp_utils/layerspp.py:32-41,44-59,62-91,212-274,
  sgmse/backbones/ncsnpp_utils/layers.py:546-555 (NIN),
  sgmse/backbones/ncsnpp_utils/up_or_down_sampling.py:195-257 + op/upfirdn2d.py:162-203.

Parameters live in a flat dict keyed by the reference ``state_dict`` names
(``output_layer.weight``, ``all_modules.{i}.…``).  The module list is described
by ``build_layout`` (a list of small records) instead of ``nn.Module`` objects.
Pinned against the reference by oracle/make_golden.py -> tests/golden/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class NetCfg:
    """Constructor arguments of NCSNpp (ncsnpp.py:50-72) that change the graph."""
    variant: str = "ncsnpp"                 # "ncsnpp" | "ncsnpp_48k"
    nf: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 2, 2, 2)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = (16,)
    image_size: int = 256
    progressive: str = "output_skip"        # 'none' | 'output_skip'
    progressive_input: str = "input_skip"   # 'none' | 'input_skip'
    scale_by_sigma: bool = True
    fourier_scale: float = 16.0

    @staticmethod
    def for_variant(variant: str, **kw) -> "NetCfg":
        if variant == "ncsnpp":
            return NetCfg(variant="ncsnpp", **kw)
        if variant == "ncsnpp_v2":       # ncsnpp_v2.py:50-66: the ncsnpp graph; forward(x, y, t) without the final /t
            return NetCfg(variant="ncsnpp_v2", **kw)
        if variant == "ncsnpp_48k":      # ncsnpp_48k.py:59,66-67
            base = dict(attn_resolutions=(), progressive="none", progressive_input="none")
            base.update(kw)
            return NetCfg(variant="ncsnpp_48k", **base)
        raise ValueError(variant)


@dataclass
class Mod:
    kind: str          # fourier | linear | conv3 | res | attn | combine | gn
    idx: int           # index into all_modules
    cin: int = 0
    cout: int = 0
    up: bool = False
    down: bool = False
    shapes: Dict[str, Tuple[int, ...]] = field(default_factory=dict)


def _res_shapes(cin, cout, temb_dim, up, down):
    s = {
        "GroupNorm_0.weight": (cin,), "GroupNorm_0.bias": (cin,),
        "Conv_0.weight": (cout, cin, 3, 3), "Conv_0.bias": (cout,),
        "Dense_0.weight": (cout, temb_dim), "Dense_0.bias": (cout,),
        "GroupNorm_1.weight": (cout,), "GroupNorm_1.bias": (cout,),
        "Conv_1.weight": (cout, cout, 3, 3), "Conv_1.bias": (cout,),
    }
    if cin != cout or up or down:          # layerspp.py:234-235
        s["Conv_2.weight"] = (cout, cin, 1, 1)
        s["Conv_2.bias"] = (cout,)
    return s


def _attn_shapes(c):
    s = {"GroupNorm_0.weight": (c,), "GroupNorm_0.bias": (c,)}
    for i in range(4):
        s[f"NIN_{i}.W"] = (c, c)
        s[f"NIN_{i}.b"] = (c,)
    return s


def build_layout(cfg: NetCfg) -> List[Mod]:
    """The ``all_modules`` list, in constructor order (ncsnpp.py:107-253)."""
    nf, nres = cfg.nf, len(cfg.ch_mult)
    all_res = [cfg.image_size // (2 ** i) for i in range(nres)]      # :84
    temb_dim = nf * 4
    mods: List[Mod] = []

    def add(kind, **kw):
        m = Mod(kind=kind, idx=len(mods), **kw)
        mods.append(m)
        return m

    add("fourier", shapes={"W": (nf,)})                                # :111-113
    add("linear", cin=2 * nf, cout=temb_dim,
        shapes={"weight": (temb_dim, 2 * nf), "bias": (temb_dim,)})   # :121
    add("linear", cin=temb_dim, cout=temb_dim,
        shapes={"weight": (temb_dim, temb_dim), "bias": (temb_dim,)})  # :124
    channels = 4
    add("conv3", cin=channels, cout=nf,
        shapes={"weight": (nf, channels, 3, 3), "bias": (nf,)})       # :167
    hs_c = [nf]
    in_ch = nf
    for lvl in range(nres):                                            # :171-197
        for _ in range(cfg.num_res_blocks):
            out_ch = nf * cfg.ch_mult[lvl]
            add("res", cin=in_ch, cout=out_ch, shapes=_res_shapes(in_ch, out_ch, temb_dim, False, False))
            in_ch = out_ch
            if all_res[lvl] in cfg.attn_resolutions:
                add("attn", cin=in_ch, cout=in_ch, shapes=_attn_shapes(in_ch))
            hs_c.append(in_ch)
        if lvl != nres - 1:
            add("res", cin=in_ch, cout=in_ch, down=True, shapes=_res_shapes(in_ch, in_ch, temb_dim, False, True))
            if cfg.progressive_input == "input_skip":
                add("combine", cin=channels, cout=in_ch,
                    shapes={"Conv_0.weight": (in_ch, channels, 1, 1), "Conv_0.bias": (in_ch,)})
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    add("res", cin=in_ch, cout=in_ch, shapes=_res_shapes(in_ch, in_ch, temb_dim, False, False))   # :200-202
    add("attn", cin=in_ch, cout=in_ch, shapes=_attn_shapes(in_ch))
    add("res", cin=in_ch, cout=in_ch, shapes=_res_shapes(in_ch, in_ch, temb_dim, False, False))
    for lvl in reversed(range(nres)):                                  # :206-244
        for _ in range(cfg.num_res_blocks + 1):
            out_ch = nf * cfg.ch_mult[lvl]
            cin = in_ch + hs_c.pop()
            add("res", cin=cin, cout=out_ch, shapes=_res_shapes(cin, out_ch, temb_dim, False, False))
            in_ch = out_ch
        if all_res[lvl] in cfg.attn_resolutions:
            add("attn", cin=in_ch, cout=in_ch, shapes=_attn_shapes(in_ch))
        if cfg.progressive == "output_skip":
            add("gn", cin=in_ch, cout=in_ch, shapes={"weight": (in_ch,), "bias": (in_ch,)})
            add("conv3", cin=in_ch, cout=channels,
                shapes={"weight": (channels, in_ch, 3, 3), "bias": (channels,)})
        if lvl != 0:
            add("res", cin=in_ch, cout=in_ch, up=True, shapes=_res_shapes(in_ch, in_ch, temb_dim, True, False))
    assert not hs_c                                                    # :246
    if cfg.progressive != "output_skip":                               # :248-251
        add("gn", cin=in_ch, cout=in_ch, shapes={"weight": (in_ch,), "bias": (in_ch,)})
        add("conv3", cin=in_ch, cout=channels,
            shapes={"weight": (channels, in_ch, 3, 3), "bias": (channels,)})
    return mods


def param_shapes(cfg: NetCfg) -> Dict[str, Tuple[int, ...]]:
    """Reference state_dict key -> shape, in ``named_parameters()`` order
    (``output_layer`` first: ncsnpp.py:105 registers it before ``all_modules`` :253)."""
    out = {"output_layer.weight": (2, 4, 1, 1), "output_layer.bias": (2,)}
    for m in build_layout(cfg):
        for k, shp in m.shapes.items():
            out[f"all_modules.{m.idx}.{k}"] = shp
    return out


# ----------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------

def group_norm(x, w, b):
    """nn.GroupNorm(min(C//4, 32), C, eps=1e-6) (layerspp.py:219)."""
    c = x.shape[1]
    return F.group_norm(x, min(c // 4, 32), w, b, eps=1e-6)


def silu(x):
    return x * torch.sigmoid(x)            # nn.SiLU, layers.py:38-39


def nin(x, W, b):
    """layers.py:546-555: per-pixel x @ W + b with W [C_in, C_out]."""
    return torch.einsum("bchw,cd->bdhw", x, W) + b[None, :, None, None]


FIR_TAPS = (1.0, 3.0, 3.0, 1.0)


def fir_down2(x):
    """downsample_2d(x, (1,3,3,1), factor=2) (up_or_down_sampling.py:227-257) =
    upfirdn2d(down=2, pad=(1,1)) with k = outer(taps)/64 (upfirdn2d.py:162-203).
    Closed form: out[i,j] = sum_ab k[a]k[b] x[2i+a-1, 2j+b-1], zero outside."""
    k1 = torch.tensor(FIR_TAPS, dtype=x.dtype, device=x.device) / 8.0
    k2 = torch.outer(k1, k1)
    c = x.shape[1]
    w = k2.flip(0, 1)[None, None].expand(c, 1, 4, 4).contiguous()
    xp = F.pad(x, (1, 1, 1, 1))
    return F.conv2d(xp, w, stride=2, groups=c)


def fir_up2(x):
    """upsample_2d(x, (1,3,3,1), factor=2) (up_or_down_sampling.py:195-224) =
    zero-insert, pad (2,1), correlate with flipped k*4."""
    k1 = torch.tensor(FIR_TAPS, dtype=x.dtype, device=x.device) / 8.0
    k2 = torch.outer(k1, k1) * 4.0
    b, c, h, w = x.shape
    z = torch.zeros(b, c, 2 * h, 2 * w, dtype=x.dtype, device=x.device)
    z[:, :, ::2, ::2] = x
    zp = F.pad(z, (2, 1, 2, 1))
    wk = k2.flip(0, 1)[None, None].expand(c, 1, 4, 4).contiguous()
    return F.conv2d(zp, wk, groups=c)


def upfirdn2d_ref(x, kernel, up=1, down=1, pad=(0, 0)):
    """Generic restatement of op/upfirdn2d.py:162-203 for [N,C,H,W] input and a
    2-D kernel; same pad on both axes as the wrapper :148-159."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    z = torch.zeros(n, c, h * up, w * up, dtype=x.dtype, device=x.device)
    z[:, :, ::up, ::up] = x
    p0, p1 = pad
    z = F.pad(z, (max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)))
    z = z[:, :, max(-p0, 0): z.shape[2] - max(-p1, 0), max(-p0, 0): z.shape[3] - max(-p1, 0)]
    wk = kernel.flip(0, 1)[None, None].to(x.dtype).expand(c, 1, kh, kw).contiguous()
    y = F.conv2d(z, wk, groups=c)
    return y[:, :, ::down, ::down]


def res_block(P, pre, m: Mod, x, temb_act):
    """ResnetBlockBigGANpp.forward, layerspp.py:242-274 (dropout p=0 -> identity)."""
    g = lambda k: P[f"{pre}.{k}"]
    h = silu(group_norm(x, g("GroupNorm_0.weight"), g("GroupNorm_0.bias")))
    if m.up:
        h, x = fir_up2(h), fir_up2(x)
    elif m.down:
        h, x = fir_down2(h), fir_down2(x)
    h = F.conv2d(h, g("Conv_0.weight"), g("Conv_0.bias"), padding=1)
    h = h + F.linear(temb_act, g("Dense_0.weight"), g("Dense_0.bias"))[:, :, None, None]
    h = silu(group_norm(h, g("GroupNorm_1.weight"), g("GroupNorm_1.bias")))
    h = F.conv2d(h, g("Conv_1.weight"), g("Conv_1.bias"), padding=1)
    if f"{pre}.Conv_2.weight" in P:
        x = F.conv2d(x, g("Conv_2.weight"), g("Conv_2.bias"))
    return (x + h) / math.sqrt(2.0)


def attn_block(P, pre, x):
    """AttnBlockpp.forward, layerspp.py:75-91 (skip_rescale=True)."""
    g = lambda k: P[f"{pre}.{k}"]
    b, c, hh, ww = x.shape
    h = group_norm(x, g("GroupNorm_0.weight"), g("GroupNorm_0.bias"))
    q = nin(h, g("NIN_0.W"), g("NIN_0.b")).reshape(b, c, hh * ww)
    k = nin(h, g("NIN_1.W"), g("NIN_1.b")).reshape(b, c, hh * ww)
    v = nin(h, g("NIN_2.W"), g("NIN_2.b")).reshape(b, c, hh * ww)
    w = torch.einsum("bcs,bcr->bsr", q, k) * (int(c) ** (-0.5))
    w = torch.softmax(w, dim=-1)
    o = torch.einsum("bsr,bcr->bcs", w, v).reshape(b, c, hh, ww)
    o = nin(o, g("NIN_3.W"), g("NIN_3.b"))
    return (x + o) / math.sqrt(2.0)


def time_embedding(P, t):
    """ncsnpp.py:265-284: Fourier features of log(t), Linear, SiLU, Linear.
    Returns temb (pre-activation); consumers apply SiLU (layerspp.py:263)."""
    W = P["all_modules.0.W"]
    proj = torch.log(t)[:, None] * W[None, :] * 2 * math.pi           # layerspp.py:40
    emb = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
    temb = F.linear(emb, P["all_modules.1.weight"], P["all_modules.1.bias"])
    temb = F.linear(silu(temb), P["all_modules.2.weight"], P["all_modules.2.bias"])
    return temb


def ncsnpp_forward(P: Dict[str, torch.Tensor], cfg: NetCfg, x: torch.Tensor, t: torch.Tensor,
                   taps: Optional[dict] = None) -> torch.Tensor:
    """NCSNpp.forward (ncsnpp.py:256-419) / NCSNpp_48k.forward.

    x: complex64 [B,2,F,T] (channel 0 = x_t, channel 1 = y); t: float32 [B].
    Returns complex64 [B,1,F,T].  ``taps`` (optional dict) receives named
    intermediate tensors for per-stage debugging of the HIP path."""
    mods = build_layout(cfg)
    it = iter(mods)
    nres = len(cfg.ch_mult)
    next(it); next(it); next(it)                                       # fourier + 2 linear
    xr = torch.cat([x[:, [0]].real, x[:, [0]].imag, x[:, [1]].real, x[:, [1]].imag], dim=1)  # :262-263
    temb_act = silu(time_embedding(P, t))
    pre = lambda m: f"all_modules.{m.idx}"

    def tap(name, val):
        if taps is not None:
            taps[name] = val

    m = next(it)
    pyr_in = xr if cfg.progressive_input != "none" else None
    hs = [F.conv2d(xr, P[pre(m) + ".weight"], P[pre(m) + ".bias"], padding=1)]   # :298
    tap("conv_in", hs[0])
    for lvl in range(nres):                                            # :302-335
        for _ in range(cfg.num_res_blocks):
            m = next(it)
            h = res_block(P, pre(m), m, hs[-1], temb_act)
            tap(f"m{m.idx}", h)
            if h.shape[-2] in cfg.attn_resolutions:                    # :308
                m = next(it)
                h = attn_block(P, pre(m), h)
                tap(f"m{m.idx}", h)
            hs.append(h)
        if lvl != nres - 1:
            m = next(it)
            h = res_block(P, pre(m), m, hs[-1], temb_act)
            tap(f"m{m.idx}", h)
            if cfg.progressive_input == "input_skip":                  # :322-325
                m = next(it)
                pyr_in = fir_down2(pyr_in)
                h = F.conv2d(pyr_in, P[pre(m) + ".Conv_0.weight"], P[pre(m) + ".Conv_0.bias"]) + h
                tap(f"m{m.idx}", h)
            hs.append(h)
    h = hs[-1]
    m = next(it); h = res_block(P, pre(m), m, h, temb_act); tap(f"m{m.idx}", h)   # :338
    m = next(it); h = attn_block(P, pre(m), h); tap(f"m{m.idx}", h)               # :340
    m = next(it); h = res_block(P, pre(m), m, h, temb_act); tap(f"m{m.idx}", h)   # :342
    pyramid = None
    for lvl in reversed(range(nres)):                                  # :348-398
        for _ in range(cfg.num_res_blocks + 1):
            m = next(it)
            h = res_block(P, pre(m), m, torch.cat([h, hs.pop()], dim=1), temb_act)
            tap(f"m{m.idx}", h)
        if h.shape[-2] in cfg.attn_resolutions:                        # :354
            m = next(it)
            h = attn_block(P, pre(m), h)
            tap(f"m{m.idx}", h)
        if cfg.progressive == "output_skip":                           # :358-379
            mg = next(it); mc = next(it)
            ph = silu(group_norm(h, P[pre(mg) + ".weight"], P[pre(mg) + ".bias"]))
            ph = F.conv2d(ph, P[pre(mc) + ".weight"], P[pre(mc) + ".bias"], padding=1)
            pyramid = ph if pyramid is None else fir_up2(pyramid) + ph
            tap(f"m{mc.idx}", pyramid)
        if lvl != 0:
            m = next(it)
            h = res_block(P, pre(m), m, h, temb_act)
            tap(f"m{m.idx}", h)
    assert not hs                                                      # :400
    if cfg.progressive == "output_skip":
        h = pyramid
    else:                                                              # :404-408
        mg = next(it); mc = next(it)
        h = silu(group_norm(h, P[pre(mg) + ".weight"], P[pre(mg) + ".bias"]))
        h = F.conv2d(h, P[pre(mc) + ".weight"], P[pre(mc) + ".bias"], padding=1)
    assert next(it, None) is None                                      # :410
    ow, ob = P["output_layer.weight"], P["output_layer.bias"]
    if cfg.variant == "ncsnpp_48k":                                    # ncsnpp_48k.py:414-421
        h = F.conv2d(h, ow, ob)
        if cfg.scale_by_sigma:
            h = h / t[:, None, None, None]
    elif cfg.variant == "ncsnpp_v2":                                   # ncsnpp_v2.py:388-394: no division by t
        h = F.conv2d(h, ow, ob)
    else:                                                              # ncsnpp.py:411-416
        if cfg.scale_by_sigma:
            h = h / t[:, None, None, None]
        h = F.conv2d(h, ow, ob)
    h = h.permute(0, 2, 3, 1).contiguous()
    return torch.view_as_complex(h)[:, None]


def score_fn(P, cfg, x_t, y, t):
    """ScoreModel.forward, old-code branch (model.py:307-310)."""
    return -ncsnpp_forward(P, cfg, torch.cat([x_t, y], dim=1), t)


def score_fn_v2(P, cfg, sde, x_t, y, t, loss_type="score_matching", network_scaling=None, c_in="1", c_out="1", c_skip="0",
                sigma_data=0.1):
    """ScoreModel.forward, new-code branch for backbone 'ncsnpp_v2' (model.py:284-304 with _c_in/_c_out/_c_skip :312-341).
    ``sde`` provides std(t) (OUVESDE._std)."""
    b4 = lambda v: v[:, None, None, None]
    std = sde.std(t)
    cin = 1.0 if c_in == "1" else b4(1.0 / torch.sqrt(std ** 2 + sigma_data ** 2))
    if c_out == "1":
        cout = 1.0
    elif c_out == "sigma":
        cout = b4(std)
    elif c_out == "1/sigma":
        cout = 1.0 / b4(std)
    else:                                                              # "edm"
        cout = b4((std * sigma_data) / torch.sqrt(sigma_data ** 2 + std ** 2))
    cskip = 0.0 if c_skip == "0" else b4(sigma_data ** 2 / (std ** 2 + sigma_data ** 2))
    Fo = ncsnpp_forward(P, cfg, torch.cat([cin * x_t, cin * y], dim=1), t)
    if network_scaling == "1/sigma":
        Fo = Fo / b4(std)
    elif network_scaling == "1/t":
        Fo = Fo / b4(t)
    if loss_type == "score_matching":
        return cskip * x_t + cout * Fo
    if loss_type == "denoiser":
        return (Fo - x_t) / b4(std).pow(2)
    if loss_type == "data_prediction":
        return cskip * x_t + cout * Fo
    raise ValueError(loss_type)
