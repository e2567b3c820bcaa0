"""Parity checks shared by the CPU (emulator) and GPU (HIP) test modules: every function runs one piece of the
sgmse_amd path on ``dev`` and compares it with the oracle (oracle/, CPU torch fp32) or the golden fixtures produced by
the reference itself (tests/golden/, oracle/make_golden.py)."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ROOT, rel_l2
from oracle import ncsnpp_oracle as NO
from oracle import sde_oracle as SO
from oracle import stft_oracle as FO
from oracle import synth

OP_TOL = 1e-5          # per-op relative L2 (SURVEY 8-d parity gate)
NET_TOL = 2e-5         # one network evaluation vs the reference's own output
SAMPLER_TOL = 1e-4     # sampler output vs the reference (contractive chain)
WAVE_TOL = 1e-3        # enhanced waveform, the north-star tolerance


def gen(seed=0):
    return torch.Generator().manual_seed(seed)


def R(g, *s):
    return torch.randn(*s, generator=g)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def check_conv(dev, B, Ci, Co, H, W, ks, direct=False, dual=0, xform=False):
    from sgmse_amd import ops
    g = gen(B * 1000 + Ci + Co + H + W)
    x = R(g, B, Ci, H, W); w = R(g, Co, Ci, ks, ks) / math.sqrt(Ci * ks * ks); b = R(g, Co); r = R(g, B, Co, H, W)
    sc = sh = None
    xin = x
    if xform:
        sc, sh = R(g, B, Ci), R(g, B, Ci)
        xin = x * sc[:, :, None, None] + sh[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    ref = (F.conv2d(xin, w, b, padding=ks // 2) + r) / math.sqrt(2.0)
    x1, x2 = (x[:, :Ci - dual].contiguous(), x[:, Ci - dual:].contiguous()) if dual else (x, None)
    mv = lambda t: None if t is None else t.to(dev)
    out = ops.conv2d(mv(x1), mv(w), mv(b), residual=mv(r), out_scale=1 / math.sqrt(2.0), x2=mv(x2), in_scale=mv(sc), in_shift=mv(sh),
                     in_act=xform, force_direct=direct)
    assert rel_l2(out.cpu(), ref) < OP_TOL, (B, Ci, Co, H, W, ks, direct, dual, xform)


def check_conv_b3(dev, B, Ci, Co, H, W, dual=0, xform=False, split="bf16x3", slack=2.0, ks=3, xmul=1.0):
    """The bf16x3 3x3 kernel: fp32 operands split exactly into three bf16 terms, six partial products on the bf16 MFMA
    pipe, fp32 accumulate.  Gate: the same per-op 1e-5 as the fp32 kernels against the fp32 oracle, AND an error against
    an fp64 convolution that is no worse than 2x the fp32 kernel's own (i.e. fp32 accuracy, not bf16 accuracy)."""
    from sgmse_amd import ops
    g = gen(B * 1000 + Ci + Co + H + W)
    x = R(g, B, Ci, H, W) * xmul; w = R(g, Co, Ci, ks, ks) / math.sqrt(Ci * ks * ks); b = R(g, Co) * xmul; r = R(g, B, Co, H, W) * xmul
    if xmul != 1.0:
        x[0] *= 0.01          # utterances of one batch with very different ranges: the scale is per utterance
    sc = sh = None
    xin = x
    if xform:
        sc, sh = R(g, B, Ci), R(g, B, Ci)
        xin = x * sc[:, :, None, None] + sh[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    ref32 = (F.conv2d(xin, w, b, padding=ks // 2) + r) / math.sqrt(2.0)
    ref64 = (F.conv2d(xin.double(), w.double(), b.double(), padding=ks // 2) + r.double()) / math.sqrt(2.0)
    x1, x2 = (x[:, :Ci - dual].contiguous(), x[:, Ci - dual:].contiguous()) if dual else (x, None)
    mv = lambda t: None if t is None else t.to(dev)
    kw = dict(residual=mv(r), out_scale=1 / math.sqrt(2.0), x2=mv(x2), in_scale=mv(sc), in_shift=mv(sh), in_act=xform)
    out_b3 = ops.conv2d(mv(x1), mv(w), mv(b), force_split=split, **kw).cpu()
    out_f32 = ops.conv2d(mv(x1), mv(w), mv(b), **kw).cpu()
    assert rel_l2(out_b3, ref32) < OP_TOL, (B, Ci, Co, H, W, dual, xform)
    e_b3, e_f32 = rel_l2(out_b3.double(), ref64), rel_l2(out_f32.double(), ref64)
    print(f"conv_split {Ci}->{Co} @{B}x{H}x{W}: error vs fp64  {split} {e_b3:.2e}  fp32-MFMA {e_f32:.2e}  torch-fp32 {rel_l2(ref32.double(), ref64):.2e}")
    assert e_b3 < max(slack * e_f32, 3e-7), (split, e_b3, e_f32)


def check_conv_wino(dev, B, Ci, Co, H, W, dual=0, xform=True, res=True, xmul=1.0, wmul=None, slack=2.0):
    """The Winograd F(2,3) x fp16x2 3x3 kernel of the wide levels (kernels_conv_wino.h): the per-op gate against the fp32 oracle, an error
    against an fp64 convolution within `slack` x the fp32-MFMA kernel's, and its 4-row workgroup shape equal to the 8-row shape bit for bit.
    wmul: per-output-channel weight magnitudes spread over that many decades plus single outlier weights (the scale is per channel)."""
    from sgmse_amd import ops
    g = gen(B * 1000 + Ci + Co + H + W + 7)
    x = R(g, B, Ci, H, W) * xmul; w = R(g, Co, Ci, 3, 3) / math.sqrt(Ci * 9); b = R(g, Co) * xmul
    r = R(g, B, Co, H, W) * xmul if res else None
    if wmul:
        w = w * torch.logspace(-wmul / 2, wmul / 2, Co)[:, None, None, None]
        w[1, 0, 1, 1] *= 1e6; w[Co // 2, Ci - 1, 0, 2] *= 1e4
    if xmul != 1.0 and B > 1:
        x[0] *= 0.01
    sc = sh = None
    xin = x
    if xform:
        sc, sh = R(g, B, Ci), R(g, B, Ci)
        xin = x * sc[:, :, None, None] + sh[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    fin = lambda t: (t + (r.to(t.dtype) if res else 0)) / math.sqrt(2.0)
    ref32 = fin(F.conv2d(xin, w, b, padding=1))
    ref64 = fin(F.conv2d(xin.double(), w.double(), b.double(), padding=1))
    x1, x2 = (x[:, :Ci - dual].contiguous(), x[:, Ci - dual:].contiguous()) if dual else (x, None)
    mv = lambda t: None if t is None else t.to(dev)
    kw = dict(residual=mv(r), out_scale=1 / math.sqrt(2.0), x2=mv(x2), in_scale=mv(sc), in_shift=mv(sh), in_act=xform)
    out8 = ops.conv2d(mv(x1), mv(w), mv(b), force_split="wino", **kw).cpu()
    out4 = ops.conv2d(mv(x1), mv(w), mv(b), force_split="wino4", **kw).cpu()
    out_f32 = ops.conv2d(mv(x1), mv(w), mv(b), **kw).cpu()
    assert torch.equal(out8, out4), "the 4-row and 8-row Winograd shapes differ"
    # per-channel error: an outlier channel must not cost the others their accuracy
    if wmul:
        num = (out8.double() - ref64).pow(2).sum(dim=(0, 2, 3)).sqrt(); den = ref64.pow(2).sum(dim=(0, 2, 3)).sqrt()
        worst = float((num / den).max())
        print(f"conv_wino {Ci}->{Co} @{B}x{H}x{W} weights over {wmul} decades: worst per-channel error vs fp64 {worst:.2e}")
        assert worst < OP_TOL, worst
    else:
        assert rel_l2(out8, ref32) < OP_TOL, (B, Ci, Co, H, W, dual, xform)
    e_w, e_f32 = rel_l2(out8.double(), ref64), rel_l2(out_f32.double(), ref64)
    print(f"conv_wino {Ci}->{Co} @{B}x{H}x{W}: error vs fp64  winograd-fp16x2 {e_w:.2e}  fp32-MFMA {e_f32:.2e}  torch-fp32 {rel_l2(ref32.double(), ref64):.2e}")
    assert e_w < max(slack * e_f32, 3e-7), (e_w, e_f32)


def check_conv_wino2d(dev, B, Ci, Co, H, W, dual=0, xform=True, res=True, xmul=1.0, wmul=None, slack=2.0):
    """The 2-D Winograd F(2x2,3x3) x fp16x2 kernel (kernels_conv_wino2d.h; round 6 -- built and measured against the 1-D kernel, not the
    product path): the same gates as check_conv_wino -- per-op 1e-5 against the fp32 oracle and an error against an fp64 convolution within
    `slack` x the fp32-MFMA kernel's -- and the 1-D kernel's error printed beside it."""
    from sgmse_amd import ops
    g = gen(B * 1000 + Ci + Co + H + W + 11)
    x = R(g, B, Ci, H, W) * xmul; w = R(g, Co, Ci, 3, 3) / math.sqrt(Ci * 9); b = R(g, Co) * xmul
    r = R(g, B, Co, H, W) * xmul if res else None
    if wmul:
        w = w * torch.logspace(-wmul / 2, wmul / 2, Co)[:, None, None, None]
        w[1, 0, 1, 1] *= 1e6; w[Co // 2, Ci - 1, 0, 2] *= 1e4
    if xmul != 1.0 and B > 1:
        x[0] *= 0.01
    sc = sh = None
    xin = x
    if xform:
        sc, sh = R(g, B, Ci), R(g, B, Ci)
        xin = x * sc[:, :, None, None] + sh[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    fin = lambda t: (t + (r.to(t.dtype) if res else 0)) / math.sqrt(2.0)
    ref32 = fin(F.conv2d(xin, w, b, padding=1))
    ref64 = fin(F.conv2d(xin.double(), w.double(), b.double(), padding=1))
    x1, x2 = (x[:, :Ci - dual].contiguous(), x[:, Ci - dual:].contiguous()) if dual else (x, None)
    mv = lambda t: None if t is None else t.to(dev)
    kw = dict(residual=mv(r), out_scale=1 / math.sqrt(2.0), x2=mv(x2), in_scale=mv(sc), in_shift=mv(sh), in_act=xform)
    out2 = ops.conv2d(mv(x1), mv(w), mv(b), force_split="wino2d", **kw).cpu()
    out1 = ops.conv2d(mv(x1), mv(w), mv(b), force_split="wino", **kw).cpu()
    out_f32 = ops.conv2d(mv(x1), mv(w), mv(b), **kw).cpu()
    if wmul:
        num = (out2.double() - ref64).pow(2).sum(dim=(0, 2, 3)).sqrt(); den = ref64.pow(2).sum(dim=(0, 2, 3)).sqrt()
        worst = float((num / den).max())
        print(f"conv_wino2d {Ci}->{Co} @{B}x{H}x{W} weights over {wmul} decades: worst per-channel error vs fp64 {worst:.2e}")
        assert worst < OP_TOL, worst
    else:
        assert rel_l2(out2, ref32) < OP_TOL, (B, Ci, Co, H, W, dual, xform, rel_l2(out2, ref32))
    e_2, e_1, e_f32 = rel_l2(out2.double(), ref64), rel_l2(out1.double(), ref64), rel_l2(out_f32.double(), ref64)
    print(f"conv_wino2d {Ci}->{Co} @{B}x{H}x{W}: error vs fp64  F(2x2,3x3)-fp16x2 {e_2:.2e}  F(2,3)-fp16x2 {e_1:.2e}  fp32-MFMA {e_f32:.2e}  "
          f"torch-fp32 {rel_l2(ref32.double(), ref64):.2e}")
    assert e_2 < max(slack * e_f32, 3e-7), (e_2, e_f32)


def check_conv_thin_batch_independence(dev):
    """The VALU kernel of the pyramid convolutions gives an utterance the same bits alone and inside a batch (its accumulation order is a
    function of the layer only), with and without the residual / producer, at widths that leave partial 16 x 64 tiles."""
    from sgmse_amd import ops
    g = gen(91)
    B, Ci, H, W = 3, 64, 20, 72
    x = R(g, B, Ci, H, W); w = R(g, 4, Ci, 3, 3) / math.sqrt(Ci * 9); b = R(g, 4); r = R(g, B, 4, H, W)
    sc, sh = R(g, B, Ci), R(g, B, Ci)
    mv = lambda t: None if t is None else t.to(dev)
    for res, xf in ((True, True), (False, False)):
        kw = lambda sl: dict(residual=mv(r[sl]) if res else None, out_scale=0.5, in_scale=mv(sc[sl]) if xf else None,
                             in_shift=mv(sh[sl]) if xf else None, in_act=xf, force_split="thin")
        full = ops.conv2d(mv(x), mv(w), mv(b), **kw(slice(None))).cpu()
        for i in range(B):
            one = ops.conv2d(mv(x[i:i + 1]), mv(w), mv(b), **kw(slice(i, i + 1))).cpu()
            assert torch.equal(one[0], full[i]), (res, xf, i)


def check_groupnorm(dev, B, C, H, W, act=True, dual=0):
    from sgmse_amd import ops
    g = gen(C + H)
    x = R(g, B, C, H, W) * 2 + 0.5; gw = R(g, C); gb = R(g, C)
    ref = NO.group_norm(x, gw, gb)
    if act:
        ref = NO.silu(ref)
    x1, x2 = (x[:, :C - dual].contiguous(), x[:, C - dual:].contiguous()) if dual else (x, None)
    out = ops.group_norm(x1.to(dev), gw.to(dev), gb.to(dev), act=act, x2=None if x2 is None else x2.to(dev))
    assert rel_l2(out.cpu(), ref) < OP_TOL


def check_fir(dev, B=2, C=3, H=12, W=20):
    from sgmse_amd import ops
    x = R(gen(5), B, C, H, W)
    if H >= 2 and W >= 2:
        assert rel_l2(ops.fir_resample(x.to(dev), False).cpu(), NO.fir_down2(x)) < 1e-6
    assert rel_l2(ops.fir_resample(x.to(dev), True).cpu(), NO.fir_up2(x)) < 1e-6


def check_fir_fused(dev, B=1, C=2, H=20, W=72):
    """The resamplers as the residual blocks use them: fused GroupNorm-affine + SiLU producer and the raw second output
    (LDS-tiled kernels when W % 4 == 0 and W >= 64, per-pixel kernels otherwise)."""
    from sgmse_amd import ops
    g = gen(H * 100 + W)
    x, sc, sh = R(g, B, C, H, W), R(g, B, C), R(g, B, C)
    xin = x * sc[:, :, None, None] + sh[:, :, None, None]
    xin = xin * torch.sigmoid(xin)
    for up in (False, True):
        f = NO.fir_up2 if up else NO.fir_down2
        out, raw = ops.fir_resample(x.to(dev), up, in_scale=sc.to(dev), in_shift=sh.to(dev), in_act=True, return_raw=True)
        assert rel_l2(out.cpu(), f(xin)) < 2e-6 and rel_l2(raw.cpu(), f(x)) < 1e-6, (up, H, W)
        assert rel_l2(ops.fir_resample(x.to(dev), up).cpu(), f(x)) < 1e-6


def check_fir_golden(dev):
    """Against the reference's own upfirdn2d_native outputs (fixture fir.npz)."""
    from sgmse_amd import ops
    z = load("fir")
    x = torch.from_numpy(z["x"]).to(dev)
    assert rel_l2(ops.fir_resample(x, False).cpu(), z["down"]) < 1e-6
    assert rel_l2(ops.fir_resample(x, True).cpu(), z["up"]) < 1e-6
    k = torch.from_numpy(z["kern"]).to(dev)
    assert rel_l2(ops.upfirdn2d(x, k, up=3, down=2, pad=(2, 1)).cpu(), z["generic"]) < 1e-6
    k4 = torch.outer(torch.tensor([1., 3, 3, 1]), torch.tensor([1., 3, 3, 1]))
    k4 = (k4 / k4.sum()).to(dev)
    assert rel_l2(ops.upfirdn2d(x, k4, down=2, pad=(1, 1)).cpu(), z["down"]) < 1e-6
    assert rel_l2(ops.upfirdn2d(x, k4 * 4, up=2, pad=(2, 1)).cpu(), z["up"]) < 1e-6
    # the other element types the reference op dispatches (op/upfirdn2d_kernel.cu:311): double against the same fixtures in its own
    # precision, half within half's rounding of the fp32 result (fp32 accumulation, one rounding)
    for args, key in ((dict(up=3, down=2, pad=(2, 1)), "generic"), (dict(down=2, pad=(1, 1)), "down")):
        kk = k if key == "generic" else k4
        o64 = ops.upfirdn2d(x.double(), kk.double(), **args)
        assert o64.dtype == torch.float64 and rel_l2(o64.cpu(), torch.from_numpy(z[key]).double()) < 1e-6
        ref64 = F.conv2d(F.pad(x.cpu().double(), (1, 1, 1, 1)).reshape(-1, 1, x.shape[2] + 2, x.shape[3] + 2),
                         torch.flip(k4.cpu().double(), [0, 1])[None, None], stride=2) if key == "down" else None
        if ref64 is not None:          # double really computes in double: 1e-12 against an fp64 convolution of the same op
            assert rel_l2(o64.cpu().reshape(ref64.shape), ref64) < 1e-12
        o16 = ops.upfirdn2d(x.half(), kk.half(), **args)
        o32 = ops.upfirdn2d(x.half().float(), kk.half().float(), **args)
        assert o16.dtype == torch.float16 and torch.equal(o16.cpu(), o32.half().cpu())      # = the fp32 result of the same half inputs, rounded once


def check_attention(dev, B, C, S):
    from sgmse_amd import ops
    qkv = R(gen(S), B, 3 * C, S)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    w = torch.softmax(torch.einsum("bcs,bcr->bsr", q, k) * C ** -0.5, -1)
    ref = torch.einsum("bsr,bcr->bcs", w, v)
    assert rel_l2(ops.attention(qkv.to(dev)).cpu(), ref) < OP_TOL


NET_CASES = {
    "fwd_nf32": NO.NetCfg.for_variant("ncsnpp", nf=32),
    "fwd_48k_nf32": NO.NetCfg.for_variant("ncsnpp_48k", nf=32),
    "fwd_nf128": NO.NetCfg.for_variant("ncsnpp"),
    "fwd_v2_nf32": NO.NetCfg.for_variant("ncsnpp_v2", nf=32),
}


def make_backbone(cfg, dev, P=None):
    from sgmse_amd.backbones import BackboneRegistry
    P = synth.synth_params(cfg, seed=0) if P is None else P
    net = BackboneRegistry.get_by_name(cfg.variant)(nf=cfg.nf, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks,
                                                    attn_resolutions=cfg.attn_resolutions, image_size=cfg.image_size,
                                                    progressive=cfg.progressive, progressive_input=cfg.progressive_input)
    net.load_state_dict(P, strict=True)     # pins the reference state_dict name/shape contract
    return net.to(dev), P


def check_forward_golden(dev, name, batch=None):
    """NCSNpp.forward against the output of the reference's own module (fixture)."""
    cfg = NET_CASES[name]
    z = load(name)
    net, _ = make_backbone(cfg, dev)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    ref = torch.from_numpy(z["out"])
    if batch is not None:
        x, t, ref = x[:batch], t[:batch], ref[:batch]
    if cfg.variant == "ncsnpp_v2":         # ncsnpp_v2.forward(x, y, t)
        out = net(x[:, :1].contiguous().to(dev), x[:, 1:].contiguous().to(dev), t.to(dev))
    else:
        out = net(x.to(dev), t.to(dev))
    assert out.shape == ref.shape and out.dtype == torch.complex64
    err = rel_l2(out.cpu(), ref)
    print(f"{name} on {dev}: rel_l2 vs the reference's output = {err:.3e}")
    assert err < NET_TOL, (name, err)


def check_profile_forward(dev, name, batch=1):
    """The per-class profile of one evaluation (what bench.py's roofline object is computed from): event pairs are recorded per
    launch and read after the whole forward has been queued; the forward it times is the ordinary one, bit for bit, every
    launch is attributed to a class, and a second profile reuses the events."""
    cfg = NET_CASES[name]
    z = load(name)
    net, _ = make_backbone(cfg, dev)
    x, t = torch.from_numpy(z["x"])[:batch].to(dev), torch.from_numpy(z["t"])[:batch].to(dev)
    ref = net(x, t)
    ctx = net.engine(x.device)
    xy = x.contiguous()
    prof, out = ctx.profile_forward(xy, t)
    prof2, out2 = ctx.profile_forward(xy, t)
    assert torch.equal(out.reshape(ref.shape), ref) and torch.equal(out2, out)
    assert set(prof) == set(prof2)
    n = sum(v["launches"] for v in prof.values())
    assert n > 50 and n == sum(v["launches"] for v in prof2.values()), n
    for k, v in prof.items():
        assert v["ms"] >= 0.0 and v["work"] >= 0.0
        assert (v["launches"] > 0) == (v["work"] > 0.0), (k, v)
        assert v["work"] == prof2[k]["work"]
    assert prof["conv3x3_other"]["launches"] + prof["conv3x3_wide"]["launches"] > 0


def check_tile_independence(dev, name, batch=None):
    """The conv tile shape follows the workgroup count (batch size, U-Net level), so it must never change a bit of the
    result: widest tiles everywhere vs narrowest tiles everywhere (SGMSE_TILE_MIN_BLOCKS is read at engine creation)."""
    cfg = NET_CASES[name]
    z = load(name)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    if batch is not None:
        x, t = x[:batch], t[:batch]
    outs = []
    old = os.environ.get("SGMSE_TILE_MIN_BLOCKS")
    try:
        for m in ("1", "1000000000"):
            os.environ["SGMSE_TILE_MIN_BLOCKS"] = m
            net, _ = make_backbone(cfg, dev)
            outs.append(net(x.to(dev), t.to(dev)).cpu())
    finally:
        if old is None:
            os.environ.pop("SGMSE_TILE_MIN_BLOCKS", None)
        else:
            os.environ["SGMSE_TILE_MIN_BLOCKS"] = old
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0], torch.from_numpy(z["out"])[:len(x)]) < NET_TOL


def check_forward_b3_everywhere(dev, name="fwd_nf128", mode=None):
    """Network-level parity with a split kernel on EVERY eligible layer (SGMSE_SPLIT_MIN_TILES=1; by default only layers
    with >= 8 tiles per image use one), against the reference's own output, at the same gate as the fp32 kernels.
    mode: None = the build default (fp16x2), 1 = bf16x3, 0 = exact-fp32 MFMA kernels everywhere."""
    keys = {"SGMSE_SPLIT_MIN_TILES": "1"}
    if mode is not None:
        keys["SGMSE_CONV_SPLIT"] = str(mode)
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update(keys)
    try:
        check_forward_golden(dev, name)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_split_workgroup_shapes_bitwise(dev, name="fwd_nf128"):
    """The split kernels' 8-row and 4-row workgroup shapes (chosen by workgroup count, i.e. by batch size) give the same
    bits: full-width network with a split kernel on every eligible layer, widest vs narrowest shapes everywhere."""
    cfg = NET_CASES[name]
    z = load(name)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    keys = ("SGMSE_SPLIT_MIN_TILES", "SGMSE_TILE_MIN_BLOCKS")
    old = {k: os.environ.get(k) for k in keys}
    outs = []
    try:
        os.environ["SGMSE_SPLIT_MIN_TILES"] = "1"
        for m in ("1", "1000000000"):
            os.environ["SGMSE_TILE_MIN_BLOCKS"] = m
            net, _ = make_backbone(cfg, dev)
            outs.append(net(x.to(dev), t.to(dev)).cpu())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0], torch.from_numpy(z["out"])) < NET_TOL


def check_xcd_map_bitwise(dev, name="fwd_nf32", batch=None, knob="SGMSE_CONV_XCD_MAP"):
    """SGMSE_CONV_XCD_MAP=1 only permutes which workgroup computes which tile (XCD k takes the k-th contiguous eighth of a launch's
    tiles); SGMSE_SIDE_STREAM=1 (round 6) moves the pyramid branches and the unfolded shortcuts to a second stream (and defers arena
    releases across the fork): the network's output must not change by a bit."""
    cfg = NET_CASES[name]
    z = load(name)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    if batch is not None:
        x, t = x[:batch], t[:batch]
    old = os.environ.get(knob)
    outs = []
    try:
        for m in ("0", "1"):
            os.environ[knob] = m
            net, _ = make_backbone(cfg, dev)
            outs.append(net(x.to(dev), t.to(dev)).cpu())
    finally:
        if old is None:
            os.environ.pop(knob, None)
        else:
            os.environ[knob] = old
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0], torch.from_numpy(z["out"])[:len(x)]) < NET_TOL


def adversarial_params(cfg, kind, seed=0):
    """Synthetic state dicts that stress the range handling of the fp16x2 / bf16x3 kernels (no real checkpoint is reachable
    offline): every kind must keep the network-level gate of the ordinary fixtures.
      gn_inside   GroupNorm gamma up to 4 and beta up to 64 (the limits of the round-2 worst-case guard)
      gn_outside  gamma 4.5 in one module (beyond that guard: the per-utterance data-driven scale must keep the fp16x2 kernels)
      gn_wild     gamma up to 8 everywhere, single channels at 30, beta up to 500
      growth      residual stream growing ~10^3 across the network (Conv_1 and the shortcut of every block scaled up)
      outliers    a few output channels of some convolutions x 10^4 (heavy-tailed activations)
      zero_init   Conv_1 of every block at the reference's init_scale=0 magnitude (1e-10, layers.py:88-91): dead branches
      single_weights  single weights x 10^6 in Conv_0 (x 10^3 in Conv_1 and the 1x1 shortcut) of every third block: the fp16x2
                  weight scales are per OUTPUT CHANNEL, so the other channels of the layer keep their low
                  bits (a per-layer scale would push them 2^20 down into fp16 subnormals)"""
    P = synth.synth_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 101)
    res_blocks = sorted({k.rsplit(".", 2)[0] for k in P if k.endswith("Conv_1.weight")}, key=lambda n: int(n.split(".")[1]))
    if kind == "gn_wild":
        for k, v in P.items():
            if "GroupNorm" in k and k.endswith("weight"):
                v.copy_(0.5 + 7.5 * torch.rand(v.shape, generator=g))                    # [0.5, 8)
                v[0] = 8.0
                v[1] = -30.0
            elif "GroupNorm" in k and k.endswith("bias"):
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * 500.0)
    elif kind in ("gn_inside", "gn_outside"):
        for k, v in P.items():
            if "GroupNorm" in k and k.endswith("weight"):
                v.copy_(1.0 + 3.0 * torch.rand(v.shape, generator=g))                    # [1, 4)
                v[0] = 4.0
            elif "GroupNorm" in k and k.endswith("bias"):
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * 64.0)
        if kind == "gn_outside":
            P[res_blocks[len(res_blocks) // 2] + ".GroupNorm_1.weight"][3] = 4.5
    elif kind == "growth":
        f = 1000.0 ** (1.0 / len(res_blocks))
        for name in res_blocks:
            P[name + ".Conv_1.weight"] *= f * 1.6                                         # (x + h) / sqrt(2) keeps ~0.7 of each branch
            if name + ".Conv_2.weight" in P:
                P[name + ".Conv_2.weight"] *= f * 1.6
    elif kind == "outliers":
        for name in res_blocks[::3]:
            P[name + ".Conv_1.weight"][:3] *= 1e4
            P[name + ".Conv_0.weight"][5:7] *= 1e4
        P["all_modules.3.weight"][:2] *= 1e4
    elif kind == "single_weights":
        # (in Conv_0 only: its output is normalised by GroupNorm_1.  The same factor in Conv_1 / the shortcut lands on the residual
        #  stream, which nothing normalises, and overflows the REFERENCE's own fp32 arithmetic a few blocks later; those two get 10^3.)
        for i, name in enumerate(res_blocks[::3]):
            w0, w1 = P[name + ".Conv_0.weight"], P[name + ".Conv_1.weight"]
            w0[1 + i % 5, i % w0.shape[1], 1, 1] *= 1e6
            w0[64 + i % 7, (3 + i) % w0.shape[1], 0, 2] *= 1e6
            w1[2 + i % 7, (3 + i) % w1.shape[1], 0, 2] *= 1e3
            if name + ".Conv_2.weight" in P:
                P[name + ".Conv_2.weight"][i % 4, 1 + i % 3, 0, 0] *= 1e3
    elif kind == "zero_init":
        for name in res_blocks:
            P[name + ".Conv_1.weight"] *= 1e-10
    else:
        raise ValueError(kind)
    return P


def check_adversarial_checkpoint(dev, kind, nf=128, T=64, expect_mode=None):
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=nf)
    Pm = adversarial_params(cfg, kind)
    net, _ = make_backbone(cfg, dev, P=Pm)
    gx = torch.Generator().manual_seed(31)
    x = torch.randn(2, 2, 256, T, dtype=torch.complex64, generator=gx) * 0.3
    t = torch.tensor([0.7, 0.08])
    with torch.no_grad():
        ref = NO.ncsnpp_forward(Pm, cfg, x, t)
    out = net(x.to(dev), t.to(dev))
    mode = net.engine(torch.device(dev)).conv_split_mode()
    err = rel_l2(out.cpu(), ref)
    print(f"adversarial checkpoint '{kind}' (nf={nf}) on {dev}: split mode {mode}, output range {float(ref.abs().max()):.3g}, rel_l2 vs oracle {err:.3e}")
    assert torch.isfinite(torch.view_as_real(out)).all()
    if expect_mode is not None:
        assert mode == expect_mode, (kind, mode)
    assert err < NET_TOL, (kind, err)


def make_model(cfg, dev, P=None, sde="ouve", **kw):
    from sgmse_amd.model import ScoreModel
    P = synth.synth_params(cfg, seed=0) if P is None else P
    sde_kw = dict(theta=1.5, sigma_min=0.05, sigma_max=0.5) if sde == "ouve" else {}
    sde_kw.update(kw)
    m = ScoreModel(cfg.variant, sde, nf=cfg.nf, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks,
                   attn_resolutions=cfg.attn_resolutions, image_size=cfg.image_size, progressive=cfg.progressive,
                   progressive_input=cfg.progressive_input, **sde_kw)
    m.dnn.load_state_dict(P, strict=True)
    m.to(dev)
    m.eval()
    return m, P


def replay_noise(shape, ndraws, seed=7):
    rep = SO.NoiseReplay(seed)
    like = torch.zeros(shape, dtype=torch.complex64)
    return torch.stack([rep(like) for _ in range(ndraws)])


def check_ode_rk45(dev, full=True, name="ode_rk45"):
    """The reference's adaptive probability-flow sampler (get_ode_sampler with denoise=False: scipy RK45 over the flattened state,
    sampling/__init__.py:96-143) against the reference's own run (oracle/make_golden_ode.py).  Two fixtures: ``ode_rk45`` at
    rtol = atol = 1e-3 (92 evaluations) and ``ode_rk45_default`` at the reference's default 1e-5 (722 evaluations).
    Gates: (1) the drift handed to the solver at the reference's own evaluation points, 1e-5 -- where the product computes; (2) the
    solver's evaluation count within two steps of the reference's; (3) the end state: at the default tolerance the integration is
    well conditioned (the oracle's own end state is 1.8e-5 from the reference's) and the gate is the samplers' 1e-4; at 1e-3 the step
    control amplifies last-digit differences of the network (oracle: 3.7e-4), so the bound is 5x the oracle's own deviation."""
    from sgmse_amd import sampling
    z = load(name)
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    m, _ = make_model(cfg, dev)
    y = torch.from_numpy(z["y"]).to(dev)
    sde = m.sde.copy()
    rsde = sde.reverse(m, probability_flow=True)
    worst = 0.0
    with torch.no_grad():
        for t, xk, fk in zip(z["probe_t"], z["probe_x"], z["probe_f"]):
            xt = torch.from_numpy(xk.reshape(tuple(y.shape))).to(dev)
            f = rsde.sde(xt, y, torch.ones(y.shape[0], device=dev) * float(t))[0]
            worst = max(worst, rel_l2(f.cpu().reshape(-1), torch.from_numpy(fk)))
    print(f"{name} on {dev}: drift at {len(z['probe_t'])} of the reference's evaluation points, worst rel_l2 = {worst:.3e}")
    assert worst < 1e-5, worst
    if not full:
        return
    noise = replay_noise(tuple(y.shape), 1).to(dev)
    sampler = m.get_ode_sampler(y, denoise=False, rtol=float(z["rtol"]), atol=float(z["atol"]), method="RK45", noise=noise)
    out, nfe = sampler()
    err = rel_l2(out.cpu(), torch.from_numpy(z["out"]))
    own = float(z["oracle_vs_reference"])
    bound = SAMPLER_TOL if float(z["rtol"]) <= 1e-5 else 5.0 * own
    print(f"{name} on {dev}: solver evaluations {nfe} (reference {int(z['nfe'])}), end state rel_l2 vs the reference's = {err:.3e} "
          f"(the oracle's own: {own:.3e}; bound {bound:.1e})")
    assert abs(nfe - int(z["nfe"])) <= 12 and err < bound, (nfe, err, bound)


def check_sampler_golden(dev, tag, batch=None, use_graph=True):
    """get_pc_sampler(...)() with replayed noise against the reference's own sampler output (fixture).  Utterances in
    a batch are independent on this path, so a batch prefix of the fixture is a valid smaller case."""
    z = load(tag)
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    m, _ = make_model(cfg, dev)
    y = torch.from_numpy(z["y"])
    ref = torch.from_numpy(z["out"])
    if tag.startswith("pfode"):
        N, ndraws = 6, 1
        noise = replay_noise(y.shape, ndraws)
        if batch is not None:
            y, ref, noise = y[:batch], ref[:batch], noise[:, :batch]
        sampler = m.get_ode_sampler(y.to(dev), N=N, noise=noise.contiguous().to(dev), use_graph=use_graph)
        out, nfe = sampler()
        assert nfe == N
    else:
        N, corr = int(z["N"]), str(z["corrector"])
        csteps = int(z["corrector_steps"]) if "corrector_steps" in z.files else 1        # (pc_N4_c2: correctors.py:69-81 looped twice)
        ndraws = 1 + N * ((csteps + 1) if corr != "none" else 1)
        noise = replay_noise(y.shape, ndraws)
        if batch is not None:
            y, ref, noise = y[:batch], ref[:batch], noise[:, :batch]
        sampler = m.get_pc_sampler(str(z["predictor"]), corr, y.to(dev), N=N, snr=float(z["snr"]), corrector_steps=csteps,
                                   noise=noise.contiguous().to(dev), use_graph=use_graph)
        out, nfe = sampler()
        assert nfe == int(z["nfe"]) == N * ((csteps if corr != "none" else 0) + 1)
    assert rel_l2(out.cpu(), ref) < SAMPLER_TOL, tag
    return out


def check_front_end(dev, fc, L):
    from sgmse_amd.data_module import SpecsDataModule
    dm = SpecsDataModule(n_fft=fc.n_fft, hop_length=fc.hop_length, spec_factor=fc.spec_factor,
                         spec_abs_exponent=fc.spec_abs_exponent, window=fc.window, transform_type=fc.transform_type)
    sig = synth.synth_waveform(L, seed=2, batch=2)
    S_ref = FO.stft(sig, fc)
    S = dm.stft(sig.to(dev))
    assert S.shape == S_ref.shape and rel_l2(S.cpu(), S_ref) < 5e-6
    assert rel_l2(dm.spec_fwd(S_ref.to(dev)).cpu(), FO.spec_fwd(S_ref, fc)) < 5e-6
    Yf = FO.spec_fwd(S_ref, fc)
    assert rel_l2(dm.spec_back(Yf.to(dev)).cpu(), FO.spec_back(Yf, fc)) < 5e-6
    assert rel_l2(dm.istft(S_ref.to(dev), L).cpu(), FO.istft(S_ref, fc, L)) < 5e-6
    # round trip (size-independent property): istft(stft(x)) == x
    assert rel_l2(dm.istft(dm.stft(sig.to(dev)), L).cpu(), sig) < 1e-5
    # padded spectrogram, shorter length: the tail envelope quirk of SURVEY Appendix C
    Yp = FO.pad_spec(S_ref[None, :1], "zero_pad")[0]
    assert rel_l2(dm.istft(Yp.to(dev), L).cpu(), FO.istft(Yp, fc, L)) < 5e-6


def check_front_golden(dev, name, fc, L):
    """The HIP front end against outputs of the reference's own SpecsDataModule (tests/golden/front.npz, oracle/make_golden_front.py):
    the `log` / `none` spectrogram transforms and the `sqrthann` window next to the defaults (data_module.py:13-19,162-218)."""
    from sgmse_amd.data_module import SpecsDataModule
    z = load("front")
    dm = SpecsDataModule(n_fft=fc.n_fft, hop_length=fc.hop_length, spec_factor=fc.spec_factor,
                         spec_abs_exponent=fc.spec_abs_exponent, window=fc.window, transform_type=fc.transform_type)
    sig = synth.synth_waveform(L, seed=5, batch=2)
    S_ref = torch.from_numpy(z[name + "/stft"])
    assert rel_l2(dm.stft(sig.to(dev)).cpu(), S_ref) < 5e-6
    assert rel_l2(dm.spec_fwd(S_ref.to(dev)).cpu(), z[name + "/fwd"]) < 5e-6
    assert rel_l2(dm.spec_back(torch.from_numpy(z[name + "/fwd"]).to(dev)).cpu(), z[name + "/back"]) < 5e-6
    assert rel_l2(dm.istft(S_ref.to(dev), L).cpu(), z[name + "/istft"]) < 5e-6


def check_enhance(dev, L=8000, N=2):
    """enhancement.py:62-99 end to end (normalise -> STFT -> sampler -> iSTFT -> renormalise) with replayed noise against
    the oracle pipeline; the north-star gate: relative L2 <= 1e-3 on the enhanced waveform."""
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    m, P = make_model(cfg, dev)
    fc = FO.FrontCfg()
    y = synth.synth_waveform(L, seed=0, batch=1)
    sde = SO.OUVE(1.5, 0.05, 0.5, N)
    rep = SO.NoiseReplay(7)
    x_ref = FO.enhance(y, fc, lambda Y: SO.pc_sample(sde, lambda a, b, c: NO.score_fn(P, cfg, a, b, c), Y, rep, eps=0.03, snr=0.5)[0])
    noise = torch.stack(rep.draws).to(dev)
    x_hat = m.enhance(y, N=N, noise=noise)
    assert rel_l2(x_hat, x_ref) < WAVE_TOL
    xb, nfe = m.enhance_batch(y, N=N, noise=noise)
    assert nfe == 2 * N and rel_l2(xb[0].cpu(), x_ref) < WAVE_TOL


def run_full_config(dev, name):
    """A BASELINE.json configuration itself, one utterance, against the REFERENCE's own output (fixture written by
    oracle/make_golden_full.py from the reference's full-width network, OUVESDE and pc_sampler with replayed noise):
    full-width network x full utterance x full N, through the batched entry point.  Inputs are rebuilt from seeds
    (oracle/full_cases.py).  Returns (rel. L2 of the sampled spectrogram, of the enhanced waveform, NFE); also called by
    bench.py's parity leg, which reports the two figures of the benched build in the driver's line."""
    from oracle.full_cases import FULL_CASES, front_cfg
    c = FULL_CASES[name]
    z = load(name)
    cfg = NO.NetCfg.for_variant(c["variant"])
    fc = front_cfg(c["front"])
    m, _ = make_model(cfg, dev, P=synth.synth_params(cfg, seed=c["param_seed"]), N=c["N"], n_fft=fc.n_fft, hop_length=fc.hop_length,
                      spec_factor=fc.spec_factor, spec_abs_exponent=fc.spec_abs_exponent, **c["sde"])
    y = synth.synth_waveform(c["L"], seed=c["wave_seed"], batch=1)
    frames = c["L"] // fc.hop_length + 1
    T = (frames + 63) // 64 * 64
    ndraws = 1 + (2 * c["N"] if c["sampler"] == "pc" else 0)
    noise = replay_noise((1, 1, fc.n_fft // 2 + 1, T), ndraws, seed=c["noise_seed"]).to(dev)
    got = {}
    orig = m.get_pc_sampler if c["sampler"] == "pc" else m.get_ode_sampler

    def spy(*a, **k):                       # keep the sampled spectrogram of the run below
        s = orig(*a, **k)
        def run():
            got["spec"], got["nfe"] = s()
            return got["spec"], got["nfe"]
        return run
    setattr(m, "get_pc_sampler" if c["sampler"] == "pc" else "get_ode_sampler", spy)
    x_hat, nfe = m.enhance_batch(y.to(dev), N=c["N"], snr=c["snr"], sampler_type=c["sampler"], noise=noise, pad_mode=c["pad"])
    e_spec, e_wave = rel_l2(got["spec"].cpu(), z["spec"]), rel_l2(x_hat[0].cpu(), z["wave"])
    assert nfe == int(z["nfe"]) and got["spec"].shape == z["spec"].shape
    return e_spec, e_wave, nfe


def check_full_config(dev, name):
    """run_full_config with the gates: sampled spectrogram <= 1e-4, enhanced waveform <= 1e-3 (north star)."""
    e_spec, e_wave, nfe = run_full_config(dev, name)
    print(f"{name} on {dev}: {nfe} NFE, rel_l2 vs the reference: spectrogram {e_spec:.3e}, waveform {e_wave:.3e}")
    assert e_spec < SAMPLER_TOL and e_wave < WAVE_TOL, (name, e_spec, e_wave)


def check_batch_equals_singles(dev, B=4):
    """enhance_batch over B utterances equals the B single-utterance runs bit for bit (full-width network, 4 s, replayed
    noise; short N to bound the run time): utterances never interact and no kernel choice depends on the batch size in a
    way that changes a bit."""
    cfg = NO.NetCfg.for_variant("ncsnpp")
    m, _ = make_model(cfg, dev)
    y = synth.synth_waveform(64000, seed=3, batch=B)
    N = 2
    noise = replay_noise((B, 1, 256, 512), 1 + 2 * N).to(dev)
    full, _ = m.enhance_batch(y.to(dev), N=N, noise=noise)
    for i in range(B):
        one, _ = m.enhance_batch(y[i:i + 1].to(dev), N=N, noise=noise[:, i:i + 1].contiguous())
        assert torch.equal(one[0], full[i]), i


def check_sampler_oracle(dev, variant="ncsnpp_48k", N=2, corrector="ald", snr=0.33, F_=192, T=64, B=1, use_graph=True):
    """PC sampler of a reduced-width model against the oracle loop with replayed noise (covers ncsnpp_48k, whose
    output_layer / division-by-t order differs: ncsnpp_48k.py:414-421)."""
    cfg = NO.NetCfg.for_variant(variant, nf=32)
    sde_kw = dict(theta=2.0, sigma_min=0.1, sigma_max=1.0) if variant == "ncsnpp_48k" else {}
    m, P = make_model(cfg, dev, **sde_kw)
    y = synth.synth_spec(B, F_, T, seed=4)
    so = SO.OUVE(m.sde.theta, m.sde.sigma_min, m.sde.sigma_max, N)
    rep = SO.NoiseReplay(7)
    ref, nfe_ref = SO.pc_sample(so, lambda a, b, c: NO.score_fn(P, cfg, a, b, c), y, rep, eps=0.03, snr=snr, corrector=corrector)
    noise = torch.stack(rep.draws).to(dev)
    out, nfe = m.get_pc_sampler("reverse_diffusion", corrector, y.to(dev), N=N, snr=snr, noise=noise, use_graph=use_graph)()
    assert nfe == nfe_ref and rel_l2(out.cpu(), ref) < SAMPLER_TOL


def check_sampler_v2(dev, loss_type, network_scaling, c_in, c_out, c_skip, N=2, use_graph=True):
    """ncsnpp_v2 model: the new-code ScoreModel.forward (model.py:284-304) -- scaled inputs, network scaling, loss-type
    dependent output map -- applied inside the fused sampler, against the oracle loop over oracle.score_fn_v2; and the
    directly called forward against the same oracle function."""
    cfg = NO.NetCfg.for_variant("ncsnpp_v2", nf=32)
    wrap = dict(loss_type=loss_type, network_scaling=network_scaling, c_in=c_in, c_out=c_out, c_skip=c_skip, sigma_data=0.1)
    m, P = make_model(cfg, dev, **wrap)
    y = synth.synth_spec(1, 256, 64, seed=4)
    so = SO.OUVE(1.5, 0.05, 0.5, N)
    score = lambda a, b, c: NO.score_fn_v2(P, cfg, so, a, b, c, **wrap)
    g = torch.Generator().manual_seed(3)
    xt = y + 0.3 * torch.randn(y.shape, dtype=torch.complex64, generator=g)
    tt = torch.tensor([0.6])
    assert rel_l2(m(xt.to(dev), y.to(dev), tt.to(dev)).cpu(), score(xt, y, tt)) < NET_TOL
    rep = SO.NoiseReplay(7)
    ref, nfe_ref = SO.pc_sample(so, score, y, rep, eps=0.03, snr=0.5)
    noise = torch.stack(rep.draws).to(dev)
    out, nfe = m.get_pc_sampler("reverse_diffusion", "ald", y.to(dev), N=N, snr=0.5, noise=noise, use_graph=use_graph)()
    assert nfe == nfe_ref and rel_l2(out.cpu(), ref) < SAMPLER_TOL


def check_sb_golden(dev, stype, batch=None, use_graph=True):
    """Schroedinger-bridge sampler ('ode' / 'sde', N=4) of an ncsnpp_v2 data-prediction model against the output of the
    reference's own get_sb_sampler + SBVESDE + NCSNpp_v2 (fixture), with replayed noise for 'sde'."""
    z = load(f"sb_{stype}_N4")
    cfg = NO.NetCfg.for_variant("ncsnpp_v2", nf=32)
    m, Pm = make_model(cfg, dev, sde="sbve", k=2.6, c=0.4, N=4, loss_type="data_prediction")
    y, ref = torch.from_numpy(z["y"]), torch.from_numpy(z["out"])
    noise = replay_noise(y.shape, 4) if stype == "sde" else None
    if batch is not None:
        y, ref = y[:batch], ref[:batch]
        noise = None if noise is None else noise[:, :batch]
    sampler = m.get_sb_sampler(m.sde, y.to(dev), sampler_type=stype, n_steps=4,
                               noise=None if noise is None else noise.contiguous().to(dev), use_graph=use_graph)
    out, n = sampler()
    err = rel_l2(out.cpu(), ref)
    print(f"sb_{stype}_N4 on {dev}: rel_l2 vs reference = {err:.3e}")
    assert n == 4
    if stype == "sde":
        assert err < SAMPLER_TOL, (stype, err)
        return
    # 'ode': the reference's first step adds the estimate to 5457 y and cancels (sdes.py:235-313 with k = 2.6, c = 0.4), which turns
    # a relative difference delta between two implementations' network outputs into ~sqrt(delta * 2e-4) in the state -- the CPU oracle
    # itself lands 1e-5 ... 7e-5 from the fixture depending on the host's libm (profiles/r03_sb_conditioning.txt, r03_sb_probe.txt).
    # The state is therefore recorded and only bounded loosely; what is GATED is what the kernels are responsible for: every network
    # estimate of the reference-style loop against the oracle's network at the SAME input, step by step.
    from oracle.sde_oracle import SBVE
    sv = SBVE(2.6, 0.4, 4)
    sde = m.sde
    b4 = lambda v: v[:, None, None, None]
    worst = 0.0
    with torch.no_grad():
        xt = y.clone()
        ts = torch.linspace(sde.T, 1e-4, sde.N + 1)
        sp, _, sbp, ap, _, _ = sde._sigmas_alphas(ts[0] * torch.ones(xt.shape[0]))
        for i, t in enumerate(ts[1:]):
            time = t * torch.ones(xt.shape[0])
            st, sT, sbt, at, aT, _ = sde._sigmas_alphas(time)
            est_hip = m(xt.to(dev), y.to(dev), time.to(dev)).cpu()
            est_orc = NO.score_fn_v2(Pm, cfg, sv, xt, y, time, loss_type="data_prediction")
            worst = max(worst, rel_l2(est_hip, est_orc))
            w_prev = at * st * sbt / (ap * sp * sbp + sde.eps)
            w_est = at / (sT ** 2 + sde.eps) * (sbt ** 2 - sbp * st * sbt / (sp + sde.eps))
            w_y = at / (aT * sT ** 2 + sde.eps) * (st ** 2 - sp * st * sbt / (sbp + sde.eps))
            xt = b4(w_prev) * xt + b4(w_est) * est_hip + b4(w_y) * y
            sp, sbp, ap = st, sbt, at
    # State bound from the conditioning model of tools/sb_conditioning.py (profiles/r03_sb_conditioning.txt): a relative difference delta of
    # the network estimates moves the sample by sqrt(delta * 0.446 * 4.5e-4) (the first step's quantum ulp(w_prev |y|) / |y| = 4.57e-4 turns a
    # perturbation into a random walk over that grid); measured on MI355X 7.2e-5 at delta = 5.1e-6, 2.3x the model.  Bound: 4x the model at
    # the delta measured in THIS run (round 4: a flat 1e-3) -- a 10x slip of the fused loop's own weight arithmetic no longer passes.
    pred = math.sqrt(max(worst, 1e-7) * 0.446 * 4.5e-4)
    mutual = rel_l2(xt, out.cpu())
    print(f"sb_ode_N4 on {dev}: worst per-step network estimate vs the oracle at the same input = {worst:.3e} (gate {OP_TOL:.0e}); "
          f"state vs the reference's output: fused loop {err:.3e}, reference-style loop {rel_l2(xt, ref):.3e}, fused vs reference-style loop "
          f"{mutual:.3e} (conditioning model at this delta: {pred:.1e}; bounds 4x / 8x)")
    assert worst < OP_TOL, worst
    assert err < 4.0 * pred and mutual < 8.0 * pred, (err, mutual, pred)


def check_weight_reload(dev):
    """Parameters changed in place (what an EMA swap does) reach the engine after mark_weights_changed(); the old packed
    buffers are released."""
    cfg = NET_CASES["fwd_nf32"]
    net, _ = make_backbone(cfg, dev)
    z = load("fwd_nf32")
    x, t = torch.from_numpy(z["x"])[:1].to(dev), torch.from_numpy(z["t"])[:1].to(dev)
    a = net(x, t)
    with torch.no_grad():
        net.output_layer.weight.mul_(2.0)
        net.output_layer.bias.mul_(2.0)
    net.mark_weights_changed()
    b = net(x, t)
    assert torch.equal(b, 2.0 * a)


def check_enhancement_script(dev, tmp_path, monkeypatch):
    """python -m sgmse_amd.enhancement: checkpoint + directory of wav files of mixed lengths -> enhanced directory (batched by
    padded frame count; seeded noise is a function of the file, so neither the batch size nor the rank count changes a sample;
    rank sharding covers every file exactly once)."""
    from scipy.io import wavfile
    from sgmse_amd import enhancement as E
    from sgmse_amd.model import ScoreModel
    from sgmse_amd.data_module import SpecsDataModule
    hp = dict(backbone="ncsnpp", sde="ouve", nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30, t_eps=0.03,
              data_module_cls=SpecsDataModule, n_fft=510, hop_length=128, spec_factor=0.15, spec_abs_exponent=0.5)
    src = ScoreModel(**hp)
    src.dnn.load_state_dict(synth.synth_params(NET_CASES["fwd_nf32"], seed=0))
    ckpt = tmp_path / "m.ckpt"
    torch.save({"state_dict": {"dnn." + k: v.clone() for k, v in src.dnn.state_dict().items()}, "hyper_parameters": hp}, ckpt)
    noisy = tmp_path / "noisy"
    (noisy / "sub").mkdir(parents=True)
    rng = torch.Generator().manual_seed(5)
    lengths = {"a.wav": 2000, "b.wav": 2000, "sub/c.wav": 8500}      # 16, 16 and 67 frames: padded to 64, 64 and 128
    for name, L in lengths.items():
        wavfile.write(str(noisy / name), 16000, (0.1 * torch.randn(L, generator=rng)).numpy())
    base = ["--test_dir", str(noisy), "--ckpt", str(ckpt), "--device", str(dev), "--N", "1", "--seed", "7"]
    with pytest.warns(UserWarning):          # checkpoint without EMA weights (model.py:106)
        assert E.main(base + ["--enhanced_dir", str(tmp_path / "o1")]) == 3
        assert E.main(base + ["--enhanced_dir", str(tmp_path / "o2"), "--batch_size", "2"]) == 3
    for name, L in lengths.items():
        sr, x1 = wavfile.read(str(tmp_path / "o1" / name))
        _, x2 = wavfile.read(str(tmp_path / "o2" / name))
        assert sr == 16000 and x1.shape == (L,) and np.isfinite(x1).all() and np.abs(x1).max() > 0
        assert np.array_equal(x1, x2)        # noise = f(seed, index in the file list): the batch size does not matter
    with pytest.warns(UserWarning):          # all three files in ONE ragged batch (frames 64, 64, 128): still the same samples
        assert E.main(base + ["--enhanced_dir", str(tmp_path / "o4"), "--ragged", "--batch_size", "3"]) == 3
    for name in lengths:
        assert np.array_equal(wavfile.read(str(tmp_path / "o1" / name))[1], wavfile.read(str(tmp_path / "o4" / name))[1])
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("MASTER_PORT", raising=False)      # no rendezvous: every rank reads the checkpoint itself
    seen = 0
    with pytest.warns(UserWarning):
        for r in (0, 1):
            monkeypatch.setenv("RANK", str(r))
            seen += E.main(base + ["--enhanced_dir", str(tmp_path / "o3")])
    assert seen == 3 and sorted(str(p.relative_to(tmp_path / "o3")) for p in (tmp_path / "o3").rglob("*.wav")) == sorted(lengths)
    for name in lengths:                                   # two ranks, other batches: the same samples
        assert np.array_equal(wavfile.read(str(tmp_path / "o1" / name))[1], wavfile.read(str(tmp_path / "o3" / name))[1])


def check_reference_script_unmodified(dev, tmp_path, monkeypatch):
    """The reference's own ``enhancement.py``, byte for byte as it lies in the reference tree, run on top of this package:
    ``sgmse_amd/compat`` resolves its ``from sgmse...`` imports, ``sgmse_amd/compat/shims`` stands in for the audio packages this
    image lacks (soundfile / torchaudio / librosa; appended to the path, so real installations win).  Its output for the first
    file equals, bit for bit, what ``python -m sgmse_amd.enhancement`` writes from the same torch seed; the second file (8 kHz:
    the resampling branch) is checked for rate, length and sanity."""
    import runpy
    from scipy.io import wavfile
    from sgmse_amd import enhancement as E
    from sgmse_amd.model import ScoreModel
    from sgmse_amd.data_module import SpecsDataModule
    ref = os.path.join(os.environ.get("SGMSE_REFERENCE_DIR", "/root/reference"), "enhancement.py")
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this machine")
    hp = dict(backbone="ncsnpp", sde="ouve", nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30, t_eps=0.03,
              data_module_cls=SpecsDataModule, n_fft=510, hop_length=128, spec_factor=0.15, spec_abs_exponent=0.5)
    src = ScoreModel(**hp)
    src.dnn.load_state_dict(synth.synth_params(NET_CASES["fwd_nf32"], seed=0))
    ckpt = tmp_path / "m.ckpt"
    torch.save({"state_dict": {"dnn." + k: v.clone() for k, v in src.dnn.state_dict().items()}, "hyper_parameters": hp}, ckpt)
    noisy = tmp_path / "noisy"
    (noisy / "sub").mkdir(parents=True)
    rng = torch.Generator().manual_seed(5)
    wavfile.write(str(noisy / "a.wav"), 16000, (0.1 * torch.randn(2000, generator=rng)).numpy())
    wavfile.write(str(noisy / "sub" / "b.wav"), 8000, (0.1 * torch.randn(1500, generator=rng)).numpy())     # -> 3000 samples at 16 kHz
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "sgmse_amd", "compat"))
    sys.path.append(os.path.join(root, "sgmse_amd", "compat", "shims"))
    args = ["--test_dir", str(noisy), "--ckpt", str(ckpt), "--device", str(dev), "--N", "1"]
    before = set(sys.modules)
    try:
        monkeypatch.setattr(sys, "argv", [ref] + args + ["--enhanced_dir", str(tmp_path / "ref_out")])
        torch.manual_seed(3)
        with pytest.warns(UserWarning):          # checkpoint without EMA weights (model.py:106)
            runpy.run_path(ref, run_name="__main__")
    finally:
        sys.path.remove(os.path.join(root, "sgmse_amd", "compat", "shims"))
        for name in set(sys.modules) - before:          # the alias package and the stand-ins leave with the script
            if name.split(".")[0] in ("sgmse", "soundfile", "torchaudio", "librosa"):
                del sys.modules[name]
    torch.manual_seed(3)
    with pytest.warns(UserWarning):
        assert E.main(args + ["--enhanced_dir", str(tmp_path / "own_out"), "--batch_size", "1"]) == 2
    sr, a_ref = wavfile.read(str(tmp_path / "ref_out" / "a.wav"))
    _, a_own = wavfile.read(str(tmp_path / "own_out" / "a.wav"))
    assert sr == 16000 and a_ref.shape == (2000,) and np.array_equal(a_ref, a_own)
    sr, b_ref = wavfile.read(str(tmp_path / "ref_out" / "sub" / "b.wav"))
    assert sr == 16000 and b_ref.shape == (3000,) and np.isfinite(b_ref).all() and np.abs(b_ref).max() > 0


def check_poison_independence(dev, name="fwd_nf128", every_layer_split=False):
    """SGMSE_POISON=1 fills every device allocation, and the activation arena before every forward, with NaN bit patterns
    (0xFF bytes): a kernel that reads memory nobody wrote -- a halo outside the image, a statistics slot of a masked row, a
    split-K partial of a masked element -- turns that into a NaN.  The forward must give the same bits either way."""
    cfg = NET_CASES[name]
    z = load(name)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    keys = {"SGMSE_POISON": "0"}
    if every_layer_split:
        keys["SGMSE_SPLIT_MIN_TILES"] = "1"
    old = {k: os.environ.get(k) for k in keys}
    outs = []
    try:
        for poison in ("0", "1"):
            os.environ.update(keys)
            os.environ["SGMSE_POISON"] = poison
            net, _ = make_backbone(cfg, dev)
            outs.append(net(x.to(dev), t.to(dev)).cpu())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert torch.isfinite(torch.view_as_real(outs[1])).all()
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[1], torch.from_numpy(z["out"])) < NET_TOL


def check_bits_under_outside_load(dev, seconds=26):
    """Round 6: the results must not depend on what ELSE runs on the GPU.  conv3x3_thin_kernel's v_pk_fma_f32 gave other bits (|diff| up
    to 0.8 at the network's output) in about half of the launches whenever another PROCESS shared the device -- alone it was deterministic,
    so no test of rounds 4-5 saw it (kernels_conv_thin.h).  A second process loads the device with forwards of its own while this one repeats
    the thin convolution, the full-width and the reduced-width network, and a seeded sampler run, each against its own solo result."""
    import subprocess
    import sys
    import time
    from sgmse_amd import ops
    g = gen(77)
    x, w, b, r = R(g, 1, 128, 256, 64), R(g, 4, 128, 3, 3) / math.sqrt(128 * 9), R(g, 4), R(g, 1, 4, 256, 64)
    sc, sh = R(g, 1, 128), R(g, 1, 128)
    mv = lambda t: t.to(dev)
    xd, wd, bd, rd, scd, shd = (mv(t) for t in (x, w, b, r, sc, sh))
    cases = {"conv3x3_thin_kernel 128->4 @256x64": (lambda: ops.conv2d(xd, wd, bd, residual=rd, out_scale=0.7, in_scale=scd, in_shift=shd, in_act=True,
                                                                      force_split="thin"), 30)}
    for name, T, reps in (("fwd_nf128", 64, 12), ("fwd_nf32", 64, 12)):
        net, _ = make_backbone(NET_CASES[name], dev)
        xx = (torch.randn(1, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3).to(dev)
        tt = torch.tensor([0.4], device=dev)
        cases[f"{name} forward, T = {T}"] = ((lambda net=net, xx=xx, tt=tt: net(xx, tt)), reps)
    m, _ = make_model(NET_CASES["fwd_nf32"], dev)
    y = synth.synth_spec(2, 256, 64, seed=3).to(dev)
    cases["pc sampler N = 3 (captured graph, in-kernel noise, seed 5)"] = (lambda: m.get_pc_sampler("reverse_diffusion", "ald", y, N=3, snr=0.5, seed=5)()[0], 6)
    wav = synth.synth_waveform(8000, seed=1, batch=2).to(dev)
    cases["enhance_batch: STFT -> PC sampler N = 2 -> iSTFT (seed 9)"] = (lambda: torch.as_tensor(m.enhance_batch(wav, N=2, snr=0.5, seed=9)[0]), 6)
    ref = {k: f().cpu() for k, (f, _) in cases.items()}
    for k, (f, _) in cases.items():
        assert torch.equal(f().cpu(), ref[k]), k + ": not reproducible even alone"
    # the load process announces itself once its first forward has run (a cold box can take a minute to import torch) and is ended by this
    # test, not by its own clock: the checks below always run beside it
    import tempfile
    stop_file = os.path.join(tempfile.mkdtemp(prefix="sgmse_load_"), "stop")
    load = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "probes", "concurrency_bits_probe.py"), "--load"],
                            env=dict({kk: vv for kk, vv in os.environ.items() if not kk.startswith("SGMSE_")}, LOAD_SECONDS=str(max(seconds, 600)),
                                     LOAD_STOP_FILE=stop_file),
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        t0 = time.time()
        line = ""
        while "LOAD-RUNNING" not in line:
            line = load.stdout.readline()
            assert line or load.poll() is None, "the load process ended before it started loading the device"
            assert time.time() - t0 < 400, "the load process did not come up"
        for k, (f, reps) in cases.items():
            bad = sum(0 if torch.equal(f().cpu(), ref[k]) else 1 for _ in range(reps))
            print(f"under outside load: {k}: {reps - bad} of {reps} identical to the solo result")
            assert bad == 0, (k, bad, reps)
        assert load.poll() is None, "the load process ended before the checks did"
    finally:
        open(stop_file, "w").close()                 # the load loop ends by itself within one forward ...
        try:
            load.wait(timeout=60)
        except subprocess.TimeoutExpired:            # ... or is ended
            load.kill()
            load.wait(timeout=60)


def check_ragged_batch(dev, name="fwd_nf32", frames=(128, 64, 192), sampler=True, quick=False):
    """Ragged batches (sgmse_set_frames): utterances of different frame counts in ONE launch, each with its own row stride in
    every tensor.  Forward and samplers must give every utterance the bits of its single-utterance call -- the arithmetic of an
    utterance does not depend on what else is in the batch, nor on its own length class (kernel families follow the U-Net
    level only)."""
    cfg = NET_CASES[name]
    m, _ = make_model(cfg, dev)
    eng = m.dnn.engine(torch.device(dev))
    g = torch.Generator().manual_seed(3)
    xs = [(torch.randn(2, 256, T, dtype=torch.complex64, generator=g) * 0.3).to(dev) for T in frames]
    t = torch.tensor([0.9, 0.4, 0.1, 0.6, 0.25][:len(frames)], device=dev)
    singles = [eng.forward(x[None], t[i:i + 1])[0] for i, x in enumerate(xs)]
    for a, b in zip(singles, eng.forward_ragged(xs, t)):
        assert a.shape == b.shape and torch.equal(a, b)
    if not quick:
        perm = list(range(len(frames)))[::-1]                  # any order, any neighbours
        for a, b in zip([singles[i] for i in perm], eng.forward_ragged([xs[i] for i in perm], t[perm])):
            assert torch.equal(a, b)
    assert torch.equal(eng.forward(xs[-1][None], t[-1:])[0], singles[-1])   # and back to uniform batches
    if not sampler:
        return
    ys = [synth.synth_spec(1, 256, T, seed=3 + i)[0].to(dev) for i, T in enumerate(frames)]
    ids = [7 + i for i in range(len(frames))]
    N = 1 if quick else 2
    makers = [lambda y, st: m.get_pc_sampler("reverse_diffusion", "ald", y, N=N, snr=0.5, seed=5, streams=st),
              lambda y, st: m.get_ode_sampler(y, N=N, seed=5, streams=st)]
    if not quick:
        makers.append(lambda y, st: m.get_pc_sampler("none", "ald", y, N=N, snr=0.5, seed=5, streams=st))
    for make in makers:
        one_by_one = [make(y[None], [ids[i]])()[0][0] for i, y in enumerate(ys)]
        together, nfe = make(ys, ids)()
        assert len(together) == len(ys) and nfe > 0
        for a, b in zip(one_by_one, together):
            assert a.shape == b.shape and torch.equal(a, b)
    with pytest.raises(RuntimeError, match="Langevin"):
        m.get_pc_sampler("reverse_diffusion", "langevin", ys, N=1, snr=0.5, seed=5)()
    with pytest.raises(TypeError):
        m.get_pc_sampler("reverse_diffusion", "ald", ys, N=1, snr=0.5, seed=5, force_python_loop=True)


def check_graph_update_path(dev, name="fwd_nf32"):
    """The captured sampler step under changing ragged compositions and batch sizes: a new composition of the same batch size must
    bring the instantiated graph up to date IN PLACE (hipGraphExecUpdate: graph_updates grows, or -- if the runtime refuses -- a
    fresh instantiation: graph_captures grows), never replay a stale executable; every output must equal the eager (graph-free) run
    of the same call bit for bit, which reads none of the buffers a stale graph would still point at."""
    cfg = NET_CASES[name]
    m, _ = make_model(cfg, dev)
    eng = m.dnn.engine(torch.device(dev))
    comps = [(128, 64), (64, 192), (192, 64), (128, 128, 64)]
    c0, u0 = eng.graph_captures(), eng.graph_updates()
    hist = []
    for frames in comps:
        ys = [synth.synth_spec(1, 256, T, seed=11 + i)[0].to(dev) for i, T in enumerate(frames)]
        ids = [3 + i for i in range(len(frames))]
        with_graph, _ = m.get_pc_sampler("reverse_diffusion", "ald", ys, N=2, snr=0.5, seed=9, streams=ids, use_graph=True)()
        hist.append((eng.graph_captures() - c0, eng.graph_updates() - u0))
        eager, _ = m.get_pc_sampler("reverse_diffusion", "ald", ys, N=2, snr=0.5, seed=9, streams=ids, use_graph=False)()
        for a, b in zip(with_graph, eager):
            assert torch.equal(a, b), frames
    print(f"captured step under {len(comps)} ragged compositions on {dev}: (captures, in-place updates) after each = {hist}")
    assert hist[0][0] == 1                                                       # the first composition captures
    for prev, cur in zip(hist, hist[1:]):
        assert cur[0] + cur[1] == prev[0] + prev[1] + 1, hist                    # every new composition re-captures: update or new instance


def check_ragged_variants(dev):
    """Ragged batches through the other two network variants: ncsnpp_v2 with the new-code score wrapper (the exit kernel reads
    x_t of the packed sampler state) and ncsnpp_48k (no pyramids, final convolution), on a 64-bin network."""
    frames = [128, 64]
    for cfg, wrap in ((NO.NetCfg.for_variant("ncsnpp_v2", nf=32, image_size=64),
                       dict(loss_type="denoiser", network_scaling="1/t", c_in="edm", c_out="edm", c_skip="edm", sigma_data=0.1)),
                      (NO.NetCfg.for_variant("ncsnpp_48k", nf=32, image_size=64), {})):
        m, _ = make_model(cfg, dev, **wrap)
        ys = [synth.synth_spec(1, 64, T, seed=3 + i)[0].to(dev) for i, T in enumerate(frames)]
        one = [m.get_pc_sampler("reverse_diffusion", "ald", y[None], N=1, snr=0.5, seed=5, streams=[4 + i])()[0][0] for i, y in enumerate(ys)]
        together, _ = m.get_pc_sampler("reverse_diffusion", "ald", ys, N=1, snr=0.5, seed=5, streams=[4, 5])()
        chunked, ns = m.get_pc_sampler("reverse_diffusion", "ald", ys, N=1, snr=0.5, seed=5, streams=[4, 5], minibatch=1)()
        assert ns == [2, 2]
        for a, b, c in zip(one, together, chunked):
            assert torch.equal(a, b) and torch.equal(a, c)
    # waveforms of different lengths through ScoreModel.enhance_batch (STFT, pad_spec, ragged sampler, iSTFT per utterance)
    m, _ = make_model(NET_CASES["fwd_nf32"], dev)
    g = torch.Generator().manual_seed(2)
    waves = [torch.randn(L, generator=g) for L in (3000, 9000)]
    alone = [m.enhance_batch(w[None], N=1, seed=4, streams=[5 + i])[0][0] for i, w in enumerate(waves)]
    mixed, nfe = m.enhance_batch(waves, N=1, seed=4, streams=[5, 6])
    assert nfe == 2 and all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(alone, mixed))
