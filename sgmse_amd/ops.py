"""Functional, tensor-level access to the HIP kernels (one layer each).  Used by the host mirror of the reference
API and by the per-op parity tests; every function enqueues on the current stream of the tensors' device."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib

_ctx: Dict[str, "_lib.Context"] = {}


def default_context(device: "torch.device | str | None" = None) -> "_lib.Context":
    """A shared weight-less context per device (ops only need its stream and scratch)."""
    if device is None:
        device = "cpu" if _lib.is_emulator() else torch.device("cuda", torch.cuda.current_device())
    key = str(torch.device(device))
    if key not in _ctx:
        _ctx[key] = _lib.Context(device)
    return _ctx[key]


def _f32(t: Optional[torch.Tensor], name: str, dev) -> Optional[torch.Tensor]:
    return None if t is None else _lib.check_tensor(t, name, torch.float32, dev)


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
           residual: Optional[torch.Tensor] = None, out_scale: float = 1.0, x2: Optional[torch.Tensor] = None,
           in_scale: Optional[torch.Tensor] = None, in_shift: Optional[torch.Tensor] = None, in_act: bool = False,
           force_direct: bool = False, force_split: Optional[str] = None) -> torch.Tensor:
    """out = (conv2d(cat[x, x2], weight, padding=k//2) + bias + residual) * out_scale, k in {1, 3};
    optional fused per-(b,c) affine (+SiLU) on the input (reference layers.py:100-124)."""
    ctx = default_context(x.device)
    dev = ctx.device
    x = _f32(x, "x", dev); weight = _f32(weight, "weight", dev)
    B, C1, H, W = x.shape
    C2 = 0 if x2 is None else x2.shape[1]
    Cout, Cin, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError("kernel must be 1x1 or 3x3")
    if Cin != C1 + C2:
        raise ValueError(f"weight expects {Cin} input channels, got {C1}+{C2}")
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_op_conv2d(ctx.h, x.data_ptr(), weight.data_ptr(), _lib.ptr(_f32(bias, "bias", dev)),
                                      _lib.ptr(_f32(residual, "residual", dev)), out.data_ptr(), B, Cin, Cout, H, W, kh,
                                      float(out_scale), {None: int(force_direct), 'bf16x3': 2, 'fp16x2': 3, 'wino': 4, 'wino4': 5, 'thin': 6, 'wino2d': 7}[force_split], _lib.ptr(_f32(in_scale, "in_scale", dev)),
                                      _lib.ptr(_f32(in_shift, "in_shift", dev)), int(in_act),
                                      _lib.ptr(_f32(x2, "x2", dev)), C2))
    return out


def group_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, *, act: bool = False,
               x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(GroupNorm(min(C//4,32), C, eps=1e-6)(cat[x, x2])) (reference layerspp.py:219,243)."""
    ctx = default_context(x.device)
    dev = ctx.device
    x = _f32(x, "x", dev)
    B, C1, H, W = x.shape
    C2 = 0 if x2 is None else x2.shape[1]
    C = C1 + C2
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_op_groupnorm(ctx.h, x.data_ptr(), _f32(weight, "weight", dev).data_ptr(),
                                         _f32(bias, "bias", dev).data_ptr(), out.data_ptr(), B, C, H, W, int(act),
                                         _lib.ptr(_f32(x2, "x2", dev)), C2))
    return out


def fir_resample(x: torch.Tensor, up: bool, in_scale: Optional[torch.Tensor] = None, in_shift: Optional[torch.Tensor] = None,
                 in_act: bool = False, return_raw: bool = False):
    """upsample_2d / downsample_2d with k=(1,3,3,1), factor 2 (reference up_or_down_sampling.py:195-257).

    As fused inside the network's residual blocks (layerspp.py:243-262): with ``in_scale``/``in_shift`` ([B, C], the
    folded GroupNorm affine) the resampler consumes ``act(x * in_scale + in_shift)`` without materialising it, and with
    ``return_raw`` it also returns the resampled ``x`` itself (the shortcut branch) from the same pass."""
    ctx = default_context(x.device)
    x = _f32(x, "x", ctx.device)
    B, Cc, H, W = x.shape
    shape = (B, Cc, 2 * H, 2 * W) if up else (B, Cc, H // 2, W // 2)
    out = torch.empty(shape, dtype=torch.float32, device=ctx.device)
    raw = torch.empty(shape, dtype=torch.float32, device=ctx.device) if return_raw else None
    sc = _f32(in_scale, "in_scale", ctx.device)
    sh = _f32(in_shift, "in_shift", ctx.device)
    if (sc is None) != (sh is None) or (sc is not None and (tuple(sc.shape) != (B, Cc) or tuple(sh.shape) != (B, Cc))):
        raise ValueError("in_scale and in_shift must both be given, with shape [B, C]")
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_op_fir(ctx.h, x.data_ptr(), out.data_ptr(), B * Cc, H, W, int(up), _lib.ptr(sc), _lib.ptr(sh),
                                   int(bool(in_act)), _lib.ptr(raw)))
    return (out, raw) if return_raw else out


def upfirdn2d(x: torch.Tensor, kernel: torch.Tensor, up: int = 1, down: int = 1, pad: Tuple[int, int] = (0, 0)) -> torch.Tensor:
    """upfirdn2d(input[N,C,H,W], kernel[kh,kw], up, down, pad) with the same pad on both axes
    (reference op/upfirdn2d.py:148-159).  float32, float64 and float16 inputs run in their own element type like the reference op
    (op/upfirdn2d_kernel.cu:311 dispatches over exactly these three; half accumulates in fp32 here, in half there); any other element
    type raises like the reference's dispatch macro does, and the kernel must live on the input's device (upfirdn2d.cpp:8,15-16: both
    tensors are checked, neither is moved); only the kernel's element type is cast to the input's."""
    ctx = default_context(x.device)
    code = {torch.float32: 0, torch.float64: 1, torch.float16: 2}.get(x.dtype)
    if code is None:
        raise ValueError(f"upfirdn2d: input dtype {x.dtype} is not supported (float32, float64, float16 -- op/upfirdn2d_kernel.cu:311)")
    if not kernel.dtype.is_floating_point:
        raise ValueError(f"upfirdn2d: kernel dtype {kernel.dtype} is not a floating-point type")
    if kernel.device != x.device:
        raise ValueError(f"upfirdn2d: kernel on {kernel.device}, input on {x.device} (the reference op checks both tensors, upfirdn2d.cpp:15-16)")
    dt = x.dtype
    x = _lib.check_tensor(x, "input", dt, ctx.device); kernel = kernel.to(dtype=dt).contiguous()
    N, Cc, H, W = x.shape
    kh, kw = kernel.shape
    Ho = (H * up + pad[0] + pad[1] - kh) // down + 1
    Wo = (W * up + pad[0] + pad[1] - kw) // down + 1
    out = torch.empty((N, Cc, Ho, Wo), dtype=dt, device=ctx.device)
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_upfirdn2d_dtype(ctx.h, code, x.data_ptr(), kernel.data_ptr(), out.data_ptr(), N * Cc, H, W, kh, kw, up, up,
                                            down, down, pad[0], pad[1], pad[0], pad[1]))
    return out


def attention(qkv: torch.Tensor) -> torch.Tensor:
    """softmax(q^T k / sqrt(C)) v for qkv [B, 3C, S] -> [B, C, S] (reference layerspp.py:82-88)."""
    ctx = default_context(qkv.device)
    qkv = _f32(qkv, "qkv", ctx.device)
    B, C3, S = qkv.shape
    out = torch.empty((B, C3 // 3, S), dtype=torch.float32, device=ctx.device)
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_op_attention(ctx.h, qkv.data_ptr(), out.data_ptr(), B, C3 // 3, S))
    return out


def stft(sig: torch.Tensor, n_fft: int, hop_length: int, window: torch.Tensor) -> torch.Tensor:
    """torch.stft(sig, n_fft, hop_length, window=window, center=True, return_complex=True) (data_module.py:212-214)."""
    ctx = default_context(sig.device)
    squeeze = sig.dim() == 1
    s = _f32(sig.reshape(-1, sig.shape[-1]), "sig", ctx.device)
    B, L = s.shape
    K = L // hop_length + 1 if n_fft % 2 == 0 else (L - 1) // hop_length + 1
    out = torch.empty((B, n_fft // 2 + 1, K), dtype=torch.complex64, device=ctx.device)
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_stft(ctx.h, s.data_ptr(), _f32(window, "window", ctx.device).data_ptr(), out.data_ptr(), B, L,
                                 n_fft, hop_length))
    return out[0] if squeeze else out.reshape(*sig.shape[:-1], n_fft // 2 + 1, K)


def istft(spec: torch.Tensor, n_fft: int, hop_length: int, window: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:
    """torch.istft(spec, n_fft, hop_length, window=window, center=True, length=length) (data_module.py:216-218)."""
    ctx = default_context(spec.device)
    squeeze = spec.dim() == 2
    s = _lib.check_tensor(spec.reshape(-1, spec.shape[-2], spec.shape[-1]), "spec", torch.complex64, ctx.device)
    B, Fq, K = s.shape
    if Fq != n_fft // 2 + 1:
        raise ValueError("spec has the wrong number of frequency bins for n_fft")
    if length is None:
        length = hop_length * (K - 1)
    out = torch.empty((B, length), dtype=torch.float32, device=ctx.device)
    ctx.use_current_stream()
    ctx.check(ctx.lib.sgmse_istft(ctx.h, s.data_ptr(), _f32(window, "window", ctx.device).data_ptr(), out.data_ptr(), B, K,
                                  n_fft, hop_length, length))
    return out[0] if squeeze else out.reshape(*spec.shape[:-2], length)


_XF = {"exponent": 0, "log": 1, "none": 2}


def spec_transform(spec: torch.Tensor, transform_type: str, factor: float, exponent: float, inverse: bool) -> torch.Tensor:
    """spec_fwd / spec_back (data_module.py:162-188)."""
    if transform_type not in _XF:
        raise NotImplementedError(f"transform_type {transform_type!r}")
    ctx = default_context(spec.device)
    s = _lib.check_tensor(spec, "spec", torch.complex64, ctx.device)
    out = torch.empty_like(s)
    ctx.use_current_stream()
    fn = ctx.lib.sgmse_spec_back if inverse else ctx.lib.sgmse_spec_fwd
    ctx.check(fn(ctx.h, s.data_ptr(), out.data_ptr(), s.numel(), _XF[transform_type], float(factor), float(exponent)))
    return out
