"""Oracle restatement of the signal front-end (TEST INFRASTRUCTURE ONLY).

Follows sgmse/data_module.py:13-19 (window), :162-188 (spec_fwd / spec_back),
:190-218 (stft / istft kwargs), sgmse/util/other.py:76-90 (pad_spec) and the
per-file pipeline of enhancement.py:62-99 / ScoreModel.enhance (model.py:426-465).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class FrontCfg:
    n_fft: int = 510
    hop_length: int = 128
    window: str = "hann"
    spec_factor: float = 0.15
    spec_abs_exponent: float = 0.5
    transform_type: str = "exponent"
    sr: int = 16000

    @staticmethod
    def ears_48k():
        """Flags of the EARS-WHAM 48 kHz model (README.md:89)."""
        return FrontCfg(n_fft=1534, hop_length=384, spec_factor=0.065, spec_abs_exponent=0.667, sr=48000)


def get_window(kind: str, n: int) -> torch.Tensor:
    if kind == "sqrthann":
        return torch.sqrt(torch.hann_window(n, periodic=True))
    if kind == "hann":
        return torch.hann_window(n, periodic=True)
    raise NotImplementedError(f"Window type {kind} not implemented!")


def stft(sig: torch.Tensor, fc: FrontCfg) -> torch.Tensor:
    """data_module.py:212-214: torch.stft(center=True, reflect pad, onesided, no normalisation)."""
    return torch.stft(sig, n_fft=fc.n_fft, hop_length=fc.hop_length, window=get_window(fc.window, fc.n_fft),
                      center=True, return_complex=True)


def istft(spec: torch.Tensor, fc: FrontCfg, length=None) -> torch.Tensor:
    """data_module.py:216-218."""
    return torch.istft(spec, n_fft=fc.n_fft, hop_length=fc.hop_length, window=get_window(fc.window, fc.n_fft),
                       center=True, length=length)


def stft_manual(sig: torch.Tensor, fc: FrontCfg) -> torch.Tensor:
    """Explicit DFT form of the same transform: reflect-pad by
    n_fft//2, frame k = padded[k*hop : k*hop+n_fft]*w, X[f,k] = sum_n frame[n] e^{-2 pi i f n / n_fft}.
    Evaluated in float64 and rounded, as an independent check of ``stft``."""
    n, hop = fc.n_fft, fc.hop_length
    w = get_window(fc.window, n).double()
    x = F.pad(sig[:, None].double(), (n // 2, n // 2), mode="reflect")[:, 0]
    frames = x.unfold(-1, n, hop) * w                                   # [B, K, n]
    f = torch.arange(n // 2 + 1, dtype=torch.float64)
    nn_ = torch.arange(n, dtype=torch.float64)
    ang = -2 * torch.pi * f[:, None] * nn_[None, :] / n
    re = torch.einsum("bkn,fn->bfk", frames, torch.cos(ang))
    im = torch.einsum("bkn,fn->bfk", frames, torch.sin(ang))
    return torch.complex(re, im).to(torch.complex64)


def istft_manual(spec: torch.Tensor, fc: FrontCfg, length: int) -> torch.Tensor:
    """irfft * window, overlap-add, divide by sum of w^2, drop n_fft//2, crop (Appendix C)."""
    n, hop = fc.n_fft, fc.hop_length
    w = get_window(fc.window, n).double()
    B, Fq, K = spec.shape
    fr = torch.fft.irfft(spec.to(torch.complex128), n=n, dim=1) * w[None, :, None]   # [B, n, K]
    total = n + hop * (K - 1)
    out = torch.zeros(B, total, dtype=torch.float64)
    env = torch.zeros(total, dtype=torch.float64)
    for k in range(K):
        out[:, k * hop:k * hop + n] += fr[:, :, k]
        env[k * hop:k * hop + n] += w * w
    y = out[:, n // 2:] / env[n // 2:]
    return y[:, :length].to(torch.float32)


def spec_fwd(spec: torch.Tensor, fc: FrontCfg) -> torch.Tensor:
    """data_module.py:162-175."""
    if fc.transform_type == "exponent":
        if fc.spec_abs_exponent != 1:
            e = fc.spec_abs_exponent
            spec = spec.abs() ** e * torch.exp(1j * spec.angle())
        return spec * fc.spec_factor
    if fc.transform_type == "log":
        return torch.log(1 + spec.abs()) * torch.exp(1j * spec.angle()) * fc.spec_factor
    return spec


def spec_back(spec: torch.Tensor, fc: FrontCfg) -> torch.Tensor:
    """data_module.py:177-188."""
    if fc.transform_type == "exponent":
        spec = spec / fc.spec_factor
        if fc.spec_abs_exponent != 1:
            e = fc.spec_abs_exponent
            spec = spec.abs() ** (1 / e) * torch.exp(1j * spec.angle())
        return spec
    if fc.transform_type == "log":
        spec = spec / fc.spec_factor
        return (torch.exp(spec.abs()) - 1) * torch.exp(1j * spec.angle())
    return spec


def pad_spec(Y: torch.Tensor, mode: str = "zero_pad") -> torch.Tensor:
    """util/other.py:76-90: right-pad the frame axis to a multiple of 64."""
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    if mode == "zero_pad":
        return F.pad(Y, (0, num_pad, 0, 0))
    if mode == "reflection":
        re = F.pad(Y.real, (0, num_pad, 0, 0), mode="reflect")
        im = F.pad(Y.imag, (0, num_pad, 0, 0), mode="reflect")
        return torch.complex(re, im)
    if mode == "replication":
        re = F.pad(Y.real, (0, num_pad, 0, 0), mode="replicate")
        im = F.pad(Y.imag, (0, num_pad, 0, 0), mode="replicate")
        return torch.complex(re, im)
    raise NotImplementedError("This function hasn't been implemented yet.")


def enhance(y: torch.Tensor, fc: FrontCfg, sampler, pad_mode: str = "zero_pad"):
    """enhancement.py:62-99: y float32 [1, L] -> enhanced float32 [L].
    ``sampler(Y)`` maps the padded spectrogram [1,1,F,T] to the sample [1,1,F,T]."""
    T_orig = y.size(1)
    norm = y.abs().max()
    y = y / norm
    Y = torch.unsqueeze(spec_fwd(stft(y, fc), fc), 0)
    Y = pad_spec(Y, pad_mode)
    sample = sampler(Y)
    x_hat = istft(spec_back(sample.squeeze(), fc), fc, T_orig)
    return x_hat * norm
