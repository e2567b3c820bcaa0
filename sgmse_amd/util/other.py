"""The two helpers of reference sgmse/util/other.py that sit on the enhancement path."""
import torch
import torch.nn.functional as F


def pad_spec(Y: torch.Tensor, mode: str = "zero_pad") -> torch.Tensor:
    """Right-pad the frame axis of Y [B,1,F,T] to the next multiple of 64 (reference util/other.py:76-90).
    Pure data movement on the tensor's device; raises NotImplementedError for unknown modes like the reference."""
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    if mode == "zero_pad":
        return F.pad(Y, (0, num_pad, 0, 0))
    if mode in ("reflection", "replication"):
        m = "reflect" if mode == "reflection" else "replicate"
        if Y.is_complex():
            return torch.complex(F.pad(Y.real, (0, num_pad, 0, 0), mode=m), F.pad(Y.imag, (0, num_pad, 0, 0), mode=m))
        return F.pad(Y, (0, num_pad, 0, 0), mode=m)
    raise NotImplementedError("This function hasn't been implemented yet.")


def set_torch_cuda_arch_list():
    """Reference util/other.py:126-141 sets TORCH_CUDA_ARCH_LIST for its import-time JIT build of the CUDA op.
    The HIP library is built ahead of time for gfx950, so there is nothing to set; kept for enhancement.py:12-13."""
    if torch.cuda.is_available():
        print(f"sgmse_amd: using the prebuilt gfx950 HIP library on {torch.cuda.get_device_name(0)}")


# ---- evaluation side: the NumPy metrics of reference util/other.py:11-74,92-124 -------------------------
# Host-side float64 NumPy on 1-D waveforms; nothing here touches the GPU path.

import os as _os

import numpy as _np


def si_sdr_components(s_hat, s, n):
    """Projection of the estimate onto clean speech ``s`` and noise ``n`` (reference util/other.py:11-24):
    returns (s_target, e_noise, e_art) with s_hat = s_target + e_noise + e_art."""
    s_hat, s, n = (_np.asarray(v) for v in (s_hat, s, n))
    s_target = (_np.dot(s_hat, s) / _np.linalg.norm(s) ** 2) * s
    e_noise = (_np.dot(s_hat, n) / _np.linalg.norm(n) ** 2) * n
    return s_target, e_noise, s_hat - s_target - e_noise


def energy_ratios(s_hat, s, n):
    """(SI-SDR, SI-SIR, SI-SAR) in dB (reference util/other.py:26-33)."""
    s_target, e_noise, e_art = si_sdr_components(s_hat, s, n)
    p = _np.linalg.norm(s_target) ** 2
    return (10 * _np.log10(p / _np.linalg.norm(e_noise + e_art) ** 2),
            10 * _np.log10(p / _np.linalg.norm(e_noise) ** 2),
            10 * _np.log10(p / _np.linalg.norm(e_art) ** 2))


def mean_conf_int(data, confidence=0.95):
    """Mean and half-width of the Student-t confidence interval (reference util/other.py:35-40)."""
    import scipy.stats
    a = 1.0 * _np.array(data)
    h = scipy.stats.sem(a) * scipy.stats.t.ppf((1 + confidence) / 2.0, len(a) - 1)
    return _np.mean(a), h


class Method:
    """Per-method metric accumulator (reference util/other.py:42-58)."""

    def __init__(self, name, base_dir, metrics):
        self.name, self.base_dir = name, base_dir
        self.metrics = {m: [] for m in metrics}

    def append(self, matric, value):
        self.metrics[matric].append(value)

    def get_mean_ci(self, metric):
        return mean_conf_int(_np.array(self.metrics[metric]))


def hp_filter(signal, cut_off=80, order=10, sr=16000):
    """Butterworth high-pass as second-order sections (reference util/other.py:60-64)."""
    from scipy.signal import butter, sosfilt
    return sosfilt(butter(order, cut_off / sr * 2, "hp", output="sos"), signal)


def si_sdr(s, s_hat):
    """Scale-invariant SDR in dB; note the argument order (reference, estimate) (reference util/other.py:66-70)."""
    s, s_hat = _np.asarray(s), _np.asarray(s_hat)
    alpha = _np.dot(s_hat, s) / _np.linalg.norm(s) ** 2
    return 10 * _np.log10(_np.linalg.norm(alpha * s) ** 2 / _np.linalg.norm(alpha * s - s_hat) ** 2)


def snr_dB(s, n):
    """Power ratio of two signals in dB (reference util/other.py:72-76)."""
    s, n = _np.asarray(s), _np.asarray(n)
    return 10 * _np.log10((_np.sum(s ** 2) / len(s)) / (_np.sum(n ** 2) / len(n)))


def ensure_dir(file_path):
    if not _os.path.exists(file_path):
        _os.makedirs(file_path)


def mean_std(data):
    """NaN-ignoring mean and (population) standard deviation (reference util/other.py:107-111)."""
    data = _np.asarray(data, dtype=float)
    data = data[~_np.isnan(data)]
    return _np.mean(data), _np.std(data)


def print_mean_std(data, decimal=2):
    mean, std = mean_std(_np.array(data, dtype=float))
    if decimal == 2:
        return f"{mean:.2f} ± {std:.2f}"
    if decimal == 1:
        return f"{mean:.1f} ± {std:.1f}"
    raise ValueError("decimal must be 1 or 2")     # the reference leaves `string` unbound here (UnboundLocalError)


def _perceptual_metrics():
    """PESQ / ESTOI come from the third-party `pesq` and `pystoi` packages (reference util/other.py:7-8); they are not
    reimplemented here.  Returns (pesq, stoi) or (None, None) when the packages are not installed."""
    try:
        from pesq import pesq
        from pystoi import stoi
        return pesq, stoi
    except ImportError:
        return None, None


def print_metrics(x, y, x_hat_list, labels, sr=16000):
    """SI-SDR (always) and PESQ / ESTOI (when `pesq` / `pystoi` are installed) of the mixture and of every estimate
    (reference util/other.py:97-105)."""
    pesq, stoi = _perceptual_metrics()

    def line(tag, est):
        p = f"{pesq(sr, x, est, 'wb'):.2f}" if pesq else "n/a"
        e = f"{stoi(x, est, sr, extended=True):.2f}" if stoi else "n/a"
        print(f"{tag}  PESQ: {p}, ESTOI: {e}, SI-SDR: {si_sdr(x, est):.2f}")

    line("Mixture:", y)
    for label, x_hat in zip(labels, x_hat_list):
        line(f"{label}:", x_hat)
