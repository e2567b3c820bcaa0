"""Stand-in for the `soundfile` package, for machines that do not have it (this image): just what the reference's scripts call --
``write(path, data, samplerate)`` (enhancement.py:103) and ``read(path)`` (calc_metrics.py) -- on top of ``scipy.io.wavfile``.
Writes wav (float32 samples); reads wav and, through sgmse_amd/util/flac.py, flac.  Put `sgmse_amd/compat/shims` on PYTHONPATH only where the real package is missing."""
import os

import numpy as np
from scipy.io import wavfile


def write(file, data, samplerate, subtype=None, **_ignored):
    data = np.asarray(data)
    if data.ndim == 2 and data.shape[0] < data.shape[1]:      # soundfile wants [frames, channels]; tolerate [channels, frames]
        data = data.T
    os.makedirs(os.path.dirname(os.path.abspath(file)), exist_ok=True)
    wavfile.write(file, int(samplerate), np.ascontiguousarray(data, dtype=np.float32))


def read(file, dtype="float64", always_2d=False, **_ignored):
    from sgmse_amd.util.flac import is_flac, read_flac
    if is_flac(file):
        xi, sr, bits = read_flac(file)
        x = (xi.astype(np.float64) / float(2 ** (bits - 1))).astype(dtype)
        return (x if always_2d or x.shape[1] > 1 else x[:, 0]), int(sr)
    sr, x = wavfile.read(file)
    if x.dtype.kind == "i":
        x = x.astype(np.float64) / float(2 ** (8 * x.dtype.itemsize - 1))
    elif x.dtype.kind == "u":
        x = (x.astype(np.float64) - 128.0) / 128.0
    x = x.astype(dtype)
    if always_2d and x.ndim == 1:
        x = x[:, None]
    return x, int(sr)
