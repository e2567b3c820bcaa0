"""Stand-in for ``torchaudio.load`` (reference enhancement.py:6,61) on machines without torchaudio: wav files through
``scipy.io.wavfile``, flac files through sgmse_amd/util/flac.py, returned like torchaudio does -- float32 tensor [channels, frames]
in [-1, 1] and the sample rate."""
import numpy as np
import torch
from scipy.io import wavfile


def load(filepath, **_ignored):
    from sgmse_amd.util.flac import is_flac, read_flac
    if is_flac(filepath):
        xi, sr, bits = read_flac(filepath)
        return torch.from_numpy(np.ascontiguousarray((xi.astype(np.float32) / float(2 ** (bits - 1))).T)), int(sr)
    sr, x = wavfile.read(filepath)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = np.asarray(x, dtype=np.float32)
    x = x[None, :] if x.ndim == 1 else x.T
    return torch.from_numpy(np.ascontiguousarray(x)), int(sr)
