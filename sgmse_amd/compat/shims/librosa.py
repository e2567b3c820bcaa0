"""Stand-in for ``librosa.resample`` (reference enhancement.py:9,65) on machines without librosa: polyphase resampling with
``scipy.signal.resample_poly`` (librosa's default is soxr_hq: same band-limited interpolation up to filter design, not bit-equal).
Only reached when a file's rate differs from the model's."""
from math import gcd

import numpy as np
from scipy.signal import resample_poly


def resample(y, *, orig_sr, target_sr, axis=-1, **_ignored):
    g = gcd(int(orig_sr), int(target_sr))
    return resample_poly(np.asarray(y), int(target_sr) // g, int(orig_sr) // g, axis=axis).astype(np.asarray(y).dtype)
