"""Evaluation of an enhanced directory against clean / noisy references (reference calc_metrics.py:15-67):
SI-SDR / SI-SIR / SI-SAR per file (NumPy), PESQ / ESTOI when the `pesq` / `pystoi` packages are installed (NaN
otherwise), `_results.csv` + `_avg_results.txt` written into the enhanced directory.

    python -m sgmse_amd.calc_metrics --clean_dir C --noisy_dir N --enhanced_dir E

Audio I/O uses `soundfile` when present and scipy.io.wavfile (PCM16/32, float32 wav) otherwise."""
import csv
from argparse import ArgumentParser
from glob import glob
from os.path import join

import numpy as np

from .util.other import _perceptual_metrics, energy_ratios, mean_std


def read_wav(path):
    """(float64 samples in [-1, 1], sample rate) like soundfile.read."""
    try:
        from soundfile import read
        return read(path)
    except ImportError:
        from scipy.io import wavfile
        sr, x = wavfile.read(path)
        if x.dtype.kind == "i":
            x = x.astype(np.float64) / float(2 ** (8 * x.dtype.itemsize - 1))
        elif x.dtype.kind == "u":       # 8-bit PCM is unsigned
            x = (x.astype(np.float64) - 128.0) / 128.0
        return x.astype(np.float64), sr


def resample_to(x, sr, target):
    if sr == target:
        return x
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr), int(target))
    return resample_poly(x, target // g, sr // g)


def evaluate(clean_dir, noisy_dir, enhanced_dir):
    pesq, stoi = _perceptual_metrics()
    data = {"filename": [], "pesq": [], "estoi": [], "si_sdr": [], "si_sir": [], "si_sar": []}
    noisy_files = sorted(glob(join(noisy_dir, "*.wav"))) + sorted(glob(join(noisy_dir, "**", "*.wav")))
    for noisy_file in noisy_files:
        filename = noisy_file.replace(noisy_dir, "")[1:]
        clean_filename = filename.split("_")[0] + ".wav" if "dB" in filename else filename
        x, sr_x = read_wav(join(clean_dir, clean_filename))
        y, sr_y = read_wav(join(noisy_dir, filename))
        x_hat, sr_x_hat = read_wav(join(enhanced_dir, filename))
        assert sr_x == sr_y == sr_x_hat
        n = y - x
        data["filename"].append(filename)
        if pesq:
            data["pesq"].append(pesq(16000, resample_to(x, sr_x, 16000), resample_to(x_hat, sr_x_hat, 16000), "wb"))
            data["estoi"].append(stoi(x, x_hat, sr_x, extended=True))
        else:
            data["pesq"].append(float("nan"))
            data["estoi"].append(float("nan"))
        sdr, sir, sar = energy_ratios(x_hat, x, n)
        data["si_sdr"].append(sdr); data["si_sir"].append(sir); data["si_sar"].append(sar)
    return data


def summary_lines(data):
    fmt = {"pesq": ("PESQ", 2), "estoi": ("ESTOI", 2), "si_sdr": ("SI-SDR", 1), "si_sir": ("SI-SIR", 1), "si_sar": ("SI-SAR", 1)}
    lines = []
    for key, (label, dec) in fmt.items():
        v = np.asarray(data[key], dtype=float)
        if np.all(np.isnan(v)):
            lines.append(f"{label}: n/a (package not installed)")
        else:
            m, s = mean_std(v)
            lines.append(f"{label}: {m:.{dec}f} ± {s:.{dec}f}")
    return lines


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument("--clean_dir", type=str, required=True, help="Directory containing the clean data")
    parser.add_argument("--noisy_dir", type=str, required=True, help="Directory containing the noisy data")
    parser.add_argument("--enhanced_dir", type=str, required=True, help="Directory containing the enhanced data")
    args = parser.parse_args(argv)
    data = evaluate(args.clean_dir, args.noisy_dir, args.enhanced_dir)
    lines = summary_lines(data)
    print("\n".join(lines))
    with open(join(args.enhanced_dir, "_avg_results.txt"), "w") as log:
        log.write("\n".join(lines) + "\n")
    with open(join(args.enhanced_dir, "_results.csv"), "w", newline="") as f:
        w = csv.writer(f)
        keys = list(data.keys())
        w.writerow(keys)
        for row in zip(*(data[k] for k in keys)):
            w.writerow(row)
    return data


if __name__ == "__main__":
    main()
