"""Front-end fixtures from the REFERENCE's own SpecsDataModule (sgmse/data_module.py:13-19, 162-218): TEST INFRASTRUCTURE ONLY.

Runs in the build container (needs /root/reference): imports the reference's data module -- with empty stand-ins for the two packages
its import line needs and this image lacks (pytorch_lightning, torchaudio; neither is touched by the functions called here) -- and stores
stft / spec_fwd / spec_back / istft outputs for the three transform types and both windows on a seeded waveform:
tests/golden/front.npz.  tests/test_oracle_golden.py pins oracle/stft_oracle.py against it, the emulator and GPU tests pin the HIP kernels.

    python oracle/make_golden_front.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SGMSE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import synth                                     # noqa: E402

CASES = {   # name -> (n_fft, hop, window, transform_type, spec_factor, spec_abs_exponent, L)
    "hann_exponent": (510, 128, "hann", "exponent", 0.15, 0.5, 6000),
    "sqrthann_log": (510, 128, "sqrthann", "log", 0.15, 0.5, 6000),
    "hann_none": (510, 128, "hann", "none", 0.15, 0.5, 6000),
    "sqrthann_exponent_48k": (1534, 384, "sqrthann", "exponent", 0.065, 0.667, 9000),
}


def reference_module():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningDataModule = type("LightningDataModule", (), {"__init__": lambda self, *a, **k: None})
    ta = types.ModuleType("torchaudio")
    ta.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchaudio stand-in: not available here"))
    sys.modules.setdefault("pytorch_lightning", pl)
    sys.modules.setdefault("torchaudio", ta)
    sys.path.insert(0, REF)
    from sgmse.data_module import SpecsDataModule
    return SpecsDataModule


def main():
    SpecsDataModule = reference_module()
    out = {}
    for name, (n_fft, hop, window, ttype, factor, expo, L) in CASES.items():
        dm = SpecsDataModule(base_dir="", n_fft=n_fft, hop_length=hop, window=window, spec_factor=factor, spec_abs_exponent=expo,
                             transform_type=ttype, gpu=False)
        sig = synth.synth_waveform(L, seed=5, batch=2)
        S = dm.stft(sig)
        Y = dm.spec_fwd(S)
        out[name + "/stft"] = S.numpy()
        out[name + "/fwd"] = Y.numpy()
        out[name + "/back"] = dm.spec_back(Y).numpy()
        out[name + "/istft"] = dm.istft(S, L).numpy()
        print(f"{name}: stft {tuple(S.shape)}, |fwd| max {Y.abs().max():.4f}, round trip {float((dm.istft(S, L) - sig).norm() / sig.norm()):.2e}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "front.npz"), **out)


if __name__ == "__main__":
    main()
