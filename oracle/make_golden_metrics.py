#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/metrics.npz: outputs of the reference's own NumPy metric functions
(/root/reference/sgmse/util/other.py:11-74,107-111) on seeded signals, for the f4 row (evaluation side).

The reference module imports `pesq` and `pystoi` at module scope (other.py:7-8); both are absent here and are not
needed by the functions recorded, so empty stand-in modules are registered before the file is loaded by path (the
`sgmse` package itself is not imported).  Run in the build container only; the fixture is what travels."""
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("SGMSE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference_other():
    for name, fn in (("pesq", "pesq"), ("pystoi", "stoi")):
        m = types.ModuleType(name)
        setattr(m, fn, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stand-in")))
        sys.modules.setdefault(name, m)
    spec = importlib.util.spec_from_file_location("_ref_other", os.path.join(REF, "sgmse", "util", "other.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_other()
    rng = np.random.default_rng(123)
    L = 4000
    t = np.arange(L) / 16000.0
    s = np.sin(2 * np.pi * 220 * t) * np.hanning(L) + 0.1 * rng.standard_normal(L)       # "clean"
    n = 0.3 * rng.standard_normal(L)                                                       # "noise"
    s_hat = 0.8 * s + 0.05 * n + 0.02 * rng.standard_normal(L)                             # "estimate"
    comps = ref.si_sdr_components(s_hat, s, n)
    vals = rng.standard_normal(25) * 2 + 10
    with_nan = vals.copy(); with_nan[[3, 7]] = np.nan
    np.savez_compressed(
        os.path.join(OUT, "metrics.npz"), s=s, n=n, s_hat=s_hat,
        s_target=comps[0], e_noise=comps[1], e_art=comps[2],
        energy_ratios=np.array(ref.energy_ratios(s_hat, s, n)),
        si_sdr=np.array(ref.si_sdr(s, s_hat)), snr_dB=np.array(ref.snr_dB(s, n)),
        hp=ref.hp_filter(s_hat), hp_48k=ref.hp_filter(s_hat, cut_off=120, order=6, sr=48000),
        vals=vals, mean_conf_int=np.array(ref.mean_conf_int(vals)), mean_conf_int_90=np.array(ref.mean_conf_int(vals, 0.9)),
        with_nan=with_nan, mean_std=np.array(ref.mean_std(with_nan)))
    m = ref.Method("m", "/x", ["a", "b"])
    for v in vals:
        m.append("a", v)
    assert np.allclose(m.get_mean_ci("a"), ref.mean_conf_int(vals))
    assert ref.print_mean_std(with_nan) == f"{np.nanmean(with_nan):.2f} ± {np.nanstd(with_nan):.2f}"
    print("metrics.npz written; energy_ratios =", ref.energy_ratios(s_hat, s, n))


if __name__ == "__main__":
    main()
