"""The oracle restatement against the fixtures produced by running the *reference itself*
(oracle/make_golden.py, build container only).  CPU-only; no reference tree needed."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from oracle import ncsnpp_oracle as NO
from oracle import sde_oracle as SO
from oracle import stft_oracle as FO
from oracle import synth


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("name,cfg", [
    ("fwd_nf32", NO.NetCfg.for_variant("ncsnpp", nf=32)),
    ("fwd_48k_nf32", NO.NetCfg.for_variant("ncsnpp_48k", nf=32)),
    ("fwd_nf128", NO.NetCfg.for_variant("ncsnpp")),
    ("fwd_v2_nf32", NO.NetCfg.for_variant("ncsnpp_v2", nf=32)),
])
def test_network_forward_matches_reference(name, cfg):
    z = load(name)
    P = synth.synth_params(cfg, seed=int(z["seed"]))
    with torch.no_grad():
        out = NO.ncsnpp_forward(P, cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["t"]))
    assert rel_l2(out, z["out"]) < 2e-5


def test_state_dict_contract():
    """647 tensors / 65,590,822 parameters for the default ncsnpp."""
    shp = NO.param_shapes(NO.NetCfg.for_variant("ncsnpp"))
    assert len(shp) == 647
    assert sum(int(np.prod(s)) for s in shp.values()) == 65_590_822
    assert list(shp)[:3] == ["output_layer.weight", "output_layer.bias", "all_modules.0.W"]
    shp48 = NO.param_shapes(NO.NetCfg.for_variant("ncsnpp_48k"))
    assert sum(int(np.prod(s)) for s in shp48.values()) == 64_739_854


@pytest.mark.parametrize("tag", ["vb", "ears"])
def test_step_table_known_answers(tag):
    z = load(f"sde_table_{tag}")
    sde = SO.OUVE(float(z["theta"]), float(z["sigma_min"]), float(z["sigma_max"]), int(z["N"]))
    tab = SO.step_table(sde, float(z["eps"]), float(z["snr"]))
    for k, v in tab.items():
        assert np.array_equal(v.numpy(), z[k]), k


def test_step_table_survey_appendix_b():
    """Known-answer values printed in SURVEY.md Appendix B (reference's own fp32 results)."""
    tab = SO.step_table(SO.OUVE(1.5, 0.05, 0.5, 30), 0.03, 0.5)
    assert abs(float(tab["std"][0]) - 0.388982654) < 1e-7
    assert abs(float(tab["dt"][29]) - 0.029999999) < 1e-8
    assert abs(float(tab["G2"][15]) - 3.820420476e-03) < 1e-9
    tab = SO.step_table(SO.OUVE(2.0, 0.1, 1.0, 50), 0.03, 0.33)
    assert abs(float(tab["ald_noise"][25]) - 0.153482258) < 1e-7


@pytest.mark.parametrize("tag", ["pc_N4", "pnone_N6", "lang_N4", "pc_N4_c2"])
def test_sampler_matches_reference(tag):
    z = load(tag)
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    P = synth.synth_params(cfg, seed=0)
    sde = SO.OUVE(1.5, 0.05, 0.5, int(z["N"]))
    out, nfe = SO.pc_sample(sde, lambda a, b, c: NO.score_fn(P, cfg, a, b, c), torch.from_numpy(z["y"]),
                            SO.NoiseReplay(int(z["noise_seed"])), eps=0.03, snr=float(z["snr"]),
                            corrector=str(z["corrector"]), predictor=str(z["predictor"]),
                            corrector_steps=int(z["corrector_steps"]) if "corrector_steps" in z.files else 1)
    assert nfe == int(z["nfe"])
    assert rel_l2(out, z["out"]) < 1e-4


def test_pfode_matches_reference_pieces():
    z = load("pfode_N6")
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    P = synth.synth_params(cfg, seed=0)
    out, nfe = SO.pc_sample(SO.OUVE(1.5, 0.05, 0.5, 6), lambda a, b, c: NO.score_fn(P, cfg, a, b, c),
                            torch.from_numpy(z["y"]), SO.NoiseReplay(7), eps=0.03, corrector="none",
                            probability_flow=True, denoise=False)
    assert nfe == 6 and rel_l2(out, z["out"]) < 1e-4


def test_fir_matches_reference():
    z = load("fir")
    x = torch.from_numpy(z["x"])
    assert rel_l2(NO.fir_down2(x), z["down"]) < 1e-6
    assert rel_l2(NO.fir_up2(x), z["up"]) < 1e-6
    assert rel_l2(NO.upfirdn2d_ref(x, torch.from_numpy(z["kern"]), up=3, down=2, pad=(2, 1)), z["generic"]) < 1e-6


@pytest.mark.parametrize("fc,L", [(FO.FrontCfg(), 4000), (FO.FrontCfg.ears_48k(), 9000)])
def test_front_end_self_consistency(fc, L):
    sig = synth.synth_waveform(L, seed=2, batch=2)
    S = FO.stft(sig, fc)
    assert S.shape == (2, fc.n_fft // 2 + 1, L // fc.hop_length + 1)
    assert rel_l2(FO.stft_manual(sig, fc), S) < 2e-6
    assert rel_l2(FO.istft_manual(S, fc, L), FO.istft(S, fc, L)) < 2e-6
    assert rel_l2(FO.spec_back(FO.spec_fwd(S, fc), fc), S) < 1e-5
    Y = FO.pad_spec(FO.spec_fwd(S, fc)[None, :1], "zero_pad")
    assert Y.shape[-1] % 64 == 0


FRONT_CASES = {   # oracle/make_golden_front.py
    "hann_exponent": (FO.FrontCfg(), 6000),
    "sqrthann_log": (FO.FrontCfg(window="sqrthann", transform_type="log"), 6000),
    "hann_none": (FO.FrontCfg(transform_type="none"), 6000),
    "sqrthann_exponent_48k": (FO.FrontCfg(n_fft=1534, hop_length=384, window="sqrthann", spec_factor=0.065, spec_abs_exponent=0.667, sr=48000), 9000),
}


@pytest.mark.parametrize("name", list(FRONT_CASES))
def test_front_end_matches_the_reference_data_module(name):
    """stft / spec_fwd / spec_back / istft of the reference's own SpecsDataModule (data_module.py:13-19,162-218) for the `log` and `none`
    transforms and the `sqrthann` window as well as the defaults: the oracle must reproduce the stored outputs."""
    fc, L = FRONT_CASES[name]
    z = load("front")
    sig = synth.synth_waveform(L, seed=5, batch=2)
    S = torch.from_numpy(z[name + "/stft"])
    assert rel_l2(FO.stft(sig, fc), S) < 1e-6
    assert rel_l2(FO.spec_fwd(S, fc), z[name + "/fwd"]) < 1e-6
    assert rel_l2(FO.spec_back(torch.from_numpy(z[name + "/fwd"]), fc), z[name + "/back"]) < 1e-6
    assert rel_l2(FO.istft(S, fc, L), z[name + "/istft"]) < 1e-6


@pytest.mark.parametrize("stype", ["ode", "sde"])
def test_sb_sampler_matches_reference(stype):
    z = load(f"sb_{stype}_N4")
    cfg = NO.NetCfg.for_variant("ncsnpp_v2", nf=32)
    P = synth.synth_params(cfg, seed=0)
    sv = SO.SBVE(2.6, 0.4, 4)
    model = lambda a, b, c: NO.score_fn_v2(P, cfg, sv, a, b, c, loss_type="data_prediction")
    out, n = SO.sb_sample(SO.SBVE(2.6, 0.4, 4), model, torch.from_numpy(z["y"]), SO.NoiseReplay(7), eps=1e-4, sampler_type=stype)
    assert n == 4 and rel_l2(out, z["out"]) < 1e-4


@pytest.mark.parametrize("name", ["ode_rk45", "ode_rk45_default"])
def test_adaptive_ode_oracle_reproduces_the_reference_drift_and_run(name):
    """oracle/sde_oracle.py::ode_sample_adaptive against the reference's get_ode_sampler(denoise=False) runs (tests/golden/ode_rk45.npz at
    rtol = atol = 1e-3, ode_rk45_default.npz at the reference's default 1e-5; oracle/make_golden_ode.py): the drift at the reference's
    own evaluation points always; the whole trajectory under SGMSE_SLOW=1."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    P = synth.synth_params(cfg, seed=0)
    y = torch.from_numpy(z["y"])
    so = SO.OUVE(1.5, 0.05, 0.5, 30)
    assert bool(z["default_call_raises_typeerror"])          # what the fixed-step sampler stands in for
    for t, xk, fk in zip(z["probe_t"][:3], z["probe_x"][:3], z["probe_f"][:3]):
        xt = torch.from_numpy(xk.reshape(tuple(y.shape)))
        vt = torch.ones(y.shape[0]) * float(t)
        f = so.drift(xt, y) - so.diffusion(vt)[:, None, None, None] ** 2 * NO.score_fn(P, cfg, xt, y, vt) * 0.5
        err = float((f.reshape(-1) - torch.from_numpy(fk)).norm() / torch.from_numpy(fk).norm())
        assert err < 1e-5, (float(t), err)
    if os.environ.get("SGMSE_SLOW"):
        out, nfe = SO.ode_sample_adaptive(so, lambda a, b, c: NO.score_fn(P, cfg, a, b, c), y, SO.NoiseReplay(7), eps=float(z["eps"]),
                                          rtol=float(z["rtol"]), atol=float(z["atol"]))
        err = float((out - torch.from_numpy(z["out"])).norm() / torch.from_numpy(z["out"]).norm())
        assert nfe == int(z["nfe"]) and err < 3.0 * float(z["oracle_vs_reference"]) + 1e-7, (nfe, err)     # (a few times what the fixture script measured: BLAS thread count / scipy version of the box move it)
