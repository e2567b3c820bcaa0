"""The BASELINE.json configurations as single-utterance parity cases (TEST INFRASTRUCTURE ONLY; no side effects on import).

Everything needed to rebuild a case's inputs is a seed: synthetic weights (oracle/synth.py), a N(0,1) waveform and the
replayed noise stream (oracle/sde_oracle.NoiseReplay).  oracle/make_golden_full.py runs the REFERENCE on these cases in
the build container and stores only the outputs (tests/golden/<name>.npz: sampled spectrogram + enhanced waveform)."""
from . import stft_oracle as FO

FULL_CASES = {
    # configs[0]/[1]: ncsnpp 65.6 M, 4 s @16 kHz (F=256, T=512), PC reverse_diffusion + ALD, N=30, snr=0.5 (60 NFE)
    "pc16k_full": dict(variant="ncsnpp", L=64000, front="vb", pad="zero_pad", sampler="pc", N=30, snr=0.5,
                       sde=dict(theta=1.5, sigma_min=0.05, sigma_max=0.5), wave_seed=0, noise_seed=7, param_seed=0),
    # configs[2]: same network and utterance, fixed-step probability-flow Euler, N=30 (30 NFE)
    "ode16k_full": dict(variant="ncsnpp", L=64000, front="vb", pad="zero_pad", sampler="ode", N=30, snr=0.5,
                        sde=dict(theta=1.5, sigma_min=0.05, sigma_max=0.5), wave_seed=0, noise_seed=7, param_seed=0),
    # configs[3]: ncsnpp_48k, F=768, T=128 (0.96 s @48 kHz: 121 frames, reflection-padded), PC N=50, snr=0.33 (100 NFE)
    "pc48k_full": dict(variant="ncsnpp_48k", L=46080, front="ears", pad="reflection", sampler="pc", N=50, snr=0.33,
                       sde=dict(theta=2.0, sigma_min=0.1, sigma_max=1.0), wave_seed=0, noise_seed=7, param_seed=0),
    # configs[3] at the BENCHED shape: 4 s @48 kHz (501 frames -> T=512, F=768), PC with N=5 (10 NFE: what fits a CPU run of the
    # reference, ~40 s per evaluation) -- pins the 48 kHz sampler and network at full length
    "pc48k_T512": dict(variant="ncsnpp_48k", L=192000, front="ears", pad="reflection", sampler="pc", N=5, snr=0.33,
                       sde=dict(theta=2.0, sigma_min=0.1, sigma_max=1.0), wave_seed=0, noise_seed=7, param_seed=0),
    # configs[3] exactly as benched: the same 4 s @48 kHz utterance with its own N = 50 (100 NFE; ~25 min of reference CPU time)
    "pc48k_T512_N50": dict(variant="ncsnpp_48k", L=192000, front="ears", pad="reflection", sampler="pc", N=50, snr=0.33,
                           sde=dict(theta=2.0, sigma_min=0.1, sigma_max=1.0), wave_seed=0, noise_seed=7, param_seed=0),
}


def front_cfg(tag):
    return FO.FrontCfg() if tag == "vb" else FO.FrontCfg.ears_48k()
