"""Full-configuration golden vectors: the BASELINE.json configurations themselves, run through the REFERENCE.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (imports /root/reference); the GPU box consumes the
fixtures it writes.  For each BASELINE configuration the real reference modules (``sgmse.backbones`` NCSNpp /
NCSNpp_48k at FULL width, ``sgmse.sdes.OUVESDE``, ``sgmse.sampling.get_pc_sampler``) are run on ONE utterance with
synthetic weights (oracle/synth.py, seed 0) and replayed noise (seed 7), wrapped in the per-file pipeline of
enhancement.py:62-99 (normalise -> STFT -> spec_fwd -> pad_spec -> sampler -> spec_back -> iSTFT -> renormalise; the
front/back end are torch.stft / torch.istft with the reference's kwargs, data_module.py:212-218 -- sgmse.data_module
itself does not import here).  Everything that is needed to rebuild the inputs is a seed, so a fixture
only stores the outputs: the sampled spectrogram [1,1,F,T] complex64 and the enhanced waveform [L] float32.

  pc16k_full   configs[0]/[1]: ncsnpp 65.6 M, 4 s @16 kHz (F=256, T=512), PC reverse_diffusion+ALD N=30 snr=0.5  (60 NFE)
  ode16k_full  configs[2]:     same network and utterance, fixed-step probability-flow Euler N=30              (30 NFE)
  pc48k_full   configs[3]:     ncsnpp_48k, F=768, T=128 (0.96 s @48 kHz, reflection pad), PC N=50 snr=0.33     (100 NFE)
  pc48k_T512   configs[3] at the benched shape: F=768, T=512 (4 s @48 kHz), PC N=5 snr=0.33                   (10 NFE)
  pc48k_T512_N50  configs[3] as benched: the same utterance with N=50                                          (100 NFE)

The oracle restatement runs beside the reference on the same inputs and its deviation goes into
tests/golden/REPORT_full.txt (this is the oracle's pin at full configuration).

    HIP_VISIBLE_DEVICES="" python -m oracle.make_golden_full [pc16k_full ode16k_full pc48k_full]
"""
from __future__ import annotations

import os
import sys
import time

os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
sys.dont_write_bytecode = True

import numpy as np
import torch

from oracle import ncsnpp_oracle as NO
from oracle import sde_oracle as SO
from oracle import stft_oracle as FO
from oracle import synth
from oracle.make_golden import REF, OUT, RefNoise, ref_model, rel
from oracle.full_cases import FULL_CASES, front_cfg

def run_case(name, with_oracle=True):
    from sgmse import sampling
    from sgmse.sdes import OUVESDE
    c = FULL_CASES[name]
    cfg = NO.NetCfg.for_variant(c["variant"])
    P = synth.synth_params(cfg, seed=c["param_seed"])
    m = ref_model(cfg, P)
    fc = front_cfg(c["front"])
    y = synth.synth_waveform(c["L"], seed=c["wave_seed"], batch=1)

    def ref_score(x, yy, t):                   # model.py:307-310 (sgmse.model is not importable here)
        return -m(torch.cat([x, yy], dim=1), t)

    rs = OUVESDE(N=c["N"], **c["sde"])
    spec = {}

    def ref_sampler(Y):
        noise = RefNoise(c["noise_seed"])
        orig = torch.randn_like
        torch.randn_like = noise
        try:
            if c["sampler"] == "pc":
                x, nfe = sampling.get_pc_sampler("reverse_diffusion", "ald", sde=rs, score_fn=ref_score, y=Y, eps=0.03,
                                                 snr=c["snr"], corrector_steps=1)()
            else:                              # fixed-step PF-ODE assembled from reference pieces
                rsde = rs.reverse(ref_score, probability_flow=True)
                with torch.no_grad():
                    x = rs.prior_sampling(Y.shape, Y)
                    ts = torch.linspace(rs.T, 0.03, rs.N)
                    for i in range(rs.N):
                        dt = ts[i] - ts[i + 1] if i != rs.N - 1 else ts[-1]
                        f, _ = rsde.discretize(x, Y, torch.ones(Y.shape[0]) * ts[i], dt)
                        x = x - f
                nfe = rs.N
        finally:
            torch.randn_like = orig
        spec["x"], spec["nfe"], spec["Y"] = x, nfe, Y
        return x

    t0 = time.time()
    wav_ref = FO.enhance(y, fc, ref_sampler, pad_mode=c["pad"])
    t_ref = time.time() - t0
    line = f"{name:12s} reference: {spec['nfe']} NFE at {tuple(spec['Y'].shape)} in {t_ref:.0f} s"
    if with_oracle:
        so = SO.OUVE(c["sde"]["theta"], c["sde"]["sigma_min"], c["sde"]["sigma_max"], c["N"])
        ospec = {}

        def orc_sampler(Y):
            x, _ = SO.pc_sample(so, lambda a, b, cc: NO.score_fn(P, cfg, a, b, cc), Y, SO.NoiseReplay(c["noise_seed"]), eps=0.03,
                                snr=c["snr"], corrector="ald" if c["sampler"] == "pc" else "none",
                                probability_flow=c["sampler"] == "ode", denoise=c["sampler"] == "pc")
            ospec["x"] = x
            return x
        wav_orc = FO.enhance(y, fc, orc_sampler, pad_mode=c["pad"])
        line += f"; oracle vs reference: spectrogram {rel(ospec['x'], spec['x']):.3e}, waveform {rel(wav_orc, wav_ref):.3e}"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), spec=spec["x"].numpy(), wave=wav_ref.numpy(), nfe=np.int64(spec["nfe"]))
    print(line, flush=True)
    return line


def main(names):
    sys.path.insert(0, REF)
    torch.set_num_threads(os.cpu_count())
    lines = [run_case(n) for n in names]
    path = os.path.join(OUT, "REPORT_full.txt")
    old = [l.rstrip("\n") for l in open(path)] if os.path.exists(path) else []
    keep = [l for l in old[1:] if l.split(" ")[0] not in names]
    with open(path, "w") as fh:
        fh.write("full-configuration fixtures: reference run time and oracle-vs-reference relative L2 (oracle/make_golden_full.py)\n")
        for l in keep + lines:
            fh.write(l + "\n")


if __name__ == "__main__":
    main(sys.argv[1:] or list(FULL_CASES))
