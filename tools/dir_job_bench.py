#!/usr/bin/env python3
"""Directory job against the bench rate: a synthetic directory of N 4-second 16 kHz wav files through
`sgmse_amd.enhancement` (random-init full-width ncsnpp checkpoint, PC N=30, batch 32) on one GPU, timed around enhance_files()
-- file reads, resampling-free load, STFT, sampler, iSTFT, file writes all inside -- next to the rate of bench.py's step
(waveforms resident in HBM).  Prints one JSON object.

    python tools/dir_job_bench.py [--files 256] [--seconds 4] [--batch 32] [--jitter 0]   (--jitter s: lengths uniform in [sec - s, sec])
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--jitter", type=float, default=0.0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--N", type=int, default=30)
    ap.add_argument("--ragged", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    from scipy.io import wavfile
    from sgmse_amd import enhancement as E
    from sgmse_amd.model import ScoreModel
    from sgmse_amd.data_module import SpecsDataModule

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    work = tempfile.mkdtemp(prefix="sgmse_dirjob_")
    try:
        hp = dict(backbone="ncsnpp", sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30, t_eps=0.03,
                  data_module_cls=SpecsDataModule, n_fft=510, hop_length=128, spec_factor=0.15, spec_abs_exponent=0.5)
        torch.manual_seed(0)
        model = ScoreModel(**hp)
        model.to(dev).eval()
        model._error_loading_ema = True
        rng = np.random.default_rng(0)
        noisy = os.path.join(work, "noisy")
        os.makedirs(noisy)
        audio_s = 0.0
        for i in range(a.files):
            L = int(16000 * (a.seconds - rng.uniform(0.0, a.jitter)))
            audio_s += L / 16000
            wavfile.write(os.path.join(noisy, f"u{i:04d}.wav"), 16000, (0.1 * rng.standard_normal(L)).astype(np.float32))
        args = argparse.Namespace(sampler_type="pc", corrector="ald", corrector_steps=1, snr=0.5, N=a.N, batch_size=a.batch, seed=3,
                                  ragged=a.ragged, io_threads=4)
        files = E.list_audio(noisy)
        # warm-up: one batch (arena plan + graph capture), not timed
        E.enhance_files(model, files[:a.batch], noisy, os.path.join(work, "warm"), args, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = E.enhance_files(model, files, noisy, os.path.join(work, "out"), args, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the bench step on the same model: waveforms resident in HBM
        y = torch.randn(a.batch, int(16000 * a.seconds), generator=torch.Generator().manual_seed(1)).to(dev)
        model.enhance_batch(y, N=a.N, snr=0.5, seed=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(2):
            model.enhance_batch(y, N=a.N, snr=0.5, seed=2 + i)
        torch.cuda.synchronize()
        bench_rate = 2 * a.batch / (time.perf_counter() - t1)
        ok = all(os.path.getsize(os.path.join(work, "out", os.path.basename(f))) > 1000 for f in files)
        print(json.dumps({"files": n, "all_written": ok, "audio_seconds": audio_s, "wall_s": dt, "utt_per_s_directory_job": n / dt,
                          "rtf_directory_job": dt / audio_s, "utt_per_s_bench_step_same_process": bench_rate,
                          "directory_over_bench": (n / dt) / bench_rate, "batch": a.batch, "N": a.N, "ragged": a.ragged,
                          "graph_captures": model.dnn.engine(dev).graph_captures()}))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
