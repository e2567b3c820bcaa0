#!/usr/bin/env python
"""One batch of utterances of MIXED lengths through the PC sampler, three ways on one GPU:

  ragged    all of them in ONE launch sequence (sgmse_set_frames: every tensor packed per utterance, own row strides)
  bucketed  grouped by padded frame count, one uniform batch per group (what the directory script does by default)
  uniform   the same number of utterances, all of the mean length (the reference point: BASELINE configs[1] when the mean is 4 s)

Every way gives an utterance the same bits (tests); this measures what the mixing costs.  Synthetic waveforms, random-init
full-size NCSN++, N = 30, lengths spread evenly over [--lo, --hi] seconds."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lo", type=float, default=2.0)
    ap.add_argument("--hi", type=float, default=6.0)
    ap.add_argument("--N", type=int, default=30)
    ap.add_argument("--modes", default="ragged,bucketed,uniform")
    ap.add_argument("--test-emulator", default=None)
    ap.add_argument("--profile", action="store_true", help="also one instrumented evaluation of the ragged and the uniform batch (per kernel class)")
    a = ap.parse_args()
    from sgmse_amd import _lib
    from sgmse_amd.model import ScoreModel
    from sgmse_amd.util.other import pad_spec
    emu = a.test_emulator is not None
    if emu:
        _lib.load_library(a.test_emulator)
        dev = torch.device("cpu")
    else:
        _lib.load_library()
        dev = torch.device("cuda", 0)
    sync = (lambda: None) if emu else torch.cuda.synchronize
    torch.manual_seed(0)
    model = ScoreModel("ncsnpp", "ouve", N=a.N, theta=1.5, sigma_min=0.05, sigma_max=0.5, **(dict(nf=32) if emu else {}))
    model.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    secs = [a.lo + (a.hi - a.lo) * i / max(a.batch - 1, 1) for i in range(a.batch)]
    waves = [torch.randn(1, int(s * 16000), generator=g).to(dev) for s in secs]

    def spec(w):
        return pad_spec(model._forward_transform(model._stft(w / w.abs().max())).unsqueeze(1), mode="zero_pad")      # [1,1,F,T_pad]

    Ys = [spec(w) for w in waves]
    frames = [int(Y.shape[-1]) for Y in Ys]
    mean_s = sum(secs) / len(secs)
    Yu = torch.cat([spec(torch.randn(1, int(mean_s * 16000), generator=g).to(dev)) for _ in range(a.batch)])

    def run_ragged(seed):
        return model.get_pc_sampler("reverse_diffusion", "ald", [Y[0] for Y in Ys], snr=0.5, seed=seed)()

    def run_bucketed(seed):
        for T in sorted(set(frames)):
            idx = [i for i, f in enumerate(frames) if f == T]
            model.get_pc_sampler("reverse_diffusion", "ald", torch.cat([Ys[i] for i in idx]), snr=0.5, seed=seed, streams=idx)()

    def run_uniform(seed):
        return model.get_pc_sampler("reverse_diffusion", "ald", Yu, snr=0.5, seed=seed)()

    out = {"batch": a.batch, "seconds": [round(s, 2) for s in secs], "frames": frames, "sum_frames": sum(frames),
           "buckets": len(set(frames)), "uniform_frames": int(Yu.shape[-1]), "N": a.N}
    for mode in a.modes.split(","):
        fn = {"ragged": run_ragged, "bucketed": run_bucketed, "uniform": run_uniform}[mode]
        fn(3)                      # warm-up: arena plan, graph capture(s)
        sync()
        t0 = time.perf_counter()
        fn(4)
        sync()
        out[mode + "_s"] = round(time.perf_counter() - t0, 4)
        print(mode, out[mode + "_s"], "s", flush=True)
    if a.profile and not emu:
        # one instrumented evaluation each (eager, HIP events per kernel class): where the mixing costs
        ctx = model.dnn.engine(dev)
        tt = torch.full((a.batch,), 0.5, device=dev)
        xr = [torch.cat([Y[0], Y[0]]) for Y in Ys]                       # [2,F,T_b]
        pr, _ = ctx.profile_forward(xr, tt)
        pu, _ = ctx.profile_forward(torch.cat([Yu, Yu], dim=1), tt)
        out["classes_ms_ragged_vs_uniform"] = {k: [round(pr[k]["ms"], 3), round(pu[k]["ms"], 3)] for k in pr}
    if "uniform_s" in out and "ragged_s" in out:
        out["ragged_vs_uniform_per_frame"] = round((out["ragged_s"] / sum(frames)) / (out["uniform_s"] / (a.batch * int(Yu.shape[-1]))), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
