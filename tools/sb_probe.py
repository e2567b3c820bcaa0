#!/usr/bin/env python3
"""Where the Schroedinger-bridge ODE sampler leaves the reference's output: step by step on the GPU.

Runs the sb_ode_N4 case (k = 2.6, c = 0.4, N = 4, ncsnpp_v2 data prediction, nf = 32) as the reference-style Python loop over the
HIP network and, at every step, evaluates the CPU oracle's network ON THE SAME INPUT: `est` = relative deviation of the HIP
network's estimate from the oracle's at that input (the per-evaluation error), `x` = deviation of the state from the oracle's own
trajectory (what the sampler makes of it).  tools/sb_conditioning.py shows the law that links the two: the first step adds the
estimate to 5457 |y| and cancels, so a relative difference delta of the estimates becomes sqrt(delta * 0.446 * 4.5e-4) in the state.
Usage: python tools/sb_probe.py [--test-emulator LIB]    (test infrastructure: imports oracle/)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--test-emulator", default=None)
    a = ap.parse_args()
    from sgmse_amd import _lib
    if a.test_emulator:
        _lib.load_library(a.test_emulator)
        dev = "cpu"
    else:
        _lib.load_library()
        dev = "cuda"
    import parity as P
    from oracle import ncsnpp_oracle as NO, sde_oracle as SO
    z = P.load("sb_ode_N4")
    cfg = NO.NetCfg.for_variant("ncsnpp_v2", nf=32)
    m, Pm = P.make_model(cfg, dev, sde="sbve", k=2.6, c=0.4, N=4, loss_type="data_prediction")
    y, ref = torch.from_numpy(z["y"]), torch.from_numpy(z["out"])
    sv = SO.SBVE(2.6, 0.4, 4)
    orc = lambda x, yy, t: NO.score_fn_v2(Pm, cfg, sv, x, yy, t, loss_type="data_prediction")

    # the oracle's own trajectory
    traj = []
    def orc_rec(x, yy, t):
        traj.append(x.clone())
        return orc(x, yy, t)
    x_orc, _ = SO.sb_sample(SO.SBVE(2.6, 0.4, 4), orc_rec, y, SO.NoiseReplay(7), eps=1e-4, sampler_type="ode")
    print(f"oracle vs the reference's output: {rel(x_orc, ref):.3e}")

    fused, _ = m.get_sb_sampler(m.sde, y.to(dev), sampler_type="ode", n_steps=4)()
    print(f"HIP fused loop vs the reference's output: {rel(fused.cpu(), ref):.3e}   vs the oracle's: {rel(fused.cpu(), x_orc):.3e}")

    # reference-style loop over the HIP network, with the oracle evaluated on the same inputs
    sde = m.sde
    b4 = lambda v: v[:, None, None, None]
    with torch.no_grad():
        xt = y.clone()
        ts = torch.linspace(sde.T, 1e-4, sde.N + 1)
        sp, _, sbp, ap, _, _ = sde._sigmas_alphas(ts[0] * torch.ones(xt.shape[0]))
        for i, t in enumerate(ts[1:]):
            time = t * torch.ones(xt.shape[0])
            st, sT, sbt, at, aT, _ = sde._sigmas_alphas(time)
            est_hip = m(xt.to(dev), y.to(dev), time.to(dev)).cpu()
            est_orc = orc(xt, y, time)
            w_prev = at * st * sbt / (ap * sp * sbp + sde.eps)
            w_est = at / (sT ** 2 + sde.eps) * (sbt ** 2 - sbp * st * sbt / (sp + sde.eps))
            w_y = at / (aT * sT ** 2 + sde.eps) * (st ** 2 - sp * st * sbt / (sbp + sde.eps))
            x_in_dev = rel(xt, traj[i])
            xt = b4(w_prev) * xt + b4(w_est) * est_hip + b4(w_y) * y
            print(f"step {i}: t = {float(t):.4f}  w_prev {float(w_prev[0]):10.4f} w_est {float(w_est[0]):.4f} w_y {float(w_y[0]):11.4f}   "
                  f"state in: {x_in_dev:.3e} off the oracle's trajectory   est(HIP) vs est(oracle) at the same input: {rel(est_hip, est_orc):.3e}   "
                  f"|est|/|y| = {float(est_orc.norm() / y.norm()):.3f}")
            sp, sbp, ap = st, sbt, at
        print(f"Python loop over the HIP network vs the reference's output: {rel(xt, ref):.3e}   vs the fused loop: {rel(xt, fused.cpu()):.3e}")


if __name__ == "__main__":
    main()
