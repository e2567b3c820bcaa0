"""Golden fixture of the reference's ADAPTIVE probability-flow sampler (sampling/__init__.py:73-143, scipy RK45), which rounds 1-3
replaced by its fixed-step counterpart only.  Build container only (imports /root/reference):

    HIP_VISIBLE_DEVICES="" PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo python -m oracle.make_golden_ode [ode_rk45] [ode_rk45_default]

Runs the reference's own ``get_ode_sampler`` (``denoise=False``: the default raises TypeError at :101) on the reference's
NCSNpp with the synthetic parameters of oracle/synth.py and a replayed prior draw, checks oracle/sde_oracle.py::ode_sample_adaptive
against it and writes tests/golden/ode_rk45.npz."""
from __future__ import annotations

import os
import sys
import time

os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = os.environ.get("SGMSE_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import ncsnpp_oracle as NO
from oracle import sde_oracle as SO
from oracle import synth
from oracle.make_golden import RefNoise, ref_model, rel

# fixtures: name -> tolerance.  ode_rk45: a loose tolerance that exercises the step control in ~90 evaluations; ode_rk45_default: the
# reference's own default rtol = atol = 1e-5 (sampling/__init__.py:73-77) on the same short utterance (round 5, VERDICT r4 missing 3)
FIXTURES = {"ode_rk45": 1e-3, "ode_rk45_default": 1e-5}
PROBES = (0, 1, 7, 40, -1)  # function evaluations of the reference run kept in the fixture: (t, state handed to the drift, drift returned)


def main(name="ode_rk45"):
    RTOL = ATOL = FIXTURES[name]
    sys.path.insert(0, REF)
    import scipy
    from sgmse import sampling
    from sgmse.sdes import OUVESDE
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    P = synth.synth_params(cfg, seed=0)
    m = ref_model(cfg, P)

    def ref_score(x, y, t):                    # model.py:307-310
        return -m(torch.cat([x, y], dim=1), t)

    y = synth.synth_spec(1, 256, 64, seed=3)
    rs = OUVESDE(theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30)
    calls = []

    class Recording:                           # sampling.integrate.solve_ivp with every function evaluation recorded
        def __init__(self, real):
            self.real = real

        def solve_ivp(self, fun, *a, **kw):
            def rec(t, x):
                f = fun(t, x)
                calls.append((float(t), np.array(x, copy=True), np.array(f, copy=True)))
                return f
            return self.real.solve_ivp(rec, *a, **kw)

    orig = torch.randn_like
    orig_int = sampling.integrate
    torch.randn_like = RefNoise(7)
    sampling.integrate = Recording(orig_int)
    t0 = time.time()
    try:
        sampler = sampling.get_ode_sampler(rs, ref_score, y, denoise=False, rtol=RTOL, atol=ATOL, method="RK45", device="cpu")
        x_ref, nfe = sampler()
    finally:
        torch.randn_like = orig
        sampling.integrate = orig_int
    t_ref = time.time() - t0
    so = SO.OUVE(1.5, 0.05, 0.5, 30)
    x_orc, nfe2 = SO.ode_sample_adaptive(so, lambda a, b, c: NO.score_fn(P, cfg, a, b, c), y, SO.NoiseReplay(7), eps=0.03,
                                         rtol=RTOL, atol=ATOL, method="RK45")
    r = rel(x_orc, x_ref)
    print(f"{name}: reference nfe {nfe} ({t_ref:.0f} s), oracle nfe {nfe2}, oracle vs reference {r:.3e}, scipy {scipy.__version__}")
    # (the trajectory of this random-weight network is sensitive: last-digit differences of the network grow ~1000x over the
    #  integration, so the END STATE is bounded loosely and the parity gate sits on single function evaluations at the reference's inputs)
    assert abs(nfe2 - nfe) <= 12 and r < 2e-2, (r, nfe, nfe2)
    assert len(calls) == nfe
    probes = [calls[i] for i in PROBES]
    worst = 0.0
    for (t, xk, fk) in probes:
        xt = torch.from_numpy(xk.reshape(tuple(y.shape))).type(torch.complex64)
        vt = torch.ones(y.shape[0]) * t
        g = so.diffusion(vt)
        f_orc = so.drift(xt, y) - g[:, None, None, None] ** 2 * NO.score_fn(P, cfg, xt, y, vt) * 0.5
        worst = max(worst, rel(f_orc.reshape(-1), torch.from_numpy(fk)))
    print(f"oracle drift vs the reference's at {len(probes)} of its evaluation points: worst {worst:.3e}")
    assert worst < 1e-5, worst
    # also the default denoise=True: pin the reference's TypeError (what the fixed-step sampler of SURVEY 8-a9 stands in for)
    raised = False
    try:
        sampling.get_ode_sampler(rs, ref_score, y, rtol=1e-2, atol=1e-2, device="cpu")()
    except TypeError:
        raised = True
    assert raised, "the reference's default get_ode_sampler() no longer raises"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), y=y.numpy(), out=x_ref.numpy(), nfe=np.int64(nfe), nfe_oracle=np.int64(nfe2),
                        noise_seed=np.int64(7), rtol=RTOL, atol=ATOL, eps=0.03,
                        probe_t=np.array([c[0] for c in probes]), probe_x=np.stack([c[1].astype(np.complex64) for c in probes]),
                        probe_f=np.stack([c[2].astype(np.complex64) for c in probes]), probe_oracle_worst=worst, scipy_version=np.array(scipy.__version__),
                        oracle_vs_reference=r, default_call_raises_typeerror=np.bool_(raised))
    with open(os.path.join(OUT, "REPORT.txt"), "a") as fh:
        fh.write(f"{name} (reference get_ode_sampler, denoise=False, rtol=atol={RTOL}): nfe {nfe}, oracle nfe {nfe2}, "
                 f"oracle end state vs reference {r:.3e}, oracle drift at {len(probes)} evaluation points worst {worst:.3e}, scipy {scipy.__version__}; default denoise=True raises TypeError: {raised}\n")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["ode_rk45"]):
        main(n)
