"""Sampler factories of the enhancement path (reference sgmse/sampling/__init__.py:26-143).

``get_pc_sampler`` returns a zero-argument callable producing ``(sample, nfe)`` like the reference.  When the score
function is a ``ScoreModel`` with a HIP backbone, the SDE is OUVE and the predictor/corrector pair has fused kernels
('reverse_diffusion' | 'none'  x  'ald' | 'langevin' | 'none'), the whole N-step loop runs inside the HIP library
(sgmse_pc_sample: one hipGraph-captured predictor-corrector step replayed N times, per-step constants from a device
table, Philox or replayed noise).  Any other combination falls back to the reference's Python loop over the registry
classes, with every score evaluation still running on the HIP network.
"""
from __future__ import annotations

import warnings
from typing import Optional

import torch

from ..sdes import OUVESDE, SBVESDE
from .correctors import Corrector, CorrectorRegistry
from .predictors import Predictor, PredictorRegistry, ReverseDiffusionPredictor

__all__ = ["PredictorRegistry", "CorrectorRegistry", "Predictor", "Corrector", "get_sampler", "get_pc_sampler",
           "get_ode_sampler", "get_sb_sampler"]

_NATIVE_PRED = ("reverse_diffusion", "none")
_NATIVE_CORR = ("ald", "langevin", "none")


def _native_engine(score_fn, y):
    """The HIP context behind ``score_fn`` if it is a ScoreModel on a HIP backbone using the plain score wrapper."""
    dnn = getattr(score_fn, "dnn", None)
    if dnn is None or not hasattr(dnn, "engine"):
        return None
    if getattr(score_fn, "_native_score_wrapper", False) is not True:
        return None
    return dnn.engine(y[0].device if isinstance(y, (list, tuple)) else y.device)


def _require_native_for_ragged(ctx, y):
    """A list of spectrograms of different lengths (ragged batch, Context.set_frames) only runs on the native samplers."""
    if ctx is None and isinstance(y, (list, tuple)):
        raise TypeError("a list of spectrograms (ragged batch) needs the native HIP sampler: OUVESDE, 'reverse_diffusion'/'none' "
                        "predictor, 'ald'/'none' corrector, no force_python_loop")


def get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, denoise=True, eps=3e-2, snr=0.1, corrector_steps=1,
                   probability_flow: bool = False, intermediate=False, noise: Optional[torch.Tensor] = None,
                   seed: Optional[int] = None, use_graph: bool = True, force_python_loop: bool = False, streams=None, **kwargs):
    """Predictor-corrector sampler (reference sampling/__init__.py:26-70).

    Extra keyword arguments of this implementation: ``noise`` (complex64 [ndraws,B,1,F,T] replayed standard-normal
    draws in the reference's call order, for bit-comparable runs), ``seed`` (Philox seed when ``noise`` is None;
    default: drawn from torch's global RNG), ``streams`` (one integer per utterance naming its noise stream: the draws of an
    utterance then depend on (seed, stream) only, not on its batch slot; default: the slot), ``use_graph``,
    ``force_python_loop``."""
    predictor_cls = PredictorRegistry.get_by_name(predictor_name)   # ValueError for unknown names, like the reference
    corrector_cls = CorrectorRegistry.get_by_name(corrector_name)

    native = (not force_python_loop and predictor_name in _NATIVE_PRED and corrector_name in _NATIVE_CORR
              and isinstance(sde, OUVESDE) and not intermediate)
    ctx = _native_engine(score_fn, y) if native else None
    _require_native_for_ragged(ctx, y)

    if ctx is not None:
        table = sde.step_table(eps, snr, sde.N)
        std1 = float(sde._std(torch.ones(1))[0])
        affine = score_fn.score_affine(table["t"])
        # like the reference's Predictor (predictors.py:18) the PC sampler ignores probability_flow
        def native_pc_sampler():
            s = seed if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            with torch.no_grad():
                out, nfe = ctx.pc_sample(y, table, theta=float(sde.theta), std1=std1, corrector=corrector_name,
                                         corrector_steps=corrector_steps, predictor=predictor_name,
                                         probability_flow=False, denoise=denoise, noise=noise, seed=s, use_graph=use_graph,
                                         affine=affine, snr=snr, streams=streams)
            return out, nfe
        return native_pc_sampler

    predictor = predictor_cls(sde, score_fn, probability_flow=probability_flow)
    corrector = corrector_cls(sde, score_fn, snr=snr, n_steps=corrector_steps)

    def pc_sampler():
        with torch.no_grad():
            xt = sde.prior_sampling(y.shape, y).to(y.device)
            timesteps = torch.linspace(sde.T, eps, sde.N, device=y.device)
            xt_mean = xt
            for i in range(sde.N):
                t = timesteps[i]
                stepsize = t - timesteps[i + 1] if i != len(timesteps) - 1 else timesteps[-1]
                vec_t = torch.ones(y.shape[0], device=y.device) * t
                xt, xt_mean = corrector.update_fn(xt, y, vec_t)
                xt, xt_mean = predictor.update_fn(xt, y, vec_t, stepsize)
            x_result = xt_mean if denoise else xt
            ns = sde.N * (corrector.n_steps + 1)
            return x_result, ns

    return pc_sampler


def _adaptive_ode_sampler(sde, score_fn, y, inverse_scaler, denoise, rtol, atol, method, eps, device, noise, solver_kwargs):
    """The reference's black-box probability-flow solver (sampling/__init__.py:96-143): scipy.integrate.solve_ivp over the flattened
    complex state from t = T down to eps, drift = rsde.sde(x, y, t)[0] with ``probability_flow=True``; every function evaluation is one
    network evaluation on the device with the state round-tripping through host NumPy, exactly as the reference drives it (the solver,
    its step control and its NFE are scipy's: reference pin scipy==1.10.1, requirements_version.txt:14)."""
    if isinstance(y, (list, tuple)):
        raise TypeError("the adaptive ODE sampler integrates one rectangular batch: pass a tensor, not a ragged list")
    from scipy import integrate
    rsde = sde.reverse(score_fn, probability_flow=True)
    dev = y.device if device is None else torch.device(device)

    def ode_sampler(z=None, **kw):
        with torch.no_grad():
            if z is not None:
                x = z
            elif noise is not None:       # replayed prior draw (first entry of the PC samplers' noise tensor layout)
                n0 = noise[0] if noise.dim() == y.dim() + 1 else noise
                x = y + n0.to(y.device) * sde._std(torch.ones(y.shape[0], device=y.device))[:, None, None, None]
            else:
                x = sde.prior_sampling(y.shape, y)
            x = x.to(dev)

            def ode_func(t, xflat):
                xt = torch.from_numpy(xflat.reshape(tuple(y.shape))).to(dev).type(torch.complex64)
                vec_t = torch.ones(y.shape[0], device=dev) * t
                drift = rsde.sde(xt, y.to(dev), vec_t)[0]
                return drift.detach().cpu().numpy().reshape((-1,))

            solution = integrate.solve_ivp(ode_func, (sde.T, eps), x.detach().cpu().numpy().reshape((-1,)), rtol=rtol, atol=atol,
                                           method=method, **solver_kwargs)
            nfe = solution.nfev
            x = torch.tensor(solution.y[:, -1]).reshape(y.shape).to(dev).type(torch.complex64)
            if denoise:
                # (the reference calls predictor.update_fn without its stepsize here and raises TypeError; this is the
                # step that call evidently intends: one reverse-diffusion predictor step of size eps at t = eps, mean returned)
                predictor = ReverseDiffusionPredictor(sde, score_fn, probability_flow=False)
                vec_eps = torch.ones(x.shape[0], device=dev) * eps
                _, x = predictor.update_fn(x, y.to(dev), vec_eps, torch.tensor(eps, device=dev))
            if inverse_scaler is not None:
                x = inverse_scaler(x)
            return x, nfe

    return ode_sampler


def get_ode_sampler(sde, score_fn, y, inverse_scaler=None, denoise=True, rtol=1e-5, atol=1e-5, method="RK45", eps=3e-2,
                    device=None, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, use_graph: bool = True,
                    streams=None, adaptive: Optional[bool] = None, **kwargs):
    """Probability-flow sampler (reference sampling/__init__.py:73-143), in two forms.

    ``adaptive=True`` -- the reference's own behaviour: scipy's black-box solver (``method``, ``rtol``, ``atol``, extra keyword
    arguments passed on to ``solve_ivp``) integrates the flattened state from T to ``eps``, one network evaluation per function
    call; returns ``(sample, solver NFE)``.  It ignores ``sde.N`` like the reference.

    ``adaptive=False`` -- the fixed-step counterpart on the sampler's own time grid (BASELINE configs[2]): N Euler
    steps of ``sde.reverse(score_fn, probability_flow=True).discretize`` (sdes.py:130-135), x <- x - rev_f, no noise, N NFE, as one
    HIP loop; ``rtol/atol/method/inverse_scaler/device`` do not apply to it.

    ``adaptive=None`` (default) follows the reference where the reference runs: its function only completes with
    ``denoise=False`` (with the default ``denoise=True`` it raises TypeError), so ``denoise=False`` selects
    the adaptive solver and the default selects the fixed-step sampler."""
    if adaptive is None:
        adaptive = denoise is False
    if adaptive:
        dropped = [k for k, v in (("seed", seed), ("streams", streams)) if v is not None] + ([] if use_graph else ["use_graph"])
        if dropped:
            warnings.warn(f"get_ode_sampler(adaptive solver): {', '.join(dropped)} do not apply -- scipy's solve_ivp drives one network "
                          f"evaluation per call through host memory (no captured loop); the prior draw comes from `noise` or torch's global "
                          f"generator, and sde.N is ignored as in the reference.  Pass adaptive=False for the fused fixed-step loop")
        return _adaptive_ode_sampler(sde, score_fn, y, inverse_scaler, denoise, rtol, atol, method, eps, device, noise, kwargs)
    ignored = [k for k, v, d in (("rtol", rtol, 1e-5), ("atol", atol, 1e-5), ("method", method, "RK45"), ("denoise", denoise, True),
                                 ("inverse_scaler", inverse_scaler, None)) if v != d]
    if ignored:
        warnings.warn(f"get_ode_sampler(adaptive=False): {', '.join(ignored)} ignored -- this is the fixed-step probability-flow Euler "
                      f"sampler (N steps, N NFE); pass adaptive=True for the reference's scipy solver")
    ctx = _native_engine(score_fn, y) if isinstance(sde, OUVESDE) else None
    _require_native_for_ragged(ctx, y)
    if ctx is None:
        rsde = sde.reverse(score_fn, probability_flow=True)

        def ode_sampler_py(z=None, **kw):
            with torch.no_grad():
                x = sde.prior_sampling(y.shape, y).to(y.device) if z is None else z
                ts = torch.linspace(sde.T, eps, sde.N, device=y.device)
                for i in range(sde.N):
                    dt = ts[i] - ts[i + 1] if i != sde.N - 1 else ts[-1]
                    f, _ = rsde.discretize(x, y, torch.ones(y.shape[0], device=y.device) * ts[i], dt)
                    x = x - f
                return x, sde.N
        return ode_sampler_py

    table = sde.step_table(eps, 0.0, sde.N)
    std1 = float(sde._std(torch.ones(1))[0])
    affine = score_fn.score_affine(table["t"])

    def ode_sampler(z=None, **kw):
        s = seed if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        with torch.no_grad():
            return ctx.pc_sample(y, table, theta=float(sde.theta), std1=std1, corrector="none", corrector_steps=1,
                                 predictor="reverse_diffusion", probability_flow=True, denoise=False, noise=noise, seed=s,
                                 use_graph=use_graph, affine=affine, streams=streams)
    return ode_sampler


def get_sb_sampler(sde, model, y, eps=1e-4, n_steps=50, sampler_type="ode", noise: Optional[torch.Tensor] = None,
                   seed: Optional[int] = None, use_graph: bool = True, force_python_loop: bool = False, streams=None, **kwargs):
    """Schroedinger-bridge samplers (reference sampling/__init__.py:145-249): 'ode' (deterministic) and 'sde'.  With a
    ScoreModel on a HIP backbone the N-step loop runs in the library (sgmse_sb_sample); otherwise the reference-style
    Python loop below.  Returns a zero-argument callable -> (sample, n_steps) like the reference."""
    if sampler_type not in ("ode", "sde"):
        raise ValueError("Invalid type. Choose 'ode' or 'sde'.")
    ctx = None if force_python_loop else (_native_engine(model, y) if isinstance(sde, SBVESDE) else None)
    if ctx is not None:
        table = sde.sb_step_table(eps, sampler_type, sde.N)
        affine = model.score_affine(table["t"])

        def native_sb_sampler():
            s = seed if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            with torch.no_grad():
                out, _ = ctx.sb_sample(y[:, [0], :, :].contiguous(), table, stochastic=(sampler_type == "sde"), noise=noise, seed=s,
                                       affine=affine, use_graph=use_graph, streams=streams)
            return out, n_steps
        return native_sb_sampler

    def python_sb_sampler():
        with torch.no_grad():
            xt = y[:, [0], :, :]
            ts = torch.linspace(sde.T, eps, sde.N + 1, device=y.device)
            prev = sde._sigmas_alphas(ts[0] * torch.ones(xt.shape[0], device=xt.device))
            for t in ts[1:]:
                time = t * torch.ones(xt.shape[0], device=xt.device)
                cur = sde._sigmas_alphas(time)
                (sp, _, sbp, ap, _, _), (st, sT, sbt, at, aT, _) = prev, cur
                est = model(xt, y, time)
                b4 = lambda v: v[:, None, None, None]
                if sampler_type == "sde":
                    tmp = 1 - st ** 2 / (sp ** 2 + sde.eps)
                    wz = 0.0 if t == ts[-1] else b4(at * st * torch.sqrt(tmp))
                    xt = b4(at * st ** 2 / (ap * sp ** 2 + sde.eps)) * xt + b4(at * tmp) * est + wz * torch.randn_like(xt)
                else:
                    xt = (b4(at * st * sbt / (ap * sp * sbp + sde.eps)) * xt
                          + b4(at / (sT ** 2 + sde.eps) * (sbt ** 2 - sbp * st * sbt / (sp + sde.eps))) * est
                          + b4(at / (aT * sT ** 2 + sde.eps) * (st ** 2 - sp * st * sbt / (sbp + sde.eps))) * y)
                prev = cur
            return xt, n_steps
    return python_sb_sampler


def get_sampler(sampler_type, *args, **kwargs):
    if sampler_type == "pc":
        return get_pc_sampler(*args, **kwargs)
    if sampler_type == "ode":
        return get_ode_sampler(*args, **kwargs)
    raise ValueError(f"Given sampler type {sampler_type} not supported")
