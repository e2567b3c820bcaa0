"""ScoreModel: the model facade of the enhancement path (reference sgmse/model.py:22-465, inference methods).

A plain ``torch.nn.Module`` (pytorch_lightning / torch_ema are not needed for inference): builds the backbone and SDE
from the registries, loads Lightning checkpoints (``state_dict`` with ``dnn.`` prefix, ``hyper_parameters``, ``ema``),
swaps the EMA weights in on ``eval()`` as the reference does (model.py:111-125), and exposes ``forward``,
``get_pc_sampler``, ``get_ode_sampler``, ``to_audio``, ``_stft/_istft/_forward_transform/_backward_transform`` and
``enhance``.  Training-side methods (losses, steps, optimisers) are out of scope and raise.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import time
import warnings
from math import ceil
from typing import List, Optional

import torch
import torch.nn as nn

from . import sampling
from .backbones import BackboneRegistry
from .data_module import SpecsDataModule
from .sdes import SDERegistry
from .util.other import pad_spec


@contextlib.contextmanager
def _reference_package_alias():
    """Make ``import sgmse[.x.y]`` resolve to sgmse_amd (the alias package sgmse_amd/compat/sgmse) while a checkpoint written by
    the reference is unpickled.  A real ``sgmse`` package that is already imported is left alone."""
    if "sgmse" in sys.modules:
        yield
        return
    compat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")
    sys.path.insert(0, compat)
    try:
        importlib.import_module("sgmse")
        yield
    finally:
        try:
            sys.path.remove(compat)
        except ValueError:
            pass


class ScoreModel(nn.Module):
    _native_score_wrapper = True   # forward is an affine wrapper of the backbone (score_affine): the fused HIP sampler applies
                                   # it inside the network's entry / exit kernels

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--lr", type=float, default=1e-4)
        parser.add_argument("--ema_decay", type=float, default=0.999)
        parser.add_argument("--t_eps", type=float, default=0.03)
        parser.add_argument("--num_eval_files", type=int, default=20)
        parser.add_argument("--loss_type", type=str, default="score_matching")
        parser.add_argument("--sr", type=int, default=16000)
        return parser

    def __init__(self, backbone, sde, lr=1e-4, ema_decay=0.999, t_eps=0.03, num_eval_files=20, loss_type="score_matching",
                 loss_weighting="sigma^2", network_scaling=None, c_in="1", c_out="1", c_skip="0", sigma_data=0.1,
                 l1_weight=0.001, pesq_weight=0.0, sr=16000, data_module_cls=None, **kwargs):
        super().__init__()
        self.backbone = backbone
        dnn_cls = BackboneRegistry.get_by_name(backbone)      # ValueError for unknown names
        self.dnn = dnn_cls(**kwargs)
        sde_cls = SDERegistry.get_by_name(sde)
        self.sde = sde_cls(**kwargs)
        self.lr, self.ema_decay, self.t_eps = lr, ema_decay, t_eps
        self.loss_type, self.loss_weighting = loss_type, loss_weighting
        self.l1_weight, self.pesq_weight = l1_weight, pesq_weight
        self.network_scaling, self.c_in, self.c_out, self.c_skip, self.sigma_data = network_scaling, c_in, c_out, c_skip, sigma_data
        self.num_eval_files, self.sr = num_eval_files, sr
        # every constructor argument, as the reference's save_hyperparameters(ignore=['no_wandb']) keeps them (model.py:87):
        # ScoreModel(**model.hparams) rebuilds the same model, score wrapper included (the torchrun path of the directory script)
        self.hparams = dict(backbone=backbone, sde=sde, lr=lr, ema_decay=ema_decay, t_eps=t_eps, num_eval_files=num_eval_files,
                            loss_type=loss_type, loss_weighting=loss_weighting, network_scaling=network_scaling, c_in=c_in,
                            c_out=c_out, c_skip=c_skip, sigma_data=sigma_data, l1_weight=l1_weight, pesq_weight=pesq_weight,
                            sr=sr, data_module_cls=data_module_cls, **kwargs)
        data_module_cls = SpecsDataModule if data_module_cls is None else data_module_cls
        self.data_module = data_module_cls(**kwargs, gpu=kwargs.get("gpus", 0) > 0)
        # EMA bookkeeping (what torch_ema's store / copy_to / restore do for the reference, model.py:111-122)
        self._ema_shadow: Optional[List[torch.Tensor]] = None
        self._ema_stored: Optional[List[torch.Tensor]] = None
        self._error_loading_ema = False

    # -- checkpoints ---------------------------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
        """Read a Lightning ``.ckpt`` written by the reference's train.py without
        pytorch_lightning: ``hyper_parameters`` -> constructor, ``state_dict`` (keys ``dnn.*``) -> backbone,
        ``ema`` -> shadow parameters used by ``eval()``."""
        # the reference's train.py pickles classes of its own package (sgmse.data_module.SpecsDataModule in hyper_parameters):
        # resolve that package name to this implementation for the duration of the unpickling
        with _reference_package_alias():
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        hp.pop("no_wandb", None)
        model = cls(**hp)
        sd = {k[len("dnn."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("dnn.")}
        model.dnn.load_state_dict(sd, strict=strict)
        model.on_load_checkpoint(ckpt)
        if map_location is not None:
            model.to(map_location)
        return model

    def on_load_checkpoint(self, checkpoint):
        ema = checkpoint.get("ema", None)
        if ema is None:
            self._error_loading_ema = True
            warnings.warn("EMA state_dict not found in checkpoint!")
            return
        shadow = [p.detach().clone().float() for p in ema["shadow_params"]]
        params = list(self.dnn.parameters())
        trainable = [p for p in params if p.requires_grad]
        if len(shadow) == len(params):
            self._ema_targets = params
        elif len(shadow) == len(trainable):      # torch_ema tracks only requires_grad parameters
            self._ema_targets = trainable
        else:
            self._error_loading_ema = True
            warnings.warn(f"EMA has {len(shadow)} tensors but the backbone has {len(params)} parameters; ignoring EMA.")
            return
        self._ema_shadow = shadow

    def train(self, mode=True, no_ema=False):
        res = super().train(mode)
        if not self._error_loading_ema and self._ema_shadow is not None:
            if mode is False and not no_ema:
                if self._ema_stored is None:
                    self._ema_stored = [p.detach().clone() for p in self._ema_targets]
                with torch.no_grad():
                    for p, s in zip(self._ema_targets, self._ema_shadow):
                        p.copy_(s.to(p.device))
                self.dnn.mark_weights_changed()
            elif self._ema_stored is not None:
                with torch.no_grad():
                    for p, s in zip(self._ema_targets, self._ema_stored):
                        p.copy_(s.to(p.device))
                self._ema_stored = None
                self.dnn.mark_weights_changed()
        return res

    def eval(self, no_ema=False):
        return self.train(False, no_ema=no_ema)

    # -- training side: out of scope ---------------------------------------------------------------------------
    def _loss(self, *a, **k):
        raise NotImplementedError("training is outside the MI355X hot path (inference only)")

    _step = training_step = validation_step = configure_optimizers = _loss

    # -- score function (reference model.py:264-341) ------------------------------------------------------------------------
    def _c_in(self, t):
        if self.c_in == "1":
            return 1.0
        if self.c_in == "edm":
            sigma = self.sde._std(t)
            return (1.0 / torch.sqrt(sigma ** 2 + self.sigma_data ** 2))[:, None, None, None]
        raise ValueError("Invalid c_in type: {}".format(self.c_in))

    def _c_out(self, t):
        if self.c_out == "1":
            return 1.0
        if self.c_out == "sigma":
            return self.sde._std(t)[:, None, None, None]
        if self.c_out == "1/sigma":
            return 1.0 / self.sde._std(t)[:, None, None, None]
        if self.c_out == "edm":
            sigma = self.sde._std(t)
            return ((sigma * self.sigma_data) / torch.sqrt(self.sigma_data ** 2 + sigma ** 2))[:, None, None, None]
        raise ValueError("Invalid c_out type: {}".format(self.c_out))

    def _c_skip(self, t):
        if self.c_skip == "0":
            return 0.0
        if self.c_skip == "edm":
            sigma = self.sde._std(t)
            return (self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2))[:, None, None, None]
        raise ValueError("Invalid c_skip type: {}".format(self.c_skip))

    def forward(self, x_t, y, t):
        """Score of the perturbed spectrogram.  Old-code branch (ncsnpp / ncsnpp_48k, model.py:307-310):
        -dnn(cat[x_t, y], t).  New-code branch (ncsnpp_v2, model.py:284-304): scaled inputs, network_scaling, and the
        loss-type dependent output map.  The backbone runs on the HIP engine; the few element-wise operations of this
        method are only used when the model is called directly (the fused sampler applies the same wrapper inside the
        network's entry/exit kernels, see ``score_affine``)."""
        if self.backbone == "ncsnpp_v2":
            F = self.dnn(self._c_in(t) * x_t, self._c_in(t) * y, t)
            if self.network_scaling == "1/sigma":
                F = F / self.sde._std(t)[:, None, None, None]
            elif self.network_scaling == "1/t":
                F = F / t[:, None, None, None]
            if self.loss_type == "score_matching":
                return self._c_skip(t) * x_t + self._c_out(t) * F
            if self.loss_type == "denoiser":
                return (F - x_t) / self.sde._std(t)[:, None, None, None].pow(2)
            if self.loss_type == "data_prediction":
                return self._c_skip(t) * x_t + self._c_out(t) * F
            raise ValueError("Invalid loss type: {}".format(self.loss_type))
        dnn_input = torch.cat([x_t, y], dim=1)
        return -self.dnn(dnn_input, t)

    def score_affine(self, ts: torch.Tensor):
        """The wrapper above as per-time-step scalars: the network sees in_scale*x_t, in_scale*y and
        score = alpha*x_t + beta*F.  Returns None for the old-code branch (in_scale 1, alpha 0, beta -1, built into the
        engine) or three fp32 tensors of len(ts) for ncsnpp_v2 models."""
        if self.backbone != "ncsnpp_v2":
            return None
        ts = ts.to("cpu", torch.float32)
        one = torch.ones_like(ts)
        flat = lambda v: (v.reshape(-1) if torch.is_tensor(v) else one * v).to(torch.float32)
        std = self.sde._std(ts)
        scale = one
        if self.network_scaling == "1/sigma":
            scale = 1.0 / std
        elif self.network_scaling == "1/t":
            scale = 1.0 / ts
        gamma = flat(self._c_in(ts))
        if self.loss_type == "denoiser":
            alpha, beta = -1.0 / std.pow(2), scale / std.pow(2)
        elif self.loss_type in ("score_matching", "data_prediction"):
            alpha, beta = flat(self._c_skip(ts)), flat(self._c_out(ts)) * scale
        else:
            raise ValueError("Invalid loss type: {}".format(self.loss_type))
        return gamma, flat(alpha), flat(beta)

    # -- samplers (reference model.py:348-390) -----------------------------------------------------------------
    @staticmethod
    def _minibatch_kwargs(kwargs, lo, hi):
        """Sampler keyword arguments of the serial chunk [lo, hi) of a batch: replayed noise and noise-stream ids are per
        utterance; without explicit stream ids a chunk's utterances keep the ids they have in the whole batch (their position),
        so that chunked and unchunked runs of one seed draw the same noise."""
        kw = dict(kwargs)
        if kw.get("noise") is not None:
            kw["noise"] = kw["noise"][:, lo:hi].contiguous()
        kw["streams"] = list(range(lo, hi)) if kw.get("streams") is None else list(kw["streams"][lo:hi])
        return kw

    def _chunked(self, make_sampler, y, minibatch, kwargs):
        """``minibatch`` wrapper of reference model.py:356-368 / 378-390: the batch in serial chunks, samples concatenated and
        the chunks' nfe collected in a list."""
        ragged = isinstance(y, (list, tuple))         # a list of spectrograms of different lengths (Context.set_frames)
        total = len(y) if ragged else y.shape[0]

        def batched_sampling_fn():
            samples, ns = [], []
            for lo in range(0, total, minibatch):
                hi = min(lo + minibatch, total)
                sample, n = make_sampler(y[lo:hi], self._minibatch_kwargs(kwargs, lo, hi))()
                samples.append(sample)
                ns.append(n)
            return ([s for chunk in samples for s in chunk] if ragged else torch.cat(samples, dim=0)), ns
        return batched_sampling_fn

    def get_pc_sampler(self, predictor_name, corrector_name, y, N=None, minibatch=None, **kwargs):
        sde = self.sde.copy()
        sde.N = self.sde.N if N is None else N
        kwargs = {"eps": self.t_eps, **kwargs}
        make = lambda yy, kw: sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=yy, **kw)
        return make(y, kwargs) if minibatch is None else self._chunked(make, y, minibatch, kwargs)

    def get_ode_sampler(self, y, N=None, minibatch=None, **kwargs):
        """reference model.py:370-390.  ``denoise=False`` (the only setting under which the reference's function completes) selects the
        reference's adaptive scipy solver -- host round trips, ``seed`` / ``use_graph`` / ``streams`` / ``N`` do not apply (a warning says
        so); the default runs the fused fixed-step probability-flow loop.  ``adaptive=True/False`` chooses explicitly
        (sampling.get_ode_sampler)."""
        sde = self.sde.copy()
        sde.N = self.sde.N if N is None else N
        kwargs = {"eps": self.t_eps, **kwargs}
        make = lambda yy, kw: sampling.get_ode_sampler(sde, self, y=yy, **kw)
        # (with minibatch the reference returns only the LAST chunk's samples, model.py:389; here all of them)
        return make(y, kwargs) if minibatch is None else self._chunked(make, y, minibatch, kwargs)

    def get_sb_sampler(self, sde, y, sampler_type="ode", N=None, **kwargs):
        """reference model.py:392-397 (the passed ``sde`` only supplies the default N; the model's own SDE is used)."""
        N = sde.N if N is None else N
        sde = self.sde.copy()
        sde.N = N if N is not None else sde.N
        return sampling.get_sb_sampler(sde, self, y=y, sampler_type=sampler_type, **kwargs)

    # -- audio helpers (reference model.py:411-424) -----------------------------------------------------------
    def to_audio(self, spec, length=None):
        return self._istft(self._backward_transform(spec), length)

    def _forward_transform(self, spec):
        return self.data_module.spec_fwd(spec)

    def _backward_transform(self, spec):
        return self.data_module.spec_back(spec)

    def _stft(self, sig):
        return self.data_module.stft(sig)

    def _istft(self, spec, length=None):
        return self.data_module.istft(spec, length)

    def _device(self):
        return next(self.dnn.parameters()).device

    def _sampler_for(self, Y, predictor, corrector, N, corrector_steps, snr, **kwargs):
        """The zero-argument sampler ``enhance`` runs for a padded spectrogram Y, chosen by the model's SDE and the SDE's
        ``sampler_type`` attribute as in reference model.py:440-452 (and enhancement.py:77-94)."""
        kind = type(self.sde).__name__
        if kind == "SBVESDE":
            return self.get_sb_sampler(sde=self.sde, y=Y, sampler_type=self.sde.sampler_type, **kwargs)
        if kind != "OUVESDE":
            raise ValueError("Invalid SDE type for speech enhancement: {}".format(kind))
        which = self.sde.sampler_type
        if which == "pc":
            return self.get_pc_sampler(predictor, corrector, Y, N=N, corrector_steps=corrector_steps, snr=snr, intermediate=False,
                                       **kwargs)
        if which == "ode":
            return self.get_ode_sampler(Y, N=N, **kwargs)
        raise ValueError("Invalid sampler type for SGMSE sampling: {}".format(which))

    def enhance(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="ald", N=30, corrector_steps=1, snr=0.5,
                timeit=False, **kwargs):
        """One-call enhancement of a noisy waveform y [1, L] (signature and result of reference model.py:426-465): peak
        normalisation -> STFT -> spec_fwd -> zero-pad -> sampler -> spec_back -> iSTFT -> undo the normalisation.  Returns a
        NumPy array; with ``timeit`` also nfe and the real-time factor.  (``sampler_type`` is accepted and, like in the
        reference, overridden by the SDE's own ``sampler_type``.)"""
        t_begin = time.time()
        n_samples = y.size(1)
        peak = y.abs().max().item()
        spec = self._forward_transform(self._stft((y / peak).to(self._device())))
        sampler = self._sampler_for(pad_spec(spec.unsqueeze(0)), predictor, corrector, N, corrector_steps, snr, **kwargs)
        sample, nfe = sampler()
        wave = (self.to_audio(sample.squeeze(), n_samples) * peak).squeeze().cpu().numpy()
        if not timeit:
            return wave
        return wave, nfe, (time.time() - t_begin) / (len(wave) / self.sr)

    def enhance_batch(self, y, N=30, corrector="ald", corrector_steps=1, snr=0.5, pad_mode="zero_pad", noise=None, seed=None,
                      predictor="reverse_diffusion", sampler_type="pc", use_graph=True, streams=None):
        """Batched form of enhancement.py:62-99: y float32 [B, L] (B utterances of equal length) -> enhanced float32 [B, L] on the
        model's device; or a LIST of 1-D waveforms of different lengths -> list of enhanced waveforms (one ragged batch, each
        utterance bit-identical to its own single call with the same seed and stream id).  Not in the reference (which loops
        files one by one)."""
        dev = self._device()
        if isinstance(y, (list, tuple)):
            # utterances of DIFFERENT lengths: one ragged batch (Context.set_frames) -- every utterance through its own STFT and
            # pad_spec, sampled together, inverted with its own length; returns a list of 1-D waveforms
            if noise is not None:
                raise ValueError("a list of waveforms (ragged batch) uses in-kernel noise only")
            ys = [w.reshape(1, -1).to(dev) for w in y]
            norms = [w.abs().max() for w in ys]
            Ys = [pad_spec(self._forward_transform(self._stft(w / n)).unsqueeze(1), mode=pad_mode)[0] for w, n in zip(ys, norms)]
            if sampler_type == "pc":
                sampler = self.get_pc_sampler(predictor, corrector, Ys, N=N, corrector_steps=corrector_steps, snr=snr, seed=seed,
                                              use_graph=use_graph, streams=streams)
            elif sampler_type == "ode":
                sampler = self.get_ode_sampler(Ys, N=N, seed=seed, use_graph=use_graph, streams=streams)
            else:
                raise ValueError(f"Sampler type {sampler_type} not supported")
            samples, nfe = sampler()
            return [self.to_audio(s, w.size(1))[0] * n for s, w, n in zip(samples, ys, norms)], nfe
        y = y.to(dev)
        T_orig = y.size(1)
        norm = y.abs().amax(dim=1, keepdim=True)
        Y = self._forward_transform(self._stft(y / norm)).unsqueeze(1)
        Y = pad_spec(Y, mode=pad_mode)
        if sampler_type == "pc":
            sampler = self.get_pc_sampler(predictor, corrector, Y, N=N, corrector_steps=corrector_steps, snr=snr, noise=noise,
                                          seed=seed, use_graph=use_graph, streams=streams)
        elif sampler_type == "ode":
            sampler = self.get_ode_sampler(Y, N=N, noise=noise, seed=seed, use_graph=use_graph, streams=streams)
        else:
            raise ValueError(f"Sampler type {sampler_type} not supported")
        sample, nfe = sampler()
        x_hat = self.to_audio(sample[:, 0], T_orig) * norm
        return x_hat, nfe
