"""ctypes binding of libsgmse_hip.so (C ABI: include/sgmse_hip.h).

The product path has exactly one implementation: the HIP library built for gfx950 by ``__graft_entry__.build()`` /
``make -C sgmse_amd/csrc``.  If it is missing, or no GPU is visible, every entry point fails loudly -- there is no
PyTorch or CPU fallback.  (The test-suite can point ``load_library`` at the CPU workgroup *emulator* build of the
same kernel sources, ``tests/emu/libsgmse_emu.so``, to check kernel logic in a GPU-less container; that library
reports a different ``backend`` string and is never loaded implicitly.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libsgmse_hip.so")

SGMSE_NCLASS = 8
CLASS_NAMES = ("conv3x3_wide", "conv3x3_other", "conv1x1", "conv_direct", "groupnorm_stats", "fir",
               "attention", "entry_exit")
CLASS_WORK_UNIT = ("flop", "flop", "flop", "flop", "byte", "byte", "flop", "byte")


class NetCfgC(C.Structure):
    _fields_ = [("variant", C.c_int), ("nf", C.c_int), ("n_levels", C.c_int), ("ch_mult", C.c_int * 8),
                ("num_res_blocks", C.c_int), ("n_attn", C.c_int), ("attn_res", C.c_int * 8), ("image_size", C.c_int),
                ("progressive", C.c_int), ("progressive_input", C.c_int), ("scale_by_sigma", C.c_int)]


class SamplerCfgC(C.Structure):
    _fields_ = [("N", C.c_int), ("corrector", C.c_int), ("corrector_steps", C.c_int), ("predictor", C.c_int),
                ("probability_flow", C.c_int), ("denoise", C.c_int), ("theta", C.c_float), ("std1", C.c_float),
                ("t", C.POINTER(C.c_float)), ("dt", C.POINTER(C.c_float)), ("ald_eps", C.POINTER(C.c_float)),
                ("ald_noise", C.POINTER(C.c_float)), ("G", C.POINTER(C.c_float)), ("G2", C.POINTER(C.c_float)),
                ("in_scale", C.POINTER(C.c_float)), ("score_alpha", C.POINTER(C.c_float)), ("score_beta", C.POINTER(C.c_float)),
                ("snr", C.c_float), ("use_graph", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_LL = C.c_longlong
_SIGS = {
    "sgmse_ctx_create": (_I, [_I, _P, C.POINTER(_P)]),
    "sgmse_ctx_destroy": (None, [_P]),
    "sgmse_set_stream": (_I, [_P, _P]),
    "sgmse_sync": (_I, [_P]),
    "sgmse_last_error": (C.c_char_p, [_P]),
    "sgmse_backend": (C.c_char_p, []),
    "sgmse_configure": (_I, [_P, C.POINTER(NetCfgC)]),
    "sgmse_load_weights": (_I, [_P, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_LL), _I, _I]),
    "sgmse_param_count": (_I, [_P, C.POINTER(_LL)]),
    "sgmse_ncsnpp_forward": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "sgmse_pc_sample": (_I, [_P, _P, _P, _I, _I, _I, C.POINTER(SamplerCfgC), _P, C.c_ulonglong, C.POINTER(_I)]),
    "sgmse_sb_sample": (_I, [_P, _P, _P, _I, _I, _I, _I] + [C.POINTER(_F)] * 8 + [_I, _P, C.c_ulonglong, _I, C.POINTER(_I)]),
    "sgmse_stft": (_I, [_P, _P, _P, _P, _I, _I, _I, _I]),
    "sgmse_istft": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "sgmse_spec_fwd": (_I, [_P, _P, _P, _LL, _I, _F, _F]),
    "sgmse_spec_back": (_I, [_P, _P, _P, _LL, _I, _F, _F]),
    "sgmse_upfirdn2d": (_I, [_P, _P, _P, _P] + [_I] * 13),
    "sgmse_upfirdn2d_dtype": (_I, [_P, _I, _P, _P, _P] + [_I] * 13),
    "sgmse_op_conv2d": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P, _I, _P, _I]),
    "sgmse_op_groupnorm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I]),
    "sgmse_conv_split_mode": (_I, [_P, _P]),
    "sgmse_conv_winograd": (_I, [_P, _P]),
    "sgmse_op_fir": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "sgmse_op_attention": (_I, [_P, _P, _P, _I, _I, _I]),
    "sgmse_profile_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, C.POINTER(_F), C.POINTER(C.c_double), C.POINTER(_I)]),
    "sgmse_bench_conv": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
    "sgmse_calib_stream": (_I, [_P, _I, _I, C.c_longlong, C.POINTER(_F)]),
    "sgmse_arena_bytes": (_I, [_P, C.POINTER(_LL)]),
    "sgmse_graph_captures": (_I, [_P, C.POINTER(_I)]),
    "sgmse_graph_updates": (_I, [_P, C.POINTER(_I)]),
    "sgmse_set_noise_streams": (_I, [_P, C.POINTER(C.c_ulonglong), _I]),
    "sgmse_set_frames": (_I, [_P, C.POINTER(_I), _I]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None
_lib_path: Optional[str] = None


class SgmseLibraryError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load (once) and return the shared library.  ``path`` defaults to the in-tree HIP build."""
    global _lib, _lib_path
    with _lock:
        want = os.path.abspath(path or DEFAULT_LIB)
        if _lib is not None:
            if path is None or want == _lib_path:
                return _lib
            raise SgmseLibraryError(f"a different sgmse library is already loaded: {_lib_path}")
        if not os.path.exists(want):
            raise SgmseLibraryError(
                f"{want} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                f"or make -C sgmse_amd/csrc).  sgmse_amd has no CPU/PyTorch fallback.")
        lib = C.CDLL(want)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib, _lib_path = lib, want
        return lib


def backend() -> str:
    return load_library().sgmse_backend().decode()


def is_emulator() -> bool:
    return backend().startswith("cpu-emulator")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def check_tensor(t: torch.Tensor, name: str, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.device != device:
        raise ValueError(f"{name}: expected device {device}, got {t.device}")
    return t if t.is_contiguous() else t.contiguous()


def _ragged_geometry(items, planes: int, what: str):
    """Frame counts and the common F of a ragged batch: every element [planes,F,T_b] (or [F,T_b] when planes == 1).  The packed
    buffer and the engine's own size arithmetic (sum F*T_b) must agree, so mismatching shapes are refused here."""
    if len(items) == 0:
        raise ValueError(f"ragged batch: empty list of {what}")
    F_ = int(items[0].shape[-2]) if items[0].dim() >= 2 else -1
    frames = []
    for i, v in enumerate(items):
        ok = v.dim() in ((2, 3) if planes == 1 else (3,)) and int(v.shape[-2]) == F_ and v.numel() == planes * F_ * int(v.shape[-1])
        if not ok:
            raise ValueError(f"ragged batch: {what}[{i}] has shape {tuple(v.shape)}; every utterance must be "
                             f"[{planes},{F_},T_b]" + (" or [F,T_b]" if planes == 1 else ""))
        frames.append(int(v.shape[-1]))
    return frames, F_


class Context:
    """One sgmse_ctx: a device, a stream, the weights and the activation arena."""

    def __init__(self, device: "torch.device | str | int | None" = None):
        lib = load_library()
        self.lib = lib
        emu = is_emulator()
        if device is None:
            device = "cpu" if emu else "cuda"
        device = torch.device(device)
        if emu:
            if device.type != "cpu":
                raise SgmseLibraryError("the emulator build only works on CPU tensors")
            self.device = device
            dev_index, stream = 0, None
        else:
            if device.type != "cuda":
                raise SgmseLibraryError("sgmse_amd runs on an AMD GPU only (device must be 'cuda'); there is no CPU path")
            if not torch.cuda.is_available():
                raise SgmseLibraryError("no GPU visible to PyTorch-ROCm; sgmse_amd has no CPU fallback")
            dev_index = device.index if device.index is not None else torch.cuda.current_device()
            self.device = torch.device("cuda", dev_index)
            stream = torch.cuda.current_stream(self.device).cuda_stream
        h = _P()
        rc = lib.sgmse_ctx_create(dev_index, stream, C.byref(h))
        if rc != 0 or not h:
            raise SgmseLibraryError(f"sgmse_ctx_create failed ({rc})")
        self.h = h
        self._keep: Dict[str, object] = {}
        self._frames: list = []          # frame table in force in the library (set_frames); [] = uniform batches

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.sgmse_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def check(self, rc: int):
        if rc == 0:
            return
        msg = self.lib.sgmse_last_error(self.h).decode(errors="replace")
        if rc == -1:
            raise ValueError(msg)
        raise RuntimeError(f"sgmse_hip error {rc}: {msg}")

    def use_current_stream(self):
        if self.device.type == "cuda":
            self.check(self.lib.sgmse_set_stream(self.h, torch.cuda.current_stream(self.device).cuda_stream))

    def sync(self):
        self.check(self.lib.sgmse_sync(self.h))

    # -- model ------------------------------------------------------------------------------------------------
    def configure(self, *, variant: str, nf: int, ch_mult: Sequence[int], num_res_blocks: int,
                  attn_resolutions: Sequence[int], image_size: int, progressive: str, progressive_input: str,
                  scale_by_sigma: bool = True):
        c = NetCfgC()
        variants = {"ncsnpp": 0, "ncsnpp_48k": 1}
        if variant not in variants:
            raise ValueError(f"unsupported backbone variant {variant!r}")
        prog = {"none": 0, "output_skip": 1}
        pin = {"none": 0, "input_skip": 1}
        if progressive not in prog:
            raise ValueError(f"progressive={progressive!r} is not supported by the HIP backbone (none|output_skip)")
        if progressive_input not in pin:
            raise ValueError(f"progressive_input={progressive_input!r} is not supported by the HIP backbone (none|input_skip)")
        if len(ch_mult) > 8 or len(attn_resolutions) > 8:
            raise ValueError("at most 8 levels / attention resolutions")
        c.variant = variants[variant]; c.nf = nf; c.n_levels = len(ch_mult)
        for i, v in enumerate(ch_mult):
            c.ch_mult[i] = int(v)
        c.num_res_blocks = num_res_blocks; c.n_attn = len(attn_resolutions)
        for i, v in enumerate(attn_resolutions):
            c.attn_res[i] = int(v)
        c.image_size = image_size; c.progressive = prog[progressive]; c.progressive_input = pin[progressive_input]
        c.scale_by_sigma = int(bool(scale_by_sigma))
        self.check(self.lib.sgmse_configure(self.h, C.byref(c)))

    def load_weights(self, state: Dict[str, torch.Tensor]):
        """state: reference state_dict of the backbone (fp32).  Host tensors are copied synchronously; device
        tensors (e.g. after an RCCL broadcast) are copied device-to-device."""
        names = list(state.keys())
        on_dev = all(v.device.type == "cuda" for v in state.values())
        tens = []
        for k in names:
            v = state[k].detach()
            if v.dtype != torch.float32:
                v = v.float()
            if not on_dev and v.device.type != "cpu":
                v = v.cpu()
            tens.append(v.contiguous())
        n = len(names)
        c_names = (C.c_char_p * n)(*[s.encode() for s in names])
        c_ptrs = (_P * n)(*[t.data_ptr() for t in tens])
        c_num = (_LL * n)(*[t.numel() for t in tens])
        self.use_current_stream()
        self.check(self.lib.sgmse_load_weights(self.h, c_names, c_ptrs, c_num, n, int(on_dev)))
        if on_dev:
            self.sync()

    def param_count(self) -> int:
        out = _LL(0)
        self.check(self.lib.sgmse_param_count(self.h, C.byref(out)))
        return out.value

    def forward(self, xy: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """NCSNpp.forward: xy complex64 [B,2,F,T], t float32 [B] -> complex64 [B,1,F,T]."""
        if xy.dim() != 4 or xy.shape[1] != 2:
            raise ValueError(f"expected x of shape [B,2,F,T], got {tuple(xy.shape)}")
        xy = check_tensor(xy, "x", torch.complex64, self.device)
        t = check_tensor(t.reshape(-1), "time_cond", torch.float32, self.device)
        B, _, F_, T = xy.shape
        if t.numel() != B:
            raise ValueError("time_cond must have one entry per batch element")
        out = torch.empty((B, 1, F_, T), dtype=torch.complex64, device=self.device)
        self.set_frames([])
        self.use_current_stream()
        self.check(self.lib.sgmse_ncsnpp_forward(self.h, xy.data_ptr(), t.data_ptr(), out.data_ptr(), B, F_, T))
        return out

    def pc_sample(self, Y: torch.Tensor, table: Dict[str, torch.Tensor], *, theta: float, std1: float,
                  corrector: str, corrector_steps: int, predictor: str, probability_flow: bool, denoise: bool,
                  noise: Optional[torch.Tensor], seed: int, use_graph: bool = True, affine=None, snr: float = 0.0, streams=None):
        """``affine``: optional (in_scale, score_alpha, score_beta) fp32 tensors of length N: the score wrapper of
        ScoreModel.forward's new-code branch (ncsnpp_v2); None = old-code branch (score = -F)."""
        frames = None
        if isinstance(Y, (list, tuple)):       # ragged batch: utterances [F,T_b] (or [1,F,T_b]) of different lengths, see set_frames
            if noise is not None:
                raise ValueError("ragged batches use in-kernel noise only")
            frames, F_ = _ragged_geometry(Y, 1, "y")
            B, T = len(Y), max(frames)
            Y = torch.cat([check_tensor(y, "y", torch.complex64, self.device).reshape(-1) for y in Y])
        else:
            Y = check_tensor(Y, "y", torch.complex64, self.device)
            if Y.dim() != 4 or Y.shape[1] != 1:
                raise ValueError(f"expected y of shape [B,1,F,T], got {tuple(Y.shape)}")
            B, _, F_, T = Y.shape
        N = int(table["t"].numel())
        cfg = SamplerCfgC()
        cfg.N = N
        cfg.corrector = {"none": 0, "ald": 1, "langevin": 2}[corrector]
        cfg.snr = float(snr)
        cfg.corrector_steps = int(corrector_steps)
        cfg.predictor = {"none": 0, "reverse_diffusion": 1}[predictor]
        cfg.probability_flow = int(bool(probability_flow)); cfg.denoise = int(bool(denoise))
        cfg.theta = float(theta); cfg.std1 = float(std1); cfg.use_graph = int(bool(use_graph))
        keep = {}
        for k in ("t", "dt", "ald_eps", "ald_noise", "G", "G2"):
            v = table[k].detach().to("cpu", torch.float32).contiguous()
            keep[k] = v
            setattr(cfg, k, C.cast(v.data_ptr(), C.POINTER(C.c_float)))
        if affine is not None:
            for k, v in zip(("in_scale", "score_alpha", "score_beta"), affine):
                v = torch.as_tensor(v, dtype=torch.float32).detach().to("cpu").expand(N).contiguous()
                keep[k] = v
                setattr(cfg, k, C.cast(v.data_ptr(), C.POINTER(C.c_float)))
        if noise is not None:
            noise = check_tensor(noise, "noise", torch.complex64, self.device)
            ncorr = cfg.corrector_steps if cfg.corrector else 0
            need = 1 + N * (ncorr + (1 if (cfg.predictor == 1 and not probability_flow) else 0))
            if noise.shape[0] < need or tuple(noise.shape[1:]) != tuple(Y.shape):
                raise ValueError(f"noise must be [{need},{B},1,{F_},{T}] complex64, got {tuple(noise.shape)}")
        out = torch.empty_like(Y)
        nfe = _I(0)
        if streams is not None:
            if len(streams) != B:
                raise ValueError(f"streams must name the {B} utterances of the batch, got {len(streams)}")
            self.set_noise_streams(streams)

        def run():
            self.use_current_stream()
            # the frame table stays in force between calls: consecutive ragged batches of one composition re-use the arena plan
            # and the captured graph; a uniform call switches it off
            self.set_frames(frames if frames is not None else [])
            self.check(self.lib.sgmse_pc_sample(self.h, Y.data_ptr(), out.data_ptr(), B, F_, T, C.byref(cfg), ptr(noise),
                                                C.c_ulonglong(seed & (2 ** 64 - 1)), C.byref(nfe)))

        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if cur is not None and use_graph and cur.cuda_stream == 0:
            # stream capture is not allowed on the legacy default stream: run the sampler on a side stream
            side = self._keep.get("side_stream")
            if side is None:
                side = self._keep["side_stream"] = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                run()
            cur.wait_stream(side)
        else:
            run()
        self._keep["sampler"] = (noise, keep)   # the captured graph refers to the replayed-noise buffer
        if frames is not None:                  # unpack: one [1,F,T_b] tensor per utterance
            outs, o = [], 0
            for T_b in frames:
                outs.append(out[o:o + F_ * T_b].reshape(1, F_, T_b))
                o += F_ * T_b
            return outs, nfe.value
        return out, nfe.value

    def sb_sample(self, Y: torch.Tensor, table: Dict[str, torch.Tensor], *, stochastic: bool, noise: Optional[torch.Tensor],
                  seed: int, affine=None, use_graph: bool = True, streams=None):
        """Schroedinger-bridge sampler; table: fp32 tensors t, w_prev, w_est, w_y, w_z of length N."""
        Y = check_tensor(Y, "y", torch.complex64, self.device)
        B, _, F_, T = Y.shape
        N = int(table["t"].numel())
        fp = lambda v: C.cast(v.data_ptr(), C.POINTER(_F))
        keep = [table[k].detach().to("cpu", torch.float32).contiguous() for k in ("t", "w_prev", "w_est", "w_y", "w_z")]
        aff = [None, None, None]
        if affine is not None:
            aff = [torch.as_tensor(v, dtype=torch.float32).detach().to("cpu").expand(N).contiguous() for v in affine]
        if noise is not None:
            noise = check_tensor(noise, "noise", torch.complex64, self.device)
            if noise.shape[0] < N or tuple(noise.shape[1:]) != tuple(Y.shape):
                raise ValueError(f"noise must be [{N},{B},1,{F_},{T}] complex64, got {tuple(noise.shape)}")
        out = torch.empty_like(Y)
        nfe = _I(0)
        if streams is not None:
            if len(streams) != B:
                raise ValueError(f"streams must name the {B} utterances of the batch, got {len(streams)}")
            self.set_noise_streams(streams)

        def run():
            self.use_current_stream()
            self.set_frames([])
            self.check(self.lib.sgmse_sb_sample(self.h, Y.data_ptr(), out.data_ptr(), B, F_, T, N, *[fp(v) for v in keep],
                                                *[None if v is None else fp(v) for v in aff], int(bool(stochastic)), ptr(noise),
                                                C.c_ulonglong(seed & (2 ** 64 - 1)), int(bool(use_graph)), C.byref(nfe)))

        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if cur is not None and use_graph and cur.cuda_stream == 0:
            side = self._keep.get("side_stream")
            if side is None:
                side = self._keep["side_stream"] = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                run()
            cur.wait_stream(side)
        else:
            run()
        self._keep["sb"] = (noise, keep, aff)
        return out, nfe.value

    def profile_forward(self, xy, t: torch.Tensor):
        """Per-kernel-class HIP-event timing of ONE network evaluation (eager).  xy: complex64 [B,2,F,T], or a list of [2,F,T_b]
        (a ragged batch, as forward_ragged takes it)."""
        t = check_tensor(t, "t", torch.float32, self.device)
        frames = []
        if isinstance(xy, (list, tuple)):
            frames, F_ = _ragged_geometry(xy, 2, "x")
            B, T = len(xy), max(frames)
            xy = torch.cat([check_tensor(x, "x", torch.complex64, self.device).reshape(-1) for x in xy])
            out = torch.empty(F_ * sum(frames), dtype=torch.complex64, device=self.device)
        else:
            xy = check_tensor(xy, "x", torch.complex64, self.device)
            B, _, F_, T = xy.shape
            out = torch.empty((B, 1, F_, T), dtype=torch.complex64, device=self.device)
        ms = (_F * SGMSE_NCLASS)()
        fl = (C.c_double * SGMSE_NCLASS)()
        nl = (_I * SGMSE_NCLASS)()
        self.use_current_stream()
        self.set_frames(frames)
        self.check(self.lib.sgmse_profile_forward(self.h, xy.data_ptr(), t.data_ptr(), out.data_ptr(), B, F_, T, ms, fl, nl))
        return {n: {"ms": ms[i], "work": fl[i], "unit": CLASS_WORK_UNIT[i], "launches": nl[i]}
                for i, n in enumerate(CLASS_NAMES)}, out

    def bench_conv(self, ks, B, Cin, Cout, H, W, variant=-1, iters=10, fused=False) -> float:
        ms = _F(0)
        self.use_current_stream()
        self.check(self.lib.sgmse_bench_conv(self.h, ks, B, Cin, Cout, H, W, variant, iters, int(fused), C.byref(ms)))
        return ms.value

    def calib_stream(self, mode: int, bytes_per_lane: int, total_bytes: int) -> float:
        """Measurement only: one launch of a known-size coalesced read (mode 0) / write (mode 1) stream; ms of the launch."""
        ms = _F(0)
        self.use_current_stream()
        self.check(self.lib.sgmse_calib_stream(self.h, int(mode), int(bytes_per_lane), int(total_bytes), C.byref(ms)))
        return ms.value

    def conv_winograd(self) -> bool:
        """True when the wide levels' 3x3 layers run on the Winograd F(2,3) x fp16x2 kernel (kernels_conv_wino.h)."""
        out = C.c_int(0)
        self.check(self.lib.sgmse_conv_winograd(self.h, C.byref(out)))
        return bool(out.value)

    def conv_split_mode(self) -> int:
        """0: fp32 MFMA, 1: bf16x3 split, 2: fp16x2 split on the wide 3x3 layers (kernels_conv_split.h)."""
        out = C.c_int(0)
        self.check(self.lib.sgmse_conv_split_mode(self.h, C.byref(out)))
        return out.value

    def set_noise_streams(self, ids) -> None:
        """Noise-stream ids (one per utterance) for the next sampler call with in-kernel noise: see sgmse_set_noise_streams."""
        ids = [int(v) & (2 ** 64 - 1) for v in (ids.tolist() if hasattr(ids, "tolist") else ids)]
        arr = (C.c_ulonglong * len(ids))(*ids)
        self.check(self.lib.sgmse_set_noise_streams(self.h, arr, len(ids)))

    def set_frames(self, frames) -> None:
        """Ragged batches (sgmse_set_frames): frame count of every utterance of the following calls; [] = uniform batches again."""
        frames = [int(v) for v in frames]
        if frames == self._frames:            # unchanged: the library keeps its tables, arena plan and captured graph
            return
        arr = (_I * max(len(frames), 1))(*frames)
        self.check(self.lib.sgmse_set_frames(self.h, arr if frames else None, len(frames)))
        self._frames = frames

    def forward_ragged(self, xys, t: torch.Tensor):
        """NCSNpp.forward on utterances of different lengths in ONE batch: xys = list of complex64 [2,F,T_b]; returns the list of
        complex64 [1,F,T_b], each bit-identical to ``forward(xy[None], t[b:b+1])[0]``."""
        frames, F_ = _ragged_geometry(xys, 2, "x")
        packed = torch.cat([check_tensor(x, "x", torch.complex64, self.device).reshape(-1) for x in xys])
        t = check_tensor(t.reshape(-1), "time_cond", torch.float32, self.device)
        if t.numel() != len(xys):
            raise ValueError("time_cond must have one entry per utterance")
        out = torch.empty(F_ * sum(frames), dtype=torch.complex64, device=self.device)
        self.set_frames(frames)
        self.use_current_stream()
        self.check(self.lib.sgmse_ncsnpp_forward(self.h, packed.data_ptr(), t.data_ptr(), out.data_ptr(), len(xys), F_, max(frames)))
        outs, o = [], 0
        for T_b in frames:
            outs.append(out[o:o + F_ * T_b].reshape(1, F_, T_b))
            o += F_ * T_b
        return outs

    def graph_captures(self) -> int:
        """Number of hipGraph captures (+ instantiations) of a sampler step this context has done."""
        out = _I(0)
        self.check(self.lib.sgmse_graph_captures(self.h, C.byref(out)))
        return out.value

    def graph_updates(self) -> int:
        """Number of times a captured step was updated in place (hipGraphExecUpdate) instead of being instantiated anew."""
        out = _I(0)
        self.check(self.lib.sgmse_graph_updates(self.h, C.byref(out)))
        return out.value

    def arena_bytes(self) -> int:
        out = _LL(0)
        self.check(self.lib.sgmse_arena_bytes(self.h, C.byref(out)))
        return out.value
