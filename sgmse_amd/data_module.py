"""Signal front-end of the enhancement path (reference sgmse/data_module.py:13-19,103-236, front-end methods only).

``SpecsDataModule`` keeps the reference's constructor signature and the methods the inference path uses
(``stft``, ``istft``, ``spec_fwd``, ``spec_back``, ``stft_kwargs``, ``istft_kwargs``, ``_get_window``); every transform
runs in the HIP library.  Dataset / dataloader functionality (training) is out of scope and raises.  The class is
importable under this name because Lightning checkpoints pickle it inside ``hyper_parameters`` (model.py:87-88)."""
import torch

from . import ops


def get_window(window_type, window_length):
    if window_type == "sqrthann":
        return torch.sqrt(torch.hann_window(window_length, periodic=True))
    elif window_type == "hann":
        return torch.hann_window(window_length, periodic=True)
    else:
        raise NotImplementedError(f"Window type {window_type} not implemented!")


class SpecsDataModule:
    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--base_dir", type=str, default=None)
        parser.add_argument("--format", type=str, default="default")
        parser.add_argument("--batch_size", type=int, default=8)
        parser.add_argument("--n_fft", type=int, default=510)
        parser.add_argument("--hop_length", type=int, default=128)
        parser.add_argument("--num_frames", type=int, default=256)
        parser.add_argument("--window", type=str, choices=("sqrthann", "hann"), default="hann")
        parser.add_argument("--num_workers", type=int, default=4)
        parser.add_argument("--dummy", action="store_true")
        parser.add_argument("--spec_factor", type=float, default=0.15)
        parser.add_argument("--spec_abs_exponent", type=float, default=0.5)
        parser.add_argument("--normalize", type=str, choices=("clean", "noisy", "not"), default="noisy")
        parser.add_argument("--transform_type", type=str, choices=("exponent", "log", "none"), default="exponent")
        return parser

    def __init__(self, base_dir=None, format="default", batch_size=8, n_fft=510, hop_length=128, num_frames=256,
                 window="hann", num_workers=4, dummy=False, spec_factor=0.15, spec_abs_exponent=0.5, gpu=True,
                 normalize="noisy", transform_type="exponent", **kwargs):
        self.base_dir, self.format, self.batch_size = base_dir, format, batch_size
        self.n_fft, self.hop_length, self.num_frames = n_fft, hop_length, num_frames
        self.window = get_window(window, self.n_fft)
        self.windows = {}
        self.num_workers, self.dummy, self.gpu = num_workers, dummy, gpu
        self.spec_factor, self.spec_abs_exponent = spec_factor, spec_abs_exponent
        self.normalize, self.transform_type = normalize, transform_type
        self.kwargs = kwargs

    # -- training-side API: out of scope ---------------------------------------------------------------------
    def setup(self, stage=None):
        raise NotImplementedError("dataset loading is outside the MI355X hot path (inference front-end only)")

    train_dataloader = val_dataloader = test_dataloader = setup

    # -- front-end ------------------------------------------------------------------------------------------------
    def spec_fwd(self, spec):
        if self.transform_type == "none":
            return spec
        return ops.spec_transform(spec, self.transform_type, self.spec_factor, self.spec_abs_exponent, inverse=False)

    def spec_back(self, spec):
        if self.transform_type == "none":
            return spec
        return ops.spec_transform(spec, self.transform_type, self.spec_factor, self.spec_abs_exponent, inverse=True)

    @property
    def stft_kwargs(self):
        return {**self.istft_kwargs, "return_complex": True}

    @property
    def istft_kwargs(self):
        return dict(n_fft=self.n_fft, hop_length=self.hop_length, window=self.window, center=True)

    def _get_window(self, x):
        """The analysis window on x's device (reference data_module.py:201-210)."""
        window = self.windows.get(x.device, None)
        if window is None:
            window = self.window.to(x.device)
            self.windows[x.device] = window
        return window

    def stft(self, sig):
        return ops.stft(sig, self.n_fft, self.hop_length, self._get_window(sig))

    def istft(self, spec, length=None):
        return ops.istft(spec, self.n_fft, self.hop_length, self._get_window(spec), length=length)
