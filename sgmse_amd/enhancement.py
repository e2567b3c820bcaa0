"""Directory-to-directory enhancement on MI355X: the job of the reference's ``enhancement.py`` (same command line,
``enhancement.py:20-31``) on top of the HIP library.

    python -m sgmse_amd.enhancement --test_dir noisy/ --enhanced_dir out/ --ckpt model.ckpt [--N 30 --snr 0.5 ...]

Differences from the reference script, none of them in the arithmetic of one utterance:

* files of equal length are enhanced together in batches of up to ``--batch_size``: utterances never interact -- the
  network and sampler arithmetic of an utterance is bit-identical in any batch (``test_full_size_batch_independence``),
  only its noise draws depend on its slot -- whereas the reference loops over files one by one (``enhancement.py:57``);
* under ``torchrun`` every rank takes a contiguous shard of the sorted file list (the split of the reference's validation
  loop, ``model.py:212-223``); there is no collective on the data path;
* audio I/O uses ``soundfile`` when it is installed and ``scipy.io.wavfile`` (wav only) otherwise; resampling to the
  model's rate uses ``scipy.signal.resample_poly`` (the reference: ``librosa.resample``).
"""
from __future__ import annotations

import glob
import os
from argparse import ArgumentParser
from math import gcd
from os.path import dirname, join
from typing import Dict, List, Tuple

import numpy as np
import torch

from .model import ScoreModel
from .parallel import shard_range
from .util.other import pad_spec


def read_audio(path: str) -> Tuple[np.ndarray, int]:
    """Mono float32 samples in [-1, 1] and the sample rate (first channel of multi-channel files, like
    ``torchaudio.load(...)[0]`` followed by the reference's single-channel handling)."""
    try:
        import soundfile
        x, sr = soundfile.read(path, dtype="float32", always_2d=True)
        return np.ascontiguousarray(x[:, 0]), int(sr)
    except ImportError:
        from scipy.io import wavfile
        sr, x = wavfile.read(path)
        if x.ndim > 1:
            x = x[:, 0]
        if x.dtype.kind == "i":
            x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
        elif x.dtype.kind == "u":
            x = (x.astype(np.float32) - 128.0) / 128.0
        return np.ascontiguousarray(x, dtype=np.float32), int(sr)


def write_audio(path: str, x: np.ndarray, sr: int) -> None:
    os.makedirs(dirname(path) or ".", exist_ok=True)
    try:
        import soundfile
        soundfile.write(path, x, sr)
    except ImportError:
        from scipy.io import wavfile
        wavfile.write(path, sr, x.astype(np.float32))


def list_audio(test_dir: str) -> List[str]:
    """Same globbing order as the reference (``enhancement.py:39-43``); scipy can only read the wav files."""
    files: List[str] = []
    for pat in ("*.wav", join("**", "*.wav"), "*.flac", join("**", "*.flac")):
        files += sorted(glob.glob(join(test_dir, pat)))
    return files


def model_audio_settings(model: ScoreModel) -> Tuple[int, str]:
    """Target sample rate and pad mode by backbone (``enhancement.py:46-54``)."""
    if model.backbone == "ncsnpp_48k":
        return 48000, "reflection"
    if model.backbone == "ncsnpp_v2":
        return 16000, "reflection"
    return 16000, "zero_pad"


def build_sampler(model: ScoreModel, Y: torch.Tensor, args, seed=None):
    """Sampler dispatch of ``enhancement.py:77-94``.  ``seed``: base of the in-kernel Philox noise stream (None: drawn from
    torch's generator, like the reference's unseeded ``randn_like``)."""
    sde = model.sde.__class__.__name__
    if sde == "OUVESDE":
        if args.sampler_type == "pc":
            return model.get_pc_sampler("reverse_diffusion", args.corrector, Y, N=args.N, corrector_steps=args.corrector_steps,
                                        snr=args.snr, seed=seed)
        if args.sampler_type == "ode":
            return model.get_ode_sampler(Y, N=args.N, seed=seed)
        raise ValueError(f"Sampler type {args.sampler_type} not supported")
    if sde == "SBVESDE":
        return model.get_sb_sampler(sde=model.sde, y=Y, sampler_type="ode" if args.sampler_type == "pc" else args.sampler_type,
                                    seed=seed)
    raise ValueError(f"SDE {sde} not supported")


def enhance_files(model: ScoreModel, files: List[str], test_dir: str, enhanced_dir: str, args, device) -> int:
    target_sr, pad_mode = model_audio_settings(model)
    # load, resample, normalise (enhancement.py:62-72) and bucket by length
    buckets: Dict[int, List[Tuple[str, torch.Tensor, float, int]]] = {}
    for path in files:
        y, sr = read_audio(path)
        if sr != target_sr:
            from scipy.signal import resample_poly
            g = gcd(sr, target_sr)
            y = resample_poly(y, target_sr // g, sr // g).astype(np.float32)
        norm = float(np.abs(y).max()) or 1.0
        name = path.replace(test_dir, "")
        name = name[1:] if name.startswith("/") else name
        buckets.setdefault(len(y), []).append((name, torch.from_numpy(y / norm), norm, len(y)))
    done = 0
    nbatch = 0
    for length, items in sorted(buckets.items()):
        for i in range(0, len(items), args.batch_size):
            chunk = items[i:i + args.batch_size]
            seed = None if args.seed is None else args.seed + 1000003 * nbatch
            nbatch += 1
            y = torch.stack([c[1] for c in chunk]).to(device)                     # [B, L], equal length
            Y = pad_spec(model._forward_transform(model._stft(y)).unsqueeze(1), mode=pad_mode)
            sample, _ = build_sampler(model, Y, args, seed)()
            x_hat = model.to_audio(sample[:, 0], length).cpu().numpy()            # spec_back + iSTFT (enhancement.py:99)
            for (name, _, norm, _), x in zip(chunk, x_hat):
                write_audio(join(enhanced_dir, name), x * norm, target_sr)        # renormalise (enhancement.py:102)
                done += 1
    return done


def main(argv=None) -> int:
    parser = ArgumentParser()
    parser.add_argument("--test_dir", type=str, required=True, help="Directory containing the test data")
    parser.add_argument("--enhanced_dir", type=str, required=True, help="Directory containing the enhanced data")
    parser.add_argument("--ckpt", type=str, help="Path to model checkpoint")
    parser.add_argument("--sampler_type", type=str, default="pc", help="Sampler type for the PC sampler.")
    parser.add_argument("--corrector", type=str, choices=("ald", "langevin", "none"), default="ald", help="Corrector class for the PC sampler.")
    parser.add_argument("--corrector_steps", type=int, default=1, help="Number of corrector steps")
    parser.add_argument("--snr", type=float, default=0.5, help="SNR value for (annealed) Langevin dynmaics")
    parser.add_argument("--N", type=int, default=30, help="Number of reverse steps")
    parser.add_argument("--device", type=str, default="cuda", help="Device to use for inference")
    parser.add_argument("--t_eps", type=float, default=0.03, help="The minimum process time (0.03 by default)")
    parser.add_argument("--batch_size", type=int, default=32, help="Utterances of equal padded length enhanced together")
    parser.add_argument("--seed", type=int, default=None, help="Base seed of the sampler noise (default: unseeded, like the reference)")
    args = parser.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device(args.device)
    if device.type == "cuda":
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
    model = ScoreModel.load_from_checkpoint(args.ckpt, map_location=device)      # every rank reads the checkpoint itself
    model.t_eps = args.t_eps
    model.eval()
    model.to(device)

    files = list_audio(args.test_dir)
    lo, hi = shard_range(len(files), rank, world)
    n = enhance_files(model, files[lo:hi], args.test_dir, args.enhanced_dir, args, device)
    print(f"[rank {rank}/{world}] enhanced {n} of {len(files)} files into {args.enhanced_dir}")
    return n


if __name__ == "__main__":
    main()
