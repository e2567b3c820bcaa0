"""Directory-to-directory enhancement on MI355X: the job of the reference's ``enhancement.py`` (same command line,
``enhancement.py:20-31``) on top of the HIP library.

    python -m sgmse_amd.enhancement --test_dir noisy/ --enhanced_dir out/ --ckpt model.ckpt [--N 30 --snr 0.5 ...]

Differences from the reference script, none of them in the arithmetic of one utterance:

* files whose PADDED spectrograms have the same number of frames (``pad_spec``: next multiple of 64, i.e. 0.5 s steps at
  16 kHz) are enhanced together in batches of up to ``--batch_size``, each after its own STFT and padding: utterances never
  interact -- the network and sampler arithmetic of an utterance is bit-identical in any batch
  (``test_full_size_batch_independence``), only its noise draws depend on its slot -- whereas the reference loops over files
  one by one (``enhancement.py:57``);
* with ``--ragged`` files of DIFFERENT padded lengths share a batch too (``sgmse_set_frames``: every kernel addresses an
  utterance's block with the utterance's own row stride, so its arithmetic is still that of its single-file run, bit for bit):
  full batches of one padded length still run as uniform batches, the leftovers of all lengths are pooled, sorted by length and
  cut into ragged batches of ``--batch_size`` (a ragged batch costs 1.16 x a uniform one per frame, a small uniform batch more);
* under ``torchrun`` every rank takes a contiguous shard of the sorted file list (the split of the reference's validation
  loop, ``model.py:212-223``), rank 0 reads the checkpoint and the weights reach the other ranks in one RCCL broadcast; there
  is no collective on the data path;
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


def build_sampler(model: ScoreModel, Y: torch.Tensor, args, seed=None, streams=None):
    """Sampler dispatch of ``enhancement.py:77-94``.  ``seed``: base of the in-kernel Philox noise stream (None: drawn from
    torch's generator, like the reference's unseeded ``randn_like``); ``streams``: per-utterance noise-stream ids."""
    sde = model.sde.__class__.__name__
    if sde == "OUVESDE":
        if args.sampler_type == "pc":
            return model.get_pc_sampler("reverse_diffusion", args.corrector, Y, N=args.N, corrector_steps=args.corrector_steps,
                                        snr=args.snr, seed=seed, streams=streams)
        if args.sampler_type == "ode":
            return model.get_ode_sampler(Y, N=args.N, seed=seed, streams=streams)
        raise ValueError(f"Sampler type {args.sampler_type} not supported")
    if sde == "SBVESDE":
        return model.get_sb_sampler(sde=model.sde, y=Y, sampler_type="ode" if args.sampler_type == "pc" else args.sampler_type,
                                    seed=seed, streams=streams)
    raise ValueError(f"SDE {sde} not supported")


def _load_normalised(path: str, target_sr: int) -> Tuple[torch.Tensor, float]:
    """File -> peak-normalised mono waveform at the model's rate and its peak (enhancement.py:62-72)."""
    y, sr = read_audio(path)
    if sr != target_sr:
        from scipy.signal import resample_poly
        g = gcd(sr, target_sr)
        y = resample_poly(y, target_sr // g, sr // g).astype(np.float32)
    peak = float(np.abs(y).max()) or 1.0
    return torch.from_numpy(y / peak), peak


def enhance_files(model: ScoreModel, files: List[str], test_dir: str, enhanced_dir: str, args, device, first_index: int = 0) -> int:
    """Enhance ``files`` in batches.  What has to agree inside a batch is the PADDED spectrogram shape, not the waveform
    length: every utterance goes through its own STFT and its own ``pad_spec`` (zero / reflection padding to the next multiple
    of 64 frames, exactly what the reference does per file), utterances with the same padded frame count are stacked, sampled
    together and inverted one by one with their own lengths.  A corpus of mixed lengths therefore runs in batches (64 frames =
    0.5 s at 16 kHz granularity) while every utterance keeps the arithmetic of its single-file run bit for bit.  With ``--seed``
    the noise of an utterance is a function of (seed, its index in the sorted file list): the enhanced files do not depend on
    the batch size, the bucket an utterance landed in or the number of ranks."""
    target_sr, pad_mode = model_audio_settings(model)
    hop = model.data_module.hop_length
    by_frames: Dict[int, List[str]] = {}
    lengths: Dict[str, int] = {}
    for path in files:                                   # pass 1: lengths only (headers would do; files are small)
        y, _ = _load_normalised(path, target_sr)
        lengths[path] = len(y)
        frames = len(y) // hop + 1
        by_frames.setdefault((frames + 63) // 64 * 64, []).append(path)
    index = {path: first_index + k for k, path in enumerate(files)}
    ragged = bool(getattr(args, "ragged", False))
    if ragged and (model.sde.__class__.__name__ != "OUVESDE" or args.corrector == "langevin"):
        import warnings
        warnings.warn("--ragged needs an OUVE model and the 'ald' / 'none' corrector (the Langevin corrector couples the utterances of "
                      "a batch; the Schroedinger-bridge sampler is not built for it): batching by padded length instead")
        ragged = False
    if ragged:
        # full batches of one padded length run as uniform batches (the fastest form); what is left over of every length is pooled,
        # shortest first (neighbours in a batch have similar lengths), and cut into ragged batches of any mix of lengths
        batches, rest = [], []
        for _, paths in sorted(by_frames.items()):
            nfull = len(paths) // args.batch_size * args.batch_size
            batches += [paths[i:i + args.batch_size] for i in range(0, nfull, args.batch_size)]
            rest += paths[nfull:]
        rest.sort(key=lambda p: (lengths[p], p))
        batches += [rest[i:i + args.batch_size] for i in range(0, len(rest), args.batch_size)]
    else:
        batches = [paths[i:i + args.batch_size] for _, paths in sorted(by_frames.items()) for i in range(0, len(paths), args.batch_size)]
    done = 0
    for chunk in batches:
        specs, peaks = [], []
        for path in chunk:
            y, peak = _load_normalised(path, target_sr)
            Y = model._forward_transform(model._stft(y[None].to(device))).unsqueeze(1)      # [1,1,F,frames]
            specs.append(pad_spec(Y, mode=pad_mode))
            peaks.append(peak)
        uniform = len({int(Y.shape[-1]) for Y in specs}) == 1
        Ys = torch.cat(specs) if uniform else [Y[0] for Y in specs]                         # ragged: list of [1,F,T_b]
        sample, _ = build_sampler(model, Ys, args, args.seed, [index[p] for p in chunk])()
        for j, path in enumerate(chunk):
            spec = sample[j:j + 1, 0] if uniform else sample[j]
            x = model.to_audio(spec, lengths[path])[0].cpu().numpy()                        # spec_back + iSTFT (enhancement.py:99)
            name = path.replace(test_dir, "")
            name = name[1:] if name.startswith("/") else name
            write_audio(join(enhanced_dir, name), x * peaks[j], target_sr)                  # renormalise (enhancement.py:102)
            done += 1
    return done


def load_model(ckpt: str, device, rank: int, world: int) -> ScoreModel:
    """Single process: read the checkpoint.  Several ranks with an initialised process group: rank 0 reads it (and swaps the
    EMA weights in), the others build the same architecture from the broadcast hyper-parameters and receive the weights in
    ONE broadcast over RCCL/xGMI (parallel.broadcast_backbone_weights) -- the only collective of the job."""
    import torch.distributed as dist
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        model = ScoreModel.load_from_checkpoint(ckpt, map_location=device)
        model.eval()
        return model
    from .parallel import broadcast_backbone_weights
    box = [None]
    if rank == 0:
        model = ScoreModel.load_from_checkpoint(ckpt, map_location=device)
        model.eval()                                                 # EMA weights in, before they are broadcast
        box[0] = dict(model.hparams)
    dist.broadcast_object_list(box, src=0)
    if rank != 0:
        model = ScoreModel(**box[0])
        model._error_loading_ema = True                              # nothing to swap: the received weights ARE the EMA weights
        model.eval()
        model.to(device)
    broadcast_backbone_weights(model.dnn, src=0)
    return model


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
    parser.add_argument("--ragged", action="store_true", help="Batch files of different padded lengths together (OUVE models, pc / ode samplers, ald / none correctors)")
    args = parser.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device(args.device)
    if device.type == "cuda":
        if device.index is None:                                     # plain "cuda": one GPU per rank under torchrun
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
    import torch.distributed as dist
    started_pg = False
    if world > 1 and dist.is_available() and not dist.is_initialized() and os.environ.get("MASTER_PORT"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if device.type == "cuda" else "gloo", rank=rank, world_size=world)
        started_pg = True
    model = load_model(args.ckpt, device, rank, world)
    model.t_eps = args.t_eps
    model.to(device)

    files = list_audio(args.test_dir)
    lo, hi = shard_range(len(files), rank, world)
    n = enhance_files(model, files[lo:hi], args.test_dir, args.enhanced_dir, args, device, first_index=lo)
    print(f"[rank {rank}/{world}] enhanced {n} of {len(files)} files into {args.enhanced_dir}")
    if started_pg:
        dist.barrier()
        dist.destroy_process_group()
    return n


if __name__ == "__main__":
    main()
