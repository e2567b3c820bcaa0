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
  cut into ragged batches of ``--batch_size`` when the measured cost model (``plan_batches``) says that beats the small
  bucketed batches;
* under ``torchrun`` every rank takes a contiguous shard of the sorted file list (the split of the reference's validation
  loop, ``model.py:212-223``) or, with ``--balance frames``, a longest-processing-time share by padded frame count; rank 0
  reads the checkpoint and the weights reach the other ranks in one RCCL broadcast; there is no collective on the data path;
* every file is read / resampled once on background threads ahead of the GPU, and the inverse transform + file writes of a
  batch overlap the sampling of the next one;
* audio I/O uses ``soundfile`` when it is installed; otherwise ``scipy.io.wavfile`` for wav and the package's own decoder
  (util/flac.py) for flac input; resampling to the
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
        from .util.flac import is_flac, read_flac
        if is_flac(path):                               # (scipy reads wav only; the reference's glob also yields *.flac)
            xi, sr, bits = read_flac(path)
            return np.ascontiguousarray(xi[:, 0].astype(np.float32) / float(2 ** (bits - 1))), int(sr)
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
    """Same globbing order as the reference (``enhancement.py:39-43``)."""
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


RAGGED_COST = 1.16          # GPU time per frame of a ragged batch relative to a uniform one (profiles/r02_ragged_bench.txt; 1.15 in r03_ragged_bench.txt)


def probe_samples(path: str, target_sr: int) -> int:
    """Number of samples the file has at the model's rate, from its HEADER (no decoding, no resampling): what the batching and the
    rank assignment are planned from.  ``resample_poly`` yields ceil(n * up / down) samples."""
    try:
        import soundfile
        info = soundfile.info(path)
        n, sr = int(info.frames), int(info.samplerate)
    except ImportError:
        from .util.flac import flac_info, is_flac
        from scipy.io import wavfile
        if is_flac(path):
            n, sr, _, _ = flac_info(path)
            return n if sr == target_sr else -(-n * (target_sr // gcd(sr, target_sr)) // (sr // gcd(sr, target_sr)))
        try:
            sr, x = wavfile.read(path, mmap=True)          # maps the data chunk: the header gives the shape
            n = int(x.shape[0])
            del x
        except ValueError:                                  # a sample format scipy cannot map
            sr, x = wavfile.read(path)
            n = int(x.shape[0])
    if sr == target_sr:
        return n
    g = gcd(sr, target_sr)
    up, down = target_sr // g, sr // g
    return -(-n * up // down)


def padded_frames(n_samples: int, hop: int) -> int:
    """Frames of the padded spectrogram of an n-sample waveform: centred STFT (n // hop + 1 frames, data_module.py:212-214), then
    ``pad_spec`` to the next multiple of 64 (util/other.py:76-90)."""
    return (n_samples // hop + 1 + 63) // 64 * 64


def assign_files(frames: List[int], rank: int, world: int, balance: str) -> List[int]:
    """Indices (into the sorted file list) of the files of ``rank``.

    'contiguous': the reference's split (model.py:212-223): equal contiguous shards, remainder to the last rank.
    'frames':     longest-processing-time assignment by padded frame count -- files by decreasing length (ties by index), each to
                  the rank with the fewest frames so far (ties to the lowest rank).  Every rank computes the same table from the
                  headers alone.  An utterance's noise is a function of (seed, its index in the file list), never of its rank or
                  batch, so the enhanced files are the same under either assignment."""
    if balance == "contiguous" or world == 1:
        lo, hi = shard_range(len(frames), rank, world)
        return list(range(lo, hi))
    if balance != "frames":
        raise ValueError(f"--balance {balance} not supported")
    load = [0] * world
    mine: List[int] = []
    for i in sorted(range(len(frames)), key=lambda k: (-frames[k], k)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += frames[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def plan_batches(idx: List[int], frames: List[int], batch_size: int, ragged: bool, ragged_factor: float = RAGGED_COST) -> List[List[int]]:
    """Batches of file indices.  Uniform batches hold files of ONE padded frame count.  With ``ragged`` the leftovers of every
    length (what does not fill a uniform batch) are pooled, shortest first, and cut into ragged batches -- unless the cost model
    says the bucketed leftovers are cheaper: a ragged batch costs ``ragged_factor`` x its frames (measured,
    profiles/r03_ragged_bench.txt), a uniform batch of n < batch_size utterances its frames x the small-batch penalty
    ``batch_cost(n)`` (measured batch-1 / batch-8 / batch-32 rates)."""
    by_frames: Dict[int, List[int]] = {}
    for i in idx:
        by_frames.setdefault(frames[i], []).append(i)
    if not ragged:
        return [g[k:k + batch_size] for _, g in sorted(by_frames.items()) for k in range(0, len(g), batch_size)]
    batches, rest = [], []
    for _, g in sorted(by_frames.items()):
        nfull = len(g) // batch_size * batch_size
        batches += [g[k:k + batch_size] for k in range(0, nfull, batch_size)]
        if nfull < len(g):
            rest.append(g[nfull:])
    pooled = sorted((i for g in rest for i in g), key=lambda i: (frames[i], i))
    ragged_batches = [pooled[k:k + batch_size] for k in range(0, len(pooled), batch_size)]
    cost_bucketed = sum(batch_cost(len(g)) * sum(frames[i] for i in g) for g in rest)
    cost_ragged = sum((ragged_factor if len({frames[i] for i in g}) > 1 else 1.0) * batch_cost(len(g)) * sum(frames[i] for i in g)
                      for g in ragged_batches)
    return batches + (ragged_batches if cost_ragged < cost_bucketed else rest)


def batch_cost(n: int) -> float:
    """GPU time per utterance-frame of a uniform batch of n utterances relative to batch 32, interpolated in log2(n) between the
    measured rates 2.2 (n=1), 4.6 (n=8) and 5.1 utt/s (n=32)."""
    import math
    pts = [(0.0, 5.1 / 2.2), (3.0, 5.1 / 4.6), (5.0, 1.0)]
    x = min(max(math.log2(max(n, 1)), 0.0), 5.0)
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if x <= x1:
            return y0 + (y1 - y0) * (x - x0) / (x1 - x0)
    return 1.0


class _Writer:
    """Background thread that finishes a batch while the next one samples: waits for the device-to-host copy of the enhanced
    waveforms (an event on the stream they were produced on), undoes the peak normalisation and writes the files."""

    def __init__(self):
        import queue
        import threading
        self.q = queue.Queue(maxsize=4)
        self.err = None
        self.done = 0
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            if self.err is not None:
                continue                # after the first failure: drain the queue without writing (put() raises on the main thread)
            try:
                event, items = job
                if event is not None:
                    event.synchronize()
                for host, peak, out_path, sr in items:
                    write_audio(out_path, host.numpy() * peak, sr)                 # renormalise (enhancement.py:102)
                    self.done += 1
            except Exception as e:      # noqa: BLE001 -- re-raised on the main thread by put() / close()
                self.err = e

    def put(self, event, items):
        if self.err is not None:
            raise self.err
        self.q.put((event, items))

    def close(self, reraise: bool = True) -> int:
        self.q.put(None)
        self.th.join()
        if reraise and self.err is not None:
            raise self.err
        return self.done


def enhance_files(model: ScoreModel, files: List[str], test_dir: str, enhanced_dir: str, args, device, first_index: int = 0,
                  indices=None, frames=None) -> int:
    """Enhance ``files`` in batches.  What has to agree inside a (uniform) batch is the PADDED spectrogram shape, not the waveform
    length: every utterance goes through its own STFT and its own ``pad_spec`` (zero / reflection padding to the next multiple
    of 64 frames, exactly what the reference does per file), utterances with the same padded frame count are stacked, sampled
    together and inverted one by one with their own lengths.  A corpus of mixed lengths therefore runs in batches (64 frames =
    0.5 s at 16 kHz granularity) while every utterance keeps the arithmetic of its single-file run bit for bit.  With ``--seed``
    the noise of an utterance is a function of (seed, its index in the sorted file list): the enhanced files do not depend on
    the batch size, the bucket an utterance landed in or the number of ranks.

    Plumbing (none of it touches an utterance's arithmetic): the batches are planned from the file HEADERS; every file is read,
    resampled and peak-normalised once, on a pool of background threads that runs ahead of the GPU; waveforms of one length share
    one STFT launch; the sampler call does not synchronise the host, so the inverse transform, the device-to-host copy and the
    file writes of batch k (a writer thread waiting on an event) overlap the sampling of batch k + 1.

    ``indices``: the global index (noise stream id) of every file; default ``first_index + position``.  ``frames``: their padded
    frame counts if the caller already probed the headers."""
    from concurrent.futures import ThreadPoolExecutor
    target_sr, pad_mode = model_audio_settings(model)
    hop = model.data_module.hop_length
    n = len(files)
    gidx = list(indices) if indices is not None else [first_index + k for k in range(n)]
    if frames is None:
        frames = [padded_frames(probe_samples(p, target_sr), hop) for p in files]
    ragged = bool(getattr(args, "ragged", False))
    if ragged and (model.sde.__class__.__name__ != "OUVESDE" or args.corrector == "langevin"):
        import warnings
        warnings.warn("--ragged needs an OUVE model and the 'ald' / 'none' corrector (the Langevin corrector couples the utterances of "
                      "a batch; the Schroedinger-bridge sampler is not built for it): batching by padded length instead")
        ragged = False
    batches = plan_batches(list(range(n)), frames, args.batch_size, ragged)
    on_gpu = device.type == "cuda"
    pool = ThreadPoolExecutor(max_workers=int(getattr(args, "io_threads", 4) or 4))
    ahead = 2                                                  # batches whose files are loading while one samples
    loads: Dict[int, object] = {}

    def submit(bi):
        if bi < len(batches):
            for k in batches[bi]:
                loads[k] = pool.submit(_load_normalised, files[k], target_sr)

    for bi in range(min(ahead, len(batches))):
        submit(bi)
    writer = _Writer()
    try:
        for bi, chunk in enumerate(batches):
            submit(bi + ahead)
            waves = {k: loads.pop(k).result() for k in chunk}          # (waveform, peak)
            # one STFT launch per distinct waveform length of the batch (a fixed-length corpus: one launch)
            specs: Dict[int, torch.Tensor] = {}
            by_len: Dict[int, List[int]] = {}
            for k in chunk:
                by_len.setdefault(int(waves[k][0].numel()), []).append(k)
            for L, ks in by_len.items():
                y = torch.stack([waves[k][0] for k in ks])
                if on_gpu:
                    y = y.pin_memory().to(device, non_blocking=True)
                Y = pad_spec(model._forward_transform(model._stft(y)).unsqueeze(1), mode=pad_mode)      # [n,1,F,T]
                for j, k in enumerate(ks):
                    specs[k] = Y[j:j + 1]
            # the header-based plan holds unless a header lied: group by the ACTUAL padded shape and run every group on its own
            # (a uniform batch that turns out mixed would otherwise fall into the ragged path of a sampler that may not take it)
            groups: Dict[int, List[int]] = {}
            for k in chunk:
                groups.setdefault(int(specs[k].shape[-1]), []).append(k)
            runs = [chunk] if (len(groups) == 1 or ragged) else list(groups.values())
            for run in runs:
                uniform = len({int(specs[k].shape[-1]) for k in run}) == 1
                Ys = torch.cat([specs[k] for k in run]) if uniform else [specs[k][0] for k in run]      # ragged: list of [1,F,T_b]
                sample, _ = build_sampler(model, Ys, args, args.seed, [gidx[k] for k in run])()
                items = []
                for j, k in enumerate(run):
                    spec = sample[j:j + 1, 0] if uniform else sample[j]
                    x = model.to_audio(spec, int(waves[k][0].numel()))[0]                               # spec_back + iSTFT (enhancement.py:99)
                    if on_gpu:
                        host = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
                        host.copy_(x, non_blocking=True)
                    else:
                        host = x.cpu()
                    name = files[k].replace(test_dir, "")
                    name = name[1:] if name.startswith("/") else name
                    items.append((host, waves[k][1], join(enhanced_dir, name), target_sr))
                event = None
                if on_gpu:
                    event = torch.cuda.Event()
                    event.record(torch.cuda.current_stream(device))
                writer.put(event, items)
    except BaseException:
        # an exception of the main loop is the one to report: the writer's own stored error (if any) must not replace it
        pool.shutdown(wait=False, cancel_futures=True)
        writer.close(reraise=False)
        raise
    pool.shutdown(wait=False, cancel_futures=True)
    return writer.close()


def load_model(ckpt: str, device, rank: int, world: int) -> ScoreModel:
    """Single process: read the checkpoint.  Several ranks with an initialised process group: rank 0 reads it (and swaps the
    EMA weights in), the others build the same architecture from the broadcast hyper-parameters -- EVERY constructor argument,
    the score wrapper's (c_in / c_out / c_skip / network_scaling / sigma_data / loss_type) included -- and receive the weights
    in ONE broadcast over RCCL/xGMI (parallel.broadcast_backbone_weights) -- the only collective of the job."""
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
    parser.add_argument("--ragged", action="store_true", help="Batch files of different padded lengths together where the cost model says it pays "
                                                              "(OUVE models, pc / ode samplers, ald / none correctors)")
    parser.add_argument("--balance", type=str, choices=("contiguous", "frames"), default="contiguous",
                        help="Files per rank under torchrun: the reference's contiguous split of the sorted list (default), or balanced by "
                             "padded frame count (longest-processing-time); the enhanced files are identical either way")
    parser.add_argument("--io_threads", type=int, default=4, help="Background threads that read / resample the next batches' files")
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
    target_sr, _ = model_audio_settings(model)
    hop = model.data_module.hop_length
    frames = [padded_frames(probe_samples(p, target_sr), hop) for p in files]      # headers only; every rank sees the same table
    mine = assign_files(frames, rank, world, args.balance)
    n = enhance_files(model, [files[i] for i in mine], args.test_dir, args.enhanced_dir, args, device, indices=mine,
                      frames=[frames[i] for i in mine])
    print(f"[rank {rank}/{world}] enhanced {n} of {len(files)} files ({sum(frames[i] for i in mine)} of {sum(frames)} padded frames) "
          f"into {args.enhanced_dir}")
    if started_pg:
        dist.barrier()
        dist.destroy_process_group()
    return n


if __name__ == "__main__":
    main()
