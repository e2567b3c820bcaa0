#!/usr/bin/env python
"""How much of the full-batch rate does batching by PADDED FRAME COUNT (sgmse_amd/enhancement.py: utterances whose
pad_spec'ed spectrograms have the same number of frames run together) reach on a corpus of mixed lengths?

No corpus is available offline, so the length distribution is a parameter: by default 824 files (the size of the
VoiceBank-DEMAND test set the reference's README evaluates on) with log-normal durations, median 2.6 s, clipped to
[1, 10] s.  Cost model: seconds per batch of b utterances of 512 frames, linear between the measured points of
profiles/r02_other_workloads.txt (batch 1 / 8 / 32), scaled by frames / 512 (every kernel's work is linear in the frame
count at fixed F).  Prints the corpus time under (a) the reference's one-file loop, (b) the bucketed batches, (c) an ideal
ragged batcher that always runs full batches of 32."""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=824)
    ap.add_argument("--median", type=float, default=2.6)
    ap.add_argument("--sigma", type=float, default=0.45)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    # measured seconds per step at 512 frames: batch 1, 8, 32 (profiles/r02_other_workloads.txt, r02_bench_b32.json)
    ap.add_argument("--t1", type=float, default=0.461)
    ap.add_argument("--t8", type=float, default=1.765)
    ap.add_argument("--t32", type=float, default=6.42)
    ap.add_argument("--ragged-factor", type=float, default=1.16, help="time per frame of a ragged batch relative to a uniform one")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dur = np.clip(rng.lognormal(np.log(a.median), a.sigma, a.files), 1.0, 10.0)
    frames = (dur * 16000).astype(int) // 128 + 1
    padded = (frames + 63) // 64 * 64

    def step_seconds(b, t_pad):
        return float(np.interp(b, [1, 8, 32], [a.t1, a.t8, a.t32])) * t_pad / 512.0

    one_by_one = sum(step_seconds(1, t) for t in padded)
    bucketed, nb, partial = 0.0, 0, 0
    for t in np.unique(padded):
        n = int((padded == t).sum())
        full, rest = divmod(n, a.batch)
        bucketed += full * step_seconds(a.batch, t) + (step_seconds(rest, t) if rest else 0.0)
        nb += full + (1 if rest else 0)
        partial += rest
    ideal = step_seconds(a.batch, 512) / a.batch * float(padded.sum()) / 512.0
    print(f"{a.files} files, durations {dur.min():.1f}-{dur.max():.1f} s (median {np.median(dur):.2f}), "
          f"{len(np.unique(padded))} buckets of padded length, {nb} batches, {partial} files in partial batches")
    print(f"one file at a time (reference loop): {one_by_one:8.1f} s  = {a.files / one_by_one:.2f} utt/s")
    print(f"bucketed by padded frame count:      {bucketed:8.1f} s  = {a.files / bucketed:.2f} utt/s "
          f"({100 * ideal / bucketed:.1f} % of an ideal ragged batcher)")
    def ragged_seconds(frames_sorted):
        tot = 0.0
        for i in range(0, len(frames_sorted), a.batch):
            chunk = frames_sorted[i:i + a.batch]
            tot += a.ragged_factor * step_seconds(len(chunk), 512) * float(chunk.sum()) / (512.0 * len(chunk))
        return tot
    print(f"all batches ragged, measured factor:   {ragged_seconds(np.sort(padded)):8.1f} s  = {a.files / ragged_seconds(np.sort(padded)):.2f} utt/s")
    hybrid, rest = 0.0, []
    for t in np.unique(padded):
        n = int((padded == t).sum())
        full = n // a.batch
        hybrid += full * step_seconds(a.batch, t)
        rest += [t] * (n - full * a.batch)
    hybrid += ragged_seconds(np.sort(np.array(rest))) if rest else 0.0
    print(f"--ragged (full buckets uniform, rest ragged): {hybrid:8.1f} s  = {a.files / hybrid:.2f} utt/s "
          f"({100 * ideal / hybrid:.1f} % of an ideal ragged batcher)")
    print(f"ideal ragged batches of {a.batch}:           {ideal:8.1f} s  = {a.files / ideal:.2f} utt/s")


if __name__ == "__main__":
    main()
