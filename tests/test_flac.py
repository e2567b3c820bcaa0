"""FLAC input without soundfile / torchaudio (reference enhancement.py:39-43 globs *.flac; :61 loads it): sgmse_amd/util/flac.py against
streams written by the independent encoder in tests/flac_encode.py, one test per construct of the format the decoder has a branch for."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from flac_encode import encode_flac

from sgmse_amd.util.flac import FlacError, flac_info, is_flac, read_flac


def _signal(frames, nch, bits, seed=0, smooth=True):
    g = np.random.default_rng(seed)
    t = np.arange(frames)[:, None]
    amp = 2 ** (bits - 2)
    x = amp * 0.6 * np.sin(2 * np.pi * (0.01 + 0.003 * np.arange(nch)[None, :]) * t + np.arange(nch)[None, :])
    x = x + g.normal(0, amp * (0.002 if smooth else 0.2), size=(frames, nch))
    return np.clip(np.round(x), -(2 ** (bits - 1)), 2 ** (bits - 1) - 1).astype(np.int64)


def _roundtrip(tmp_path, x, sr, bits, **kw):
    p = str(tmp_path / "a.flac")
    with open(p, "wb") as fh:
        fh.write(encode_flac(x, sr, bits, **kw))
    y, sr2, bits2 = read_flac(p)
    assert sr2 == sr and bits2 == bits and y.dtype == np.int32
    assert np.array_equal(y, x), "decoded samples differ"
    return p


@pytest.mark.parametrize("order,po", [(0, 0), (1, 1), (2, 2), (3, 0), (4, 3)])
def test_fixed_predictors_and_rice_partitions(tmp_path, order, po):
    x = _signal(3000, 1, 16, seed=order)
    plans = [{"stereo": "independent", "sub": [dict(kind="fixed", order=order, partition_order=po)]}]
    p = _roundtrip(tmp_path, x, 16000, 16, blocksize=1024, plans=plans)      # the last frame has 952 samples: 16-bit explicit block size
    assert flac_info(p) == (3000, 16000, 1, 16) and is_flac(p)


@pytest.mark.parametrize("mode", ["independent", "left_side", "right_side", "mid_side"])
def test_stereo_decorrelation_modes(tmp_path, mode):
    x = _signal(2500, 2, 16, seed=3)
    x[:, 1] = x[:, 0] // 2 + _signal(2500, 1, 12, seed=4)[:, 0]
    plans = [{"stereo": mode, "sub": [dict(kind="fixed", order=2, partition_order=1), dict(kind="fixed", order=1)]}]
    _roundtrip(tmp_path, x, 44100, 16, blocksize=512, plans=plans)


def test_lpc_subframes_24_bit_with_the_wide_rice_parameter(tmp_path):
    x = _signal(4000, 2, 24, seed=5)
    lpc_a = dict(coefs=[1937, -1011, 95, -3], shift=10, precision=12)
    lpc_b = dict(coefs=[2, -1], shift=0, precision=3)
    plans = [{"stereo": "independent", "sub": [dict(kind="lpc", order=4, lpc=lpc_a, rice2=True, partition_order=2),
                                               dict(kind="lpc", order=2, lpc=lpc_b, partition_order=0)]},
             {"stereo": "mid_side", "sub": [dict(kind="lpc", order=4, lpc=lpc_a, rice2=True), dict(kind="fixed", order=2, rice2=True)]}]
    _roundtrip(tmp_path, x, 48000, 24, blocksize=1152, plans=plans)


def test_constant_verbatim_wasted_bits_and_escaped_partitions(tmp_path):
    frames = 256 * 4 + 100                                     # last frame: 8-bit explicit block size
    x = np.zeros((frames, 2), dtype=np.int64)
    x[:, 0] = 1234                                             # constant channel
    x[:, 1] = _signal(frames, 1, 16, seed=7, smooth=False)[:, 0]
    x[256:512, 1] = (x[256:512, 1] >> 3) << 3                  # three wasted bits in the second frame
    plans = [{"stereo": "independent", "sub": [dict(kind="constant"), dict(kind="verbatim")]},
             {"stereo": "independent", "sub": [dict(kind="constant"), dict(kind="fixed", order=1, wasted=3, partition_order=2)]},
             {"stereo": "independent", "sub": [dict(kind="constant"), dict(kind="fixed", order=2, escape_first=True, partition_order=1)]},
             {"stereo": "independent", "sub": [dict(kind="verbatim"), dict(kind="fixed", order=0, rice2=True, escape_first=True)]}]
    _roundtrip(tmp_path, x, 8000, 16, blocksize=256, plans=plans)


def test_id3_prefix_unknown_length_and_other_sample_sizes(tmp_path):
    for bits in (8, 12, 20):
        x = _signal(700, 1, bits, seed=bits)
        plans = [{"stereo": "independent", "explicit_bits": bits != 12, "sub": [dict(kind="fixed", order=2)]}]
        p = _roundtrip(tmp_path, x, 22050, bits, blocksize=576, plans=plans, id3_prefix=True, announce_total=False, with_md5=bits != 20)
        assert flac_info(p) == (700, 22050, 1, bits)


def test_corruption_is_detected(tmp_path):
    x = _signal(2048, 1, 16, seed=9)
    good = bytearray(encode_flac(x, 16000, 16, blocksize=1024))
    p = str(tmp_path / "bad.flac")
    for where, what in ((len(good) - 40, "frame checksum"), (4 + 4 + 20, "MD5")):
        bad = bytearray(good)
        bad[where] ^= 0x10
        with open(p, "wb") as fh:
            fh.write(bad)
        with pytest.raises(FlacError, match=what):
            read_flac(p)
    with open(p, "wb") as fh:
        fh.write(b"RIFF....WAVE")
    with pytest.raises(FlacError, match="not a FLAC"):
        read_flac(p)
    assert not is_flac(p)


def test_the_enhancement_path_reads_flac_files(tmp_path):
    """read_audio / probe_samples of the directory job and the torchaudio / soundfile stand-ins take .flac like .wav."""
    from sgmse_amd import enhancement
    x = _signal(8000, 2, 16, seed=11)
    p = str(tmp_path / "utt.flac")
    with open(p, "wb") as fh:
        fh.write(encode_flac(x, 16000, 16, blocksize=4096, plans=[{"stereo": "mid_side", "sub": [dict(kind="fixed", order=2)] * 2}]))
    y, sr = enhancement.read_audio(p)
    assert sr == 16000 and y.dtype == np.float32 and np.array_equal(y, (x[:, 0] / 32768.0).astype(np.float32))
    assert enhancement.probe_samples(p, 16000) == 8000 and enhancement.probe_samples(p, 48000) == 24000
    assert p in enhancement.list_audio(str(tmp_path))
    shims = os.path.join(os.path.dirname(enhancement.__file__), "compat", "shims")
    sys.path.insert(0, shims)
    try:
        for name in ("torchaudio", "soundfile"):
            sys.modules.pop(name, None)
        import soundfile
        import torchaudio
        w, sr2 = torchaudio.load(p)
        assert sr2 == 16000 and tuple(w.shape) == (2, 8000) and np.array_equal(w.numpy(), (x.T / 32768.0).astype(np.float32))
        d, sr3 = soundfile.read(p, dtype="float32", always_2d=True)
        assert sr3 == 16000 and np.array_equal(d, (x / 32768.0).astype(np.float32))
    finally:
        sys.path.remove(shims)
        for name in ("torchaudio", "soundfile"):
            sys.modules.pop(name, None)


def test_random_streams_round_trip(tmp_path):
    """Randomly drawn frame plans (subframe kinds, predictor orders, partition orders, both Rice widths, escapes, wasted bits, stereo
    modes, block sizes, sample sizes) round-trip sample for sample; every stream carries its MD5, which the decoder verifies."""
    g = np.random.default_rng(2024)
    for trial in range(24):
        bits = int(g.choice([8, 12, 16, 20, 24]))
        nch = int(g.choice([1, 2]))
        blocksize = int(g.choice([192, 256, 576, 1024, 1152]))
        frames = int(g.integers(blocksize + 1, 3 * blocksize + 50))
        x = _signal(frames, nch, bits, seed=trial, smooth=bool(g.integers(0, 2)))
        plans = []
        for _ in range(int(g.integers(1, 4))):
            mode = str(g.choice(["independent", "left_side", "right_side", "mid_side"])) if nch == 2 else "independent"
            subs = []
            for _c in range(nch):
                kind = str(g.choice(["fixed", "fixed", "lpc", "verbatim"]))
                po = int(g.choice([0, 1, 2]))
                if kind == "fixed":
                    subs.append(dict(kind="fixed", order=int(g.integers(0, 5)), partition_order=po, rice2=bool(g.integers(0, 2)),
                                     escape_first=bool(g.integers(0, 4) == 0)))
                elif kind == "lpc":
                    order = int(g.integers(1, 9))
                    prec = int(g.integers(4, 13))
                    coefs = [int(c) for c in g.integers(-(1 << (prec - 1)), 1 << (prec - 1), size=order)]
                    subs.append(dict(kind="lpc", order=order, lpc=dict(coefs=coefs, shift=int(g.integers(prec - 2, prec + 3)), precision=prec),
                                     partition_order=po, rice2=True))
                else:
                    subs.append(dict(kind="verbatim"))
            plans.append({"stereo": mode, "sub": subs, "explicit_bits": bool(g.integers(0, 2))})
        # partition orders must divide every block (the last, shorter one too) and leave room for the predictor's warm-up
        last = frames % blocksize or blocksize
        for pl in plans:
            for sb in pl["sub"]:
                if "partition_order" in sb:
                    while sb["partition_order"] and (last % (1 << sb["partition_order"]) or blocksize % (1 << sb["partition_order"])
                                                     or (last >> sb["partition_order"]) <= sb.get("order", 0)):
                        sb["partition_order"] -= 1
        _roundtrip(tmp_path, x, int(g.choice([8000, 16000, 44100, 48000])), bits, blocksize=blocksize, plans=plans)
