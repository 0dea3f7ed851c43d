"""
Round 4, CPU side: independent pins and host logic that need no GPU.

* the four (mel_scale_type, mel_scale_norm) filterbanks of spectrogram_params.py:34-35 against an implementation written by
  other people (transformers.audio_utils.mel_filter_bank, float64): the oracle's bank and the product's bank were written
  from memory by one author, this is the outside check;
* statistics of the counter RNG behind the random starts (rfx_core.h::rand_unit / rand_unit_pair through the host emulator):
  mean, variance, real-imaginary correlation, lag-1 correlation across bins and across frames, a 2-D chi-square;
* the plan cache is bounded (least recently used out), ChunkSource degrades to plain slices without a GPU, the batch entry
  points default to the gather mode that scales.
"""
import collections
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = ctypes.POINTER(ctypes.c_float)


@pytest.mark.parametrize("rate", [44100, 48000])
@pytest.mark.parametrize("scale", ["htk", "slaney"])
@pytest.mark.parametrize("norm", [None, "slaney"])
def test_filterbanks_match_an_independent_implementation(rate, scale, norm):
    tf_audio = pytest.importorskip("transformers.audio_utils")
    import riffusion_oracle as O
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(sample_rate=rate, mel_scale_type=scale, mel_scale_norm=norm)
    op = O.params_from(p)
    mine = O.mel_filterbank(op).double().numpy()  # (n_stft, n_mels)
    n_stft = p.n_fft // 2 + 1
    theirs = tf_audio.mel_filter_bank(n_stft, p.num_frequencies, float(p.min_frequency), float(p.max_frequency), rate, norm=norm, mel_scale=scale)
    assert theirs.shape == mine.shape == (n_stft, 512)
    assert np.abs(mine - theirs).max() <= 1e-4 * max(1.0, np.abs(theirs).max())
    # same support up to one bin at either edge of every filter (fp32 vs fp64 at the triangle's feet)
    for m in range(512):
        a, b = np.nonzero(mine[:, m] > 0)[0], np.nonzero(theirs[:, m] > 1e-12)[0]
        assert len(a) and len(b) and abs(int(a[0]) - int(b[0])) <= 1 and abs(int(a[-1]) - int(b[-1])) <= 1
    # the product's bank is the oracle's bit for bit (tests/test_host_logic.py checks that for the default; here for all four)
    prod = _hip.mel_filterbank(n_stft, float(p.min_frequency), float(p.max_frequency), p.num_frequencies, rate, norm, scale)
    assert torch.equal(prod, O.mel_filterbank(op))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu4") / "librfx_emu.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emu", "rfx_emu.cpp")], check=True)
    return ctypes.CDLL(so)


def test_counter_rng_statistics(emu):
    """The random starts are U[0, 1) per value (torch.rand in the reference).  4000 bins x 256 frames x 2 seeds per stream."""
    n, frames = 4000, 256
    for seed in (0, 0x9E3779B97F4A7C15):
        u = np.zeros((frames, n), np.float32)
        z = np.zeros((frames, n, 2), np.float32)
        for t in range(frames):
            emu.emu_rand_unit(ctypes.c_ulonglong(seed), ctypes.c_ulonglong(t), n, u[t].ctypes.data_as(FP))
            emu.emu_rand_unit_pair(ctypes.c_ulonglong(seed), ctypes.c_ulonglong(t), n, z[t].ctypes.data_as(FP))
        N = frames * n
        tol = 5.0 / np.sqrt(N)  # five sigma of a correlation / a normalised mean over N samples
        for name, x in (("rand_unit", u), ("pair.re", z[..., 0]), ("pair.im", z[..., 1])):
            x = x.astype(np.float64)
            assert 0.0 <= x.min() and x.max() < 1.0, name
            assert abs(x.mean() - 0.5) <= tol * np.sqrt(1 / 12), name
            assert abs(x.var() - 1 / 12) <= tol * np.sqrt(1 / 180), name  # var of (U - 1/2)^2 is 1/180
            c = x - 0.5
            lag_bin = (c[:, 1:] * c[:, :-1]).mean() * 12
            lag_frame = (c[1:] * c[:-1]).mean() * 12
            assert abs(lag_bin) <= tol and abs(lag_frame) <= tol, (name, lag_bin, lag_frame)
        re, im = z[..., 0].astype(np.float64) - 0.5, z[..., 1].astype(np.float64) - 0.5
        assert abs((re * im).mean() * 12) <= tol  # the imaginary part is derived from the real part's hash: must not show
        # 2-D uniformity of (re, im): 32 x 32 cells, chi-square with 1023 degrees of freedom (mean 1023, sigma 45)
        h, _, _ = np.histogram2d(z[..., 0].ravel(), z[..., 1].ravel(), bins=32, range=[[0, 1], [0, 1]])
        chi2 = ((h - N / 1024) ** 2 / (N / 1024)).sum()
        assert abs(chi2 - 1023) <= 5 * 45.2, chi2
        # different frames and different seeds give different streams
        assert not np.array_equal(u[0], u[1])
    a = np.zeros(16, np.float32); b = np.zeros(16, np.float32)
    emu.emu_rand_unit(ctypes.c_ulonglong(1), ctypes.c_ulonglong(5), 16, a.ctypes.data_as(FP))
    emu.emu_rand_unit(ctypes.c_ulonglong(2), ctypes.c_ulonglong(5), 16, b.ctypes.data_as(FP))
    assert not np.array_equal(a, b)


def test_plan_cache_is_bounded_least_recently_used(monkeypatch):
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    made, closed = [], []

    class FakePlan:
        def __init__(self, params, dev, gl_form, frame_engine, plan_layout, imel_form):
            self.key = (params.max_mel_iters, gl_form, frame_engine, plan_layout)
            made.append(self.key)

        def __del__(self):
            closed.append(self.key)

    monkeypatch.setattr(_hip, "Plan", FakePlan)
    monkeypatch.setattr(_hip, "resolve_device", lambda d: torch.device("cuda", 0))
    monkeypatch.setattr(_hip, "_plans", collections.OrderedDict())
    monkeypatch.setattr(_hip, "PLAN_CACHE_SIZE", 3)
    ps = [SpectrogramParams(max_mel_iters=100 + i) for i in range(5)]
    for p in ps[:3]:
        _hip.get_plan(p, "cuda")
    assert _hip.cached_plans() == 3 and not closed
    assert _hip.get_plan(ps[0], "cuda").key[0] == 100 and len(made) == 3  # a hit: nothing built, and 100 is now the most recent
    _hip.get_plan(ps[3], "cuda")  # evicts 101, the least recently used
    import gc

    gc.collect()
    assert _hip.cached_plans() == 3 and [k[0] for k in closed] == [101]
    _hip.get_plan(ps[4], "cuda")
    gc.collect()
    assert [k[0] for k in closed] == [101, 102]
    assert _hip.get_plan(ps[0], "cuda").key[0] == 100 and len(made) == 5  # still cached
    # options are part of the key
    _hip.get_plan(ps[0], "cuda", plan_layout="generic")
    assert made[-1] == (100, "auto", "auto", "generic")


def test_chunk_source_without_a_gpu_is_plain_slices_and_gather_defaults_to_none():
    import inspect

    from riffusion import batch_shard
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter

    items = torch.arange(10 * 3, dtype=torch.uint8).reshape(10, 3)
    bounds = [(2, 5), (5, 8), (8, 9)]
    src = batch_shard.ChunkSource(items, bounds, torch.device("cpu"))
    assert not src.staged
    for i, (a, b) in enumerate(bounds):
        assert torch.equal(src.get(i), items[a:b])
    assert batch_shard.default_gather(None) == "none" and batch_shard.default_gather(object()) == "none"
    assert inspect.signature(batch_shard.sharded_map).parameters["gather"].default == "none"
    assert inspect.signature(batch_shard.result_rows).parameters["gather"].default == "none"
    assert inspect.signature(SpectrogramImageConverter.audio_from_spectrogram_images).parameters["gather"].default is None
