"""
CPU checks of the GENERIC frame transform shared by the gfx950 kernels of rfx_generic.hip (csrc/rfx_gen_core.h): the header
is compiled for the host together with tests/emu/rfx_gen_emu.cpp, which loops the logical threads pass by pass.  Every
geometry the reference can be asked for (sample rate x the default 400 / 100 / 10 ms, spectrogram_params.py:62-81) is
compared with numpy's real FFT before any GPU time is spent.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = ctypes.POINTER(ctypes.c_float)
IP = ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("genemu") / "librfx_gen_emu.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emu", "rfx_gen_emu.cpp")], check=True)
    lib = ctypes.CDLL(so)
    lib.emu_gen_gl_update.argtypes = [FP, FP, ctypes.c_float, ctypes.c_float, FP]
    return lib


def n_fft_of(sample_rate, padded_ms=400):
    return int(padded_ms / 1000.0 * sample_rate)


# 48 kHz, 32 kHz, 24 kHz, 22.05 kHz, 16 kHz, 11.025 kHz, 8 kHz at the default 400 ms; 44.1 kHz at 300 ms; small odd lengths
LENGTHS = [n_fft_of(48000), n_fft_of(32000), n_fft_of(24000), n_fft_of(22050), n_fft_of(16000), n_fft_of(11025), n_fft_of(8000),
           n_fft_of(44100, 300), 17640, 3465, 1001, 143, 64, 6]


@pytest.mark.parametrize("n_fft", LENGTHS)
def test_real_fft_and_inverse_match_numpy(emu, n_fft):
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(n_fft).astype(np.float32)
    out = np.zeros(2 * (n_fft // 2 + 1), np.float32)
    assert emu.emu_gen_rfft(n_fft, x.ctypes.data_as(FP), out.ctypes.data_as(FP), 96) == 0
    ref = np.fft.rfft(x.astype(np.float64))
    err = np.abs(out.view(np.complex64) - ref).max() / np.abs(ref).max()
    assert err < 3e-6, err
    # inverse of an arbitrary one-sided spectrum; imaginary parts of DC / Nyquist are ignored like numpy's / torch's irfft
    X = (rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)).astype(np.complex64)
    back = np.zeros(n_fft, np.float32)
    assert emu.emu_gen_irfft(n_fft, X.view(np.float32).ctypes.data_as(FP), back.ctypes.data_as(FP), 64) == 0
    want = np.fft.irfft(X.astype(np.complex128), n_fft)
    assert np.abs(back - want).max() / np.abs(want).max() < 3e-6
    # thread count must not matter (each pass partitions its butterflies over the threads)
    out2 = np.zeros_like(out)
    emu.emu_gen_rfft(n_fft, x.ctypes.data_as(FP), out2.ctypes.data_as(FP), 7)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("n_fft", LENGTHS)
def test_inplace_passes_match_numpy(emu, n_fft):
    """The in-place variant (forward decimation in frequency -> digit-reversed spectrum, inverse decimation in time)."""
    rng = np.random.default_rng(n_fft + 1)
    x = rng.standard_normal(n_fft).astype(np.float32)
    out = np.zeros(2 * (n_fft // 2 + 1), np.float32)
    assert emu.emu_gen_rfft_inplace(n_fft, x.ctypes.data_as(FP), out.ctypes.data_as(FP), 96) == 0
    ref = np.fft.rfft(x.astype(np.float64))
    assert np.abs(out.view(np.complex64) - ref).max() / np.abs(ref).max() < 3e-6
    X = (rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)).astype(np.complex64)
    back = np.zeros(n_fft, np.float32)
    assert emu.emu_gen_irfft_inplace(n_fft, X.view(np.float32).ctypes.data_as(FP), back.ctypes.data_as(FP), 50) == 0
    want = np.fft.irfft(X.astype(np.complex128), n_fft)
    assert np.abs(back - want).max() / np.abs(want).max() < 3e-6


def test_factorisation_and_unsupported_lengths(emu):
    radix = np.zeros(16, np.int32)
    n = emu.emu_gen_factor(19200, radix.ctypes.data_as(IP))
    assert n > 0 and int(np.prod(radix[:n])) == 9600 and set(radix[:n]) <= {2, 3, 4, 5, 7, 11, 13}
    assert emu.emu_gen_factor(2 * 17, radix.ctypes.data_as(IP)) == 0  # prime factor 17: rejected, not mis-computed
    assert emu.emu_gen_factor(2 * 10007, radix.ctypes.data_as(IP)) == 0


def test_gl_update_matches_reference_ops(emu):
    """angles = rebuilt - tprev * m; angles / (|angles| + 1e-16); S * angles (torchaudio functional.griffinlim)."""
    import torch

    rng = np.random.default_rng(0)
    for scale in (1e-12, 1.0, 3e7):
        for _ in range(50):
            r = (rng.standard_normal(2) * scale).astype(np.float32)
            t = (rng.standard_normal(2) * scale).astype(np.float32)
            S, m = np.float32(abs(rng.standard_normal()) * scale), np.float32(0.99 / 1.99)
            out = np.zeros(2, np.float32)
            emu.emu_gen_gl_update(r.ctypes.data_as(FP), t.ctypes.data_as(FP), m, S, out.ctypes.data_as(FP))
            a = torch.complex(torch.tensor(r[0]), torch.tensor(r[1])) - torch.complex(torch.tensor(t[0]), torch.tensor(t[1])) * float(m)
            want = (a / (a.abs() + 1e-16)) * float(S)
            got = complex(out[0], out[1])
            assert abs(got - complex(want)) <= 4e-7 * max(abs(complex(want)), 1e-30)
    z = np.zeros(2, np.float32)
    out = np.ones(2, np.float32)
    emu.emu_gen_gl_update(z.ctypes.data_as(FP), z.ctypes.data_as(FP), np.float32(0.5), np.float32(3.0), out.ctypes.data_as(FP))
    assert out[0] == 0.0 and out[1] == 0.0  # 0 / 1e-16 = 0, as in the reference
