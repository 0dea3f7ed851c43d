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


@pytest.mark.parametrize("n_fft", LENGTHS)
def test_fused_frame_update_matches_numpy(emu, n_fft):
    """One frame of the fused Griffin-Lim kernel: rfft -> S * X / (|X| + 1e-16) -> irfft, with the projection done pairwise IN
    PLACE on the packed, digit-reversed spectrum between the forward and the inverse passes (gen_pair_project)."""
    rng = np.random.default_rng(n_fft + 2)
    x = rng.standard_normal(n_fft).astype(np.float32)
    S = (np.abs(rng.standard_normal(n_fft // 2 + 1)) * 100).astype(np.float32)
    out = np.zeros(n_fft, np.float32)
    assert emu.emu_gen_gl_frame(n_fft, x.ctypes.data_as(FP), S.ctypes.data_as(FP), out.ctypes.data_as(FP), 64) == 0
    X = np.fft.rfft(x.astype(np.float64))
    want = np.fft.irfft(S.astype(np.float64) * X / (np.abs(X) + 1e-16), n_fft)
    assert np.abs(out - want).max() / np.abs(want).max() < 5e-6
    out2 = np.zeros_like(out)
    emu.emu_gen_gl_frame(n_fft, x.ctypes.data_as(FP), S.ctypes.data_as(FP), out2.ctypes.data_as(FP), 13)
    assert np.array_equal(out, out2)  # the partition of pairs over threads does not matter
    # LDS padding (element i at i + (i >> ps), picked per geometry against bank conflicts) moves data, not arithmetic
    for ps in (4, 6):
        out3 = np.zeros_like(out)
        assert emu.emu_gen_gl_frame_padded(n_fft, x.ctypes.data_as(FP), S.ctypes.data_as(FP), out3.ctypes.data_as(FP), 64, ps, 0) == 0
        assert np.array_equal(out, out3)
    # the kernels' twiddles: exact per-pass tables (rounded once from double) instead of products of two table entries
    out4 = np.zeros_like(out)
    assert emu.emu_gen_gl_frame_padded(n_fft, x.ctypes.data_as(FP), S.ctypes.data_as(FP), out4.ctypes.data_as(FP), 64, 6, 1) == 0
    err_exact = np.abs(out4 - want).max() / np.abs(want).max()
    assert err_exact < 5e-6 and err_exact <= 1.25 * np.abs(out - want).max() / np.abs(want).max() + 1e-7


def test_lds_padding_choice(emu):
    """48 kHz: after the forward passes consecutive bins sit 600 elements apart (48 mod 64 banks: four banks for a wave);
    with room for it the picker takes the shift that makes the stride 609 elements (2 mod 64 banks), without room none."""
    assert emu.emu_gen_pick_pad(19200, 9600 + 200) == 6
    assert emu.emu_gen_pick_pad(19200, 9600) == 0
    assert emu.emu_gen_pick_pad(8820, 4410 + 400) in (0, 7)  # 22.05 kHz: the unpadded stride (294 elements) already spreads over 16 banks


def test_factorisation_and_unsupported_lengths(emu):
    radix = np.zeros(16, np.int32)
    n = emu.emu_gen_factor(19200, radix.ctypes.data_as(IP))
    assert n > 0 and int(np.prod(radix[:n])) == 9600 and set(radix[:n]) <= {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}
    assert n <= 4  # composite digits: 48 kHz takes four passes ([16, 15, 10, 4]), not the seven of radices 4 / 2 / 3 / 5
    for n_fft, most in ((n_fft_of(22050), 4), (n_fft_of(32000), 4), (n_fft_of(16000), 4), (n_fft_of(8000), 3), (3465, 4)):
        k = emu.emu_gen_factor(n_fft, radix.ctypes.data_as(IP))
        assert 0 < k <= most and int(np.prod(radix[:k])) == (n_fft // 2 if n_fft % 2 == 0 else n_fft)
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
