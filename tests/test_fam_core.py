"""
CPU checks of the ROW-FAMILY frame transform shared by the gfx950 kernels of rfx_fam.hip (csrc/rfx_fam_core.h): geometries
with n_fft = 40 h, win_length = 10 h - the reference's default 400 / 100 ms (spectrogram_params.py:24-27, :62-81) at 48, 32,
24, 16 and 8 kHz, plus 44.1 kHz as a cross-check of the specialised engine's factorisation - and, since round 4, n_fft = 20 h,
win_length = 5 h with h = 441: 22.05 kHz, where the window (2205 samples) is not ten hops and sits at an odd offset (3307).  The header is compiled for the
host with tests/emu/rfx_fam_emu.cpp, which loops the logical threads phase by phase, and compared with numpy's real FFT.
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
    so = str(tmp_path_factory.mktemp("famemu") / "librfx_fam_emu.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emu", "rfx_fam_emu.cpp")], check=True)
    return ctypes.CDLL(so)


RATES = [48000, 32000, 24000, 16000, 8000, 44100, 22050]


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("rs_pad", [0, 7])
def test_forward_and_inverse_match_numpy(emu, rate, rs_pad):
    n_fft, win = int(0.4 * rate), int(0.1 * rate)
    left = (n_fft - win) // 2  # where torch.stft centres the window in the frame: 15 h in the 40 h family, 3307 at 22.05 kHz
    rng = np.random.default_rng(rate)
    u = rng.standard_normal(win).astype(np.float32)
    out = np.zeros(2 * (n_fft // 2 + 1), np.float32)
    assert emu.emu_fam_transform(n_fft, 0, rs_pad, u.ctypes.data_as(FP), out.ctypes.data_as(FP)) == 0
    frame = np.zeros(n_fft)
    frame[left:left + win] = u
    ref = np.fft.rfft(frame)
    err = np.abs(out.view(np.complex64) - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
    # inverse of an arbitrary one-sided spectrum, the window's quarter of the frame; the imaginary parts of DC / Nyquist are
    # ignored like numpy's / torch's irfft
    X = (rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)).astype(np.complex64)
    back = np.zeros(win, np.float32)
    assert emu.emu_fam_transform(n_fft, 1, rs_pad, X.view(np.float32).ctypes.data_as(FP), back.ctypes.data_as(FP)) == 0
    want = np.fft.irfft(X.astype(np.complex128), n_fft)[left:left + win]
    assert np.abs(back - want).max() / np.abs(want).max() < 2e-6


@pytest.mark.parametrize("rate", RATES)
def test_every_bin_has_exactly_one_primary_slot(emu, rate):
    """The forward kernel re-orders a frame through LDS: every one-sided bin must be written by exactly one slot (bins on rows
    0 and 20 exist twice, bins with k mod 40 in 21..39 only as a conjugate slot)."""
    n_fft = int(0.4 * rate)
    count = np.zeros(n_fft // 2 + 1, np.int32)
    assert emu.emu_fam_primary_writers(n_fft, count.ctypes.data_as(IP)) == 0
    assert np.array_equal(count, np.ones_like(count))


def test_geometries_outside_the_family_are_refused(emu):
    out = (ctypes.c_int * 6)()
    assert emu.emu_fam_geom(19200, 4800, 480, out) == 0 and list(out)[:4] == [480, 24, 20, 512]
    assert emu.emu_fam_geom(19200, 4800, 123, out) == 0          # the hop is free (frames are overlap-added afterwards)
    assert emu.emu_fam_geom(8820, 2205, 220, out) == 0 and list(out)[:4] == [441, 21, 21, 448]  # 22.05 kHz: the 20 h family (round 4)
    assert emu.emu_fam_geom(4410, 1102, 110, out) == -1          # 11.025 kHz: neither 40 h nor 20 h
    assert emu.emu_fam_geom(19200, 4000, 480, out) == -1         # window not a quarter of the frame
    assert emu.emu_fam_geom(38400, 9600, 960, out) == -1         # 96 kHz: the 21 x 960 cube exceeds the LDS of a CU
    assert emu.emu_fam_geom(3465, 866, 86, out) == -1


@pytest.mark.parametrize("rate,hop_ms,T,n_iter", [(8000, 10, 31, 0), (8000, 10, 31, 3), (16000, 5, 45, 2), (22050, 10, 33, 0), (22050, 10, 33, 2)])
def test_griffinlim_on_the_emulated_kernels_matches_the_oracle(emu, rate, hop_ms, T, n_iter):
    """The whole loop as rfx_fam.hip + gen_fold_kernel run it (initial synthesis from S * angles0, analysis of
    x_k - m x_{k-1}, per-slot projection incl. the conjugate and duplicate slots, pruned synthesis, overlap-add and envelope
    division) against torchaudio's Griffin-Lim as restated by the oracle - on the CPU, before any GPU time."""
    import sys

    import torch

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import riffusion_oracle as O

    op = O.OracleParams(sample_rate=rate, step_size_ms=hop_ms, max_frequency=rate // 2)
    assert op.n_fft % 20 == 0 and op.win_length * 4 == op.n_fft
    g = torch.Generator().manual_seed(rate + n_iter)
    mag = torch.rand(1, op.n_stft, T, generator=g) * 1000
    a0 = torch.rand(1, op.n_stft, T, dtype=torch.complex64, generator=g)
    want = O.griffinlim(mag, op, angles0=a0, n_iter=n_iter)[0].numpy()
    win = O.hann_window(op).numpy().astype(np.float32)
    out = np.zeros(op.hop_length * (T - 1), np.float32)
    magn = np.ascontiguousarray(mag[0].numpy())
    angn = np.ascontiguousarray(torch.view_as_real(a0[0]).numpy())
    rc = emu.emu_fam_griffinlim(op.n_fft, op.hop_length, T, n_iter, ctypes.c_float(0.99), magn.ctypes.data_as(FP), angn.ctypes.data_as(FP),
                                win.ctypes.data_as(FP), out.ctypes.data_as(FP))
    assert rc == 0 and out.shape == want.shape
    snr = 10 * np.log10(np.sum(want.astype(np.float64) ** 2) / np.sum((want.astype(np.float64) - out) ** 2))
    print(f"{rate} Hz, hop {op.hop_length}, T = {T}: emulated row-family Griffin-Lim n_iter={n_iter}: {snr:.1f} dB vs the oracle")
    assert snr >= (110.0 if n_iter == 0 else 90.0)


def test_streamed_radix_24_pass_equals_the_plain_one_bit_for_bit(emu):
    """48 kHz: the 24-point butterfly of pass A streamed column by column (rfx_fam_core.h, round 5: what the kernels run in the
    forward direction) against the plain gen_dft<24> form - same operations, same order, so the same bits on the host."""
    n_fft = 19200
    rng = np.random.default_rng(5)
    u = (rng.standard_normal(n_fft // 4) * 100).astype(np.float32)
    X = (rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)).astype(np.complex64)
    X[0] = X[0].real
    X[-1] = X[-1].real
    res = {}
    try:
        for fwd, inv in ((0, 0), (1, 1)):
            emu.emu_fam_set_stream(fwd, inv)
            spec = np.zeros(n_fft // 2 + 1, np.complex64)
            back = np.zeros(n_fft // 4, np.float32)
            assert emu.emu_fam_transform(n_fft, 0, 0, u.ctypes.data_as(FP), spec.view(np.float32).ctypes.data_as(FP)) == 0
            assert emu.emu_fam_transform(n_fft, 1, 0, X.view(np.float32).ctypes.data_as(FP), back.ctypes.data_as(FP)) == 0
            res[(fwd, inv)] = (spec.copy(), back.copy())
    finally:
        emu.emu_fam_set_stream(1, 0)
    assert np.array_equal(res[(0, 0)][0].view(np.uint32), res[(1, 1)][0].view(np.uint32))
    assert np.array_equal(res[(0, 0)][1].view(np.uint32), res[(1, 1)][1].view(np.uint32))

