"""
CPU checks of the per-thread arithmetic shared by the gfx950 kernels (csrc/rfx_core.h): the header is
compiled for the host together with tests/emu/rfx_emu.cpp, which loops the 441 logical threads phase
by phase.  Every butterfly, twiddle, index map and the slot <-> bin correspondence is compared with
numpy's FFT here, before any GPU time is spent.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = ctypes.POINTER(ctypes.c_float)
IP = ctypes.POINTER(ctypes.c_int)
N, W, H = 17640, 4410, 441


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "librfx_emu.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emu", "rfx_emu.cpp")], check=True)
    return ctypes.CDLL(so)


@pytest.fixture(scope="module")
def maps(emu):
    bin_ = np.zeros(9261, np.int32); cj = np.zeros(9261, np.int32); pc = np.zeros(9261, np.int32); pf = np.zeros(9261, np.int32)
    emu.emu_slot_maps(bin_.ctypes.data_as(IP), cj.ctypes.data_as(IP), pc.ctypes.data_as(IP), pf.ctypes.data_as(IP))
    return bin_, cj, pc, pf


def test_dft21_both_directions(emu):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(21) + 1j * rng.standard_normal(21)).astype(np.complex64)
    for inv in (0, 1):
        b = x.copy().view(np.float32)
        emu.emu_dft21(b.ctypes.data_as(FP), inv)
        ref = np.fft.ifft(x) * 21 if inv else np.fft.fft(x)
        assert np.abs(b.view(np.complex64) - ref).max() < 5e-6


def test_slot_maps_cover_every_bin(maps):
    bin_, cj, pc, pf = maps
    assert sorted(np.unique(bin_)) == list(range(8821))
    counts = np.bincount(bin_, minlength=8821)
    assert counts.max() == 2 and (counts == 2).sum() == 440
    assert np.all(np.isin(np.where(counts == 2)[0] % 40, (0, 20)))
    for pos in (pc, pf):  # positions are unique, inside the padded frame and never on a padding lane
        assert len(np.unique(pos)) == 9261 and pos.max() < 9408
    # padding lanes: owner index 63 of each 64
    assert not np.any(((pf[(np.arange(9261) // 21 // 21 * 0 + np.arange(9261)) ] % 4) < 0))


def test_forward_frame_matches_numpy_fft(emu, maps):
    bin_, cj, _, _ = maps
    rng = np.random.default_rng(2)
    seg = rng.standard_normal(W).astype(np.float32)
    out = np.zeros(9261 * 2, np.float32)
    emu.emu_forward(seg.ctypes.data_as(FP), out.ctypes.data_as(FP))
    x = np.zeros(N); x[6615:6615 + W] = seg
    X = np.fft.fft(x)
    ref = np.where(cj == 1, np.conj(X[bin_]), X[bin_])
    assert np.abs(out.view(np.complex64) - ref).max() / np.abs(X).max() < 5e-7


def test_inverse_frame_matches_numpy_irfft(emu, maps):
    bin_, cj, _, _ = maps
    rng = np.random.default_rng(3)
    Xh = (rng.standard_normal(8821) + 1j * rng.standard_normal(8821))
    full = np.fft.irfft(Xh, n=N)  # ignores Im of DC and Nyquist exactly like torch.istft's irfft
    Z = np.where(cj == 1, np.conj(Xh[bin_]), Xh[bin_]).astype(np.complex64)
    y = np.zeros(W, np.float32)
    emu.emu_inverse(Z.view(np.float32).ctypes.data_as(FP), y.ctypes.data_as(FP))
    got = y.astype(np.float64) * 2.0 / N
    assert np.abs(got - full[6615:6615 + W]).max() / np.abs(full).max() < 5e-6


def test_round_trip_identity(emu, maps):
    rng = np.random.default_rng(4)
    seg = (rng.standard_normal(W) * 1e4).astype(np.float32)
    out = np.zeros(9261 * 2, np.float32)
    emu.emu_forward(seg.ctypes.data_as(FP), out.ctypes.data_as(FP))
    y = np.zeros(W, np.float32)
    emu.emu_inverse(out.ctypes.data_as(FP), y.ctypes.data_as(FP))
    assert np.abs(y * np.float32(2.0 / N) - seg).max() / np.abs(seg).max() < 2e-6
