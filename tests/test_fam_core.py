"""
CPU checks of the ROW-FAMILY frame transform shared by the gfx950 kernels of rfx_fam.hip (csrc/rfx_fam_core.h): geometries
with n_fft = 40 h, win_length = 10 h - the reference's default 400 / 100 ms (spectrogram_params.py:24-27, :62-81) at 48, 32,
24, 16 and 8 kHz, plus 44.1 kHz as a cross-check of the specialised engine's factorisation.  The header is compiled for the
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


RATES = [48000, 32000, 24000, 16000, 8000, 44100]


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("rs_pad", [0, 7])
def test_forward_and_inverse_match_numpy(emu, rate, rs_pad):
    n_fft, win = int(0.4 * rate), int(0.1 * rate)
    h = n_fft // 40
    rng = np.random.default_rng(rate)
    u = rng.standard_normal(win).astype(np.float32)
    out = np.zeros(2 * (n_fft // 2 + 1), np.float32)
    assert emu.emu_fam_transform(n_fft, 0, rs_pad, u.ctypes.data_as(FP), out.ctypes.data_as(FP)) == 0
    frame = np.zeros(n_fft)
    frame[15 * h:25 * h] = u
    ref = np.fft.rfft(frame)
    err = np.abs(out.view(np.complex64) - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
    # inverse of an arbitrary one-sided spectrum, the window's quarter of the frame; the imaginary parts of DC / Nyquist are
    # ignored like numpy's / torch's irfft
    X = (rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)).astype(np.complex64)
    back = np.zeros(win, np.float32)
    assert emu.emu_fam_transform(n_fft, 1, rs_pad, X.view(np.float32).ctypes.data_as(FP), back.ctypes.data_as(FP)) == 0
    want = np.fft.irfft(X.astype(np.complex128), n_fft)[15 * h:25 * h]
    assert np.abs(back - want).max() / np.abs(want).max() < 2e-6


def test_geometries_outside_the_family_are_refused(emu):
    out = (ctypes.c_int * 6)()
    assert emu.emu_fam_geom(19200, 4800, 480, out) == 0 and list(out)[:4] == [480, 24, 20, 512]
    assert emu.emu_fam_geom(19200, 4800, 123, out) == 0          # the hop is free (frames are overlap-added afterwards)
    assert emu.emu_fam_geom(8820, 2205, 220, out) == -1          # 22.05 kHz: win is not a multiple of ten
    assert emu.emu_fam_geom(19200, 4000, 480, out) == -1         # window not a quarter of the frame
    assert emu.emu_fam_geom(38400, 9600, 960, out) == -1         # 96 kHz: the 21 x 960 cube exceeds the LDS of a CU
    assert emu.emu_fam_geom(3465, 866, 86, out) == -1
