"""
The hand-derived inverse oracle (oracle/riffusion_oracle.py) against the statement-by-statement transcript of
torchaudio 0.13.0 on real autograd + torch.optim.SGD + in-place tprev.mul_() (oracle/torchaudio_transcript.py).
CPU only.  This removes "the gradient / the momentum update / the aliasing of tprev were derived by hand" from
the unpinned surface of the inverse half; what is left from memory is listed in the transcript's header.
"""
import numpy as np
import pytest
import torch

import riffusion_oracle as O
import torchaudio_transcript as TA


def _mel_input(B, T, seed, scale=3e7):
    g = torch.Generator().manual_seed(seed)
    # image-decoded mel amplitudes: 256 distinct values in [0, max_value] (image_util.py:96-108)
    u8 = torch.randint(0, 256, (B, 512, T), generator=g)
    return (((255 - u8).float() / 255) ** 4 * scale).contiguous()


@pytest.mark.parametrize("B,T,iters", [(1, 24, 200), (2, 23, 60), (1, 64, 40)])
def test_sgd_transcript_equals_hand_derived(B, T, iters):
    p = O.OracleParams(max_mel_iters=iters)
    mel = _mel_input(B, T, seed=B * 100 + T)
    spec0 = torch.rand(B, T, p.n_stft, generator=torch.Generator().manual_seed(5))
    want = TA.inverse_mel_scale(p)(mel, spec0=spec0)
    got, steps = O.inverse_mel_scale_sgd(mel, p, spec0=spec0, return_iters=True)
    assert want.shape == got.shape == (B, p.n_stft, T)
    assert steps == iters  # never stops early at spectrogram scale
    # zero filterbank rows: the initial guess passes through both bit for bit
    fb = O.mel_filterbank(p)
    dead = (fb.abs().sum(1) == 0)
    assert torch.equal(want[:, dead], got[:, dead])
    assert torch.equal(got[:, dead], spec0.transpose(1, 2)[:, dead])
    rel = float(torch.linalg.norm(want - got) / torch.linalg.norm(want))
    worst = float((want - got).abs().max() / want.abs().max())
    print(f"B={B} T={T} iters={iters}: rel-L2 {rel:.2e}, max {worst:.2e}")
    assert rel < 2e-6 and worst < 2e-5


def test_sgd_transcript_early_stop_agrees():
    """tolerance_change fires on tiny inputs: both restatements stop after the same number of steps."""
    p = O.OracleParams(max_mel_iters=200)
    mel = _mel_input(1, 22, seed=3, scale=1e-3)
    spec0 = torch.rand(1, 22, p.n_stft, generator=torch.Generator().manual_seed(6)) * 1e-4
    mod = TA.inverse_mel_scale(p)
    want = mod(mel, spec0=spec0)
    got, steps = O.inverse_mel_scale_sgd(mel, p, spec0=spec0, return_iters=True)
    assert steps == mod.steps_run and steps < 200
    assert float(torch.linalg.norm(want - got) / torch.linalg.norm(want)) < 1e-5


def test_sgd_rng_draw_is_the_same_call():
    """Without injection both draw torch.rand(B, T, F) first thing from the global generator."""
    p = O.OracleParams(max_mel_iters=3)
    mel = _mel_input(1, 22, seed=9)
    torch.manual_seed(77)
    want = TA.inverse_mel_scale(p)(mel)
    torch.manual_seed(77)
    got = O.inverse_mel_scale_sgd(mel, p)
    assert float(torch.linalg.norm(want - got) / torch.linalg.norm(want)) < 1e-6


@pytest.mark.parametrize("n_iter", [0, 1, 2, 8])
def test_griffinlim_transcript_equals_oracle(n_iter):
    """Same ATen ops in the same order: the out-of-place momentum term of the oracle is bit-identical to the
    in-place tprev.mul_() of torchaudio."""
    p = O.OracleParams()
    g = torch.Generator().manual_seed(11)
    mag = torch.rand(2, p.n_stft, 30, generator=g) * 1000
    a0 = torch.rand(2, p.n_stft, 30, dtype=torch.complex64, generator=g)
    want = TA.griffinlim(mag, p, angles0=a0, n_iter=n_iter)
    got = O.griffinlim(mag, p, angles0=a0, n_iter=n_iter)
    assert want.shape == got.shape == (2, 441 * 29)
    assert torch.equal(want, got)


def test_griffinlim_rng_draw_is_the_same_call():
    p = O.OracleParams()
    mag = torch.rand(1, p.n_stft, 24, generator=torch.Generator().manual_seed(2)) * 10
    torch.manual_seed(5)
    want = TA.griffinlim(mag, p, n_iter=2)
    torch.manual_seed(5)
    got = O.griffinlim(mag, p, n_iter=2)
    assert torch.equal(want, got)


def test_pcm16_quotient_follows_numpy_1_19():
    """audio_util.py:23-24 under the reference's pinned numpy 1.19.4 (cog.yaml:24): `32767 / np.float32` is a float64
    scalar, and `float32_array *= float64_scalar` multiplies by float32(that quotient).  numpy >= 2 would divide in
    float32 instead; the oracle states the pinned semantics explicitly, whatever numpy runs the tests."""
    rng = np.random.default_rng(1)
    hits = 0
    for i in range(400):
        x = (rng.standard_normal((1, 64)) * 10 ** rng.uniform(-3, 4)).astype(np.float32)
        peak = np.max(np.abs(x))
        q64 = np.float32(np.float64(32767.0) / np.float64(peak))
        q32 = np.float32(32767.0) / peak
        hits += int(q64 != q32)
        want = (x * q64).transpose(1, 0).astype(np.int16)
        assert np.array_equal(O.pcm16_from_waveform(x, normalize=True), want)
    print(f"{hits} of 400 peaks where the numpy-2 float32 quotient would differ (double rounding)")
