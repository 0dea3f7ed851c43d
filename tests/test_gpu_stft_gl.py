"""
GPU parity tests of the framed transform and Griffin-Lim, through the C ABI (librfx.so) against the
CPU oracle (oracle/riffusion_oracle.py) on the same seeded inputs.
"""
import numpy as np
import pytest
import torch

from helpers import snr_db, synthetic_wave

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["frames", "runs"])
def plan(request):
    """Griffin-Lim has two device forms (csrc/rfx_gl.hip): small batches (<= 4 frames per resident workgroup slot, i.e. most
    shapes in this file) take gl_frame_kernel + gl_fold_kernel, larger ones the run-based gl_iter_kernel of the headline.
    The form is a plan-creation option (rfx_plan_options.gl_form, include/rfx.h), so every test below runs on both."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    pl = _hip.get_plan(SpectrogramParams(), "cuda", gl_form=request.param)
    assert pl.lib.rfx_griffinlim_form(pl.handle, 1, 48) == _hip.GL_FORMS[request.param]
    return pl


@pytest.fixture(scope="module")
def oparams():
    import riffusion_oracle as O

    return O.OracleParams()


@pytest.mark.parametrize("length", [8821, 441 * 30, 441 * 63 + 17, 250400])
def test_stft_complex_matches_torch_stft(plan, oparams, length):
    import riffusion_oracle as O

    wave = synthetic_wave(2, length, seed=length)
    ref = O.stft_complex(wave, oparams)  # (B, 8821, T) complex64 on CPU
    mag, spec, T = plan.stft(wave.cuda(), want_mag=True, want_spec=True)
    assert T == ref.shape[-1]
    got = plan.unpack_complex(spec, 2, T).cpu()
    scale = ref.abs().max()
    err = (got - ref).abs().max() / scale
    assert err < 2e-6, f"relative max error {err}"
    # rel-L2 as well
    assert torch.linalg.norm(got - ref) / torch.linalg.norm(ref) < 2e-6
    # magnitudes equal |spec| of the same kernel
    mag_from_spec = plan.pack_magnitudes(got.abs().cuda()).cpu()
    assert (mag.cpu() - mag_from_spec).abs().max() / scale < 1e-6  # padding positions are zero in both


def _gl_case(plan, oparams, B, T, n_iter, seed):
    import riffusion_oracle as O

    g = torch.Generator().manual_seed(seed)
    # magnitudes of a real signal, so the iteration has a consistent target
    wave = synthetic_wave(B, 441 * (T - 1), seed=seed)
    mag = O.stft_complex(wave, oparams).abs()
    angles0 = torch.rand(mag.shape, dtype=torch.complex64, generator=g)
    ref = O.griffinlim(mag, oparams, angles0=angles0, n_iter=n_iter)
    S = plan.pack_magnitudes(mag.cuda())
    A = plan.pack_complex(angles0.cuda())
    got = plan.griffinlim(S, B, T, n_iter, 0.99, angles0_slots=A).cpu()
    return ref, got, mag


@pytest.mark.parametrize("n_iter,floor_db", [(0, 110.0), (1, 100.0), (4, 95.0), (32, 60.0)])
def test_griffinlim_injected_init_snr(plan, oparams, n_iter, floor_db):
    """Same |S| and same injected angles0 as the oracle: waveform SNR floors of SURVEY.md 8(d)
    (fp32 Griffin-Lim is chaotic: fp32-vs-fp64 itself sits at 78 dB after 32 iterations)."""
    ref, got, _ = _gl_case(plan, oparams, B=2, T=48, n_iter=n_iter, seed=1234)
    assert got.shape == ref.shape
    s = snr_db(ref, got)
    print(f"griffinlim n_iter={n_iter}: SNR {s:.1f} dB (floor {floor_db})")
    assert s >= floor_db, f"SNR {s:.1f} dB < {floor_db}"


def test_griffinlim_many_runs_equals_single_run(plan, oparams):
    """B=1 is split over many frame runs (halo path); B large gives long runs that cross clip boundaries (round 5: the batch's
    frames are cut into equal runs whatever B is, so identical clips at different places of a batch are cut at different
    frames and agree to summation order at the seams, not bit for bit): same answer."""
    import riffusion_oracle as O

    T = 120
    wave = synthetic_wave(1, 441 * (T - 1), seed=5)
    mag = O.stft_complex(wave, oparams).abs()
    g = torch.Generator().manual_seed(9)
    angles0 = torch.rand(mag.shape, dtype=torch.complex64, generator=g)
    S1 = plan.pack_magnitudes(mag.cuda())
    A1 = plan.pack_complex(angles0.cuda())
    one = plan.griffinlim(S1, 1, T, 3, 0.99, angles0_slots=A1).cpu()
    reps = 300  # more clips than CUs -> a single run per clip
    Sn = S1.repeat(reps, 1)
    An = A1.repeat(reps, 1)
    many = plan.griffinlim(Sn, reps, T, 3, 0.99, angles0_slots=An).cpu()
    assert snr_db(many[0:1], one) > 110.0
    for i in (1, 137, reps - 1):
        assert snr_db(many[0:1], many[i:i + 1]) > 110.0
    again = plan.griffinlim(Sn, reps, T, 3, 0.99, angles0_slots=An).cpu()
    assert torch.equal(many, again)  # the same launch twice: the same bits (no atomics at the run seams)
    ref = O.griffinlim(mag, oparams, angles0=angles0, n_iter=3)
    assert snr_db(ref, one) > 95.0


def test_griffinlim_is_deterministic(plan, oparams):
    ref, got, mag = _gl_case(plan, oparams, B=3, T=40, n_iter=5, seed=77)
    ref2, got2, _ = _gl_case(plan, oparams, B=3, T=40, n_iter=5, seed=77)
    assert torch.equal(got, got2)


def test_griffinlim_random_init_converges_like_oracle(plan, oparams):
    """Production path (device RNG): spectral convergence within a few % of the oracle's own value."""
    import riffusion_oracle as O

    B, T = 2, 64
    wave = synthetic_wave(B, 441 * (T - 1), seed=11)
    mag = O.stft_complex(wave, oparams).abs()
    ref = O.griffinlim(mag, oparams, generator=torch.Generator().manual_seed(3), n_iter=32)
    got = plan.griffinlim(plan.pack_magnitudes(mag.cuda()), B, T, 32, 0.99, seed=42).cpu()
    sc_ref = O.spectral_convergence(ref, mag, oparams)
    sc_got = O.spectral_convergence(got, mag, oparams)
    assert abs(sc_got - sc_ref) / sc_ref < 0.10, (sc_got, sc_ref)


@pytest.mark.parametrize("B,T", [(1, 22), (1, 57), (2, 101), (3, 568), (7, 33), (5, 200), (1, 1024)])
def test_griffinlim_shape_sweep(plan, oparams, B, T):
    """Run partitioning / halo logic over awkward shapes: few frames per run, uneven runs, long clips."""
    ref, got, _ = _gl_case(plan, oparams, B=B, T=T, n_iter=3, seed=1000 + 7 * B + T)
    assert got.shape == ref.shape == (B, 441 * (T - 1))
    assert snr_db(ref, got) >= 95.0
