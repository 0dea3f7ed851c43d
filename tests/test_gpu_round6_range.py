"""
Round 6: the numeric-range contract of the inverse path (include/rfx.h "Numeric range").

The reference's `spectrogram_from_image(max_value=...)` (image_util.py:59-108) scales the whole spectrogram by a caller-chosen
constant and its docstring says the value "in practice doesn't matter".  Until round 5 it did here: the SGD state was held times a
FIXED 2^-60 with an output clamp to [0, 1] (silent saturation above 1.15e18, flush below 1.4e-20) and Griffin-Lim's projection
squared |a| (zero phase factor once |a| > 1.8e19).  Now both work in a power of two chosen per clip / per row from the data (or
the caller's magnitude_hint), which commutes with every rounding of the linear parts.  Everything goes through the C ABI; the
oracle is the checker.
"""
import os

import numpy as np
import pytest
import torch

from helpers import snr_db

pytestmark = pytest.mark.gpu

MAX_VALUES = [1e-6, 1.0, 30e6, 1e12, 1e20]


@pytest.fixture(scope="module")
def O():
    import riffusion_oracle

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return riffusion_oracle


@pytest.fixture(scope="module")
def plan():
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    return _hip.get_plan(SpectrogramParams(), "cuda")


def _mel(max_value, C=1, Tn=24, seed=3):
    """image-like mel amplitudes: uint8 pixels through the reference's decode chain at this max_value"""
    from riffusion.util import image_util

    rng = np.random.default_rng(seed)
    lut = image_util.decode_lut(0.25, float(max_value))
    return torch.from_numpy(lut[rng.integers(0, 256, size=(C, 512, Tn))])


@pytest.mark.parametrize("max_value", MAX_VALUES)
@pytest.mark.parametrize("hint", [False, True])
def test_inverse_mel_against_oracle_over_fourteen_orders_of_magnitude(plan, O, max_value, hint):
    """InverseMelScale (SGD 200, injected start) vs the oracle at max_value 1e-6 .. 1e20: rel-L2 <= 1e-3 on the bins a filter
    reaches (measured ~1e-7), the others passed through bit for bit; with the scale taken from the data and from the hint."""
    op = O.OracleParams()
    mel = _mel(max_value, C=2)
    spec0 = torch.rand(2, mel.shape[-1], op.n_stft, generator=torch.Generator().manual_seed(8))
    want = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
    slots = plan.inverse_mel(mel.cuda(), 2, spec0=spec0.cuda(), magnitude_hint=max_value if hint else 0.0)
    got = plan.unpack_magnitudes(slots, 2, mel.shape[-1]).cpu()
    assert bool(torch.isfinite(got).all())
    act = slice(1, 4001)
    rel = float((got[:, act] - want[:, act]).norm() / want[:, act].norm())
    print(f"max_value {max_value:g} (hint {hint}): InverseMelScale rel-L2 {rel:.2e} on bins 1..4000, oracle max {float(want.max()):.3g}")
    assert rel <= 1e-3
    assert torch.equal(got[:, 4001:], want[:, 4001:]) and torch.equal(got[:, 0], want[:, 0])  # untouched bins: the start, bit for bit


@pytest.mark.parametrize("k", [-13, 17, 40, 60])
def test_inverse_mel_is_exact_under_powers_of_two(plan, k):
    """mel * 2^k gives magnitudes * 2^k BIT FOR BIT on the bins a filter reaches (the per-clip scale moves with the data, so the
    scaled iteration is the same iteration); the start is injected times 2^k as well so that the whole problem is the scaled one.
    (Not for every k: the reference's stopping rule is absolute - loss < 1e-5, |change| < 1e-8, spectrogram_converter.py:94-95 - so
    at 2^-60 it stops after the first step, here as there; the test above holds the tiny scales against the oracle.)"""
    Tn = 24
    mel = _mel(30e6, C=1, Tn=Tn)
    spec0 = torch.rand(1, Tn, plan.n_stft, generator=torch.Generator().manual_seed(9))
    base = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), 1, spec0=spec0.cuda()), 1, Tn)
    f = float(2.0 ** k)
    scaled = plan.unpack_magnitudes(plan.inverse_mel((mel * f).cuda(), 1, spec0=(spec0 * f).cuda()), 1, Tn)
    assert torch.equal((base[:, 1:4001] * f).view(torch.int32), scaled[:, 1:4001].view(torch.int32))


def test_the_internal_exponents_do_not_show_in_the_result(plan):
    """The same input with the scale taken from the data, from an exact hint and from hints 2^10 and 2^20 too large (other internal
    exponents, the same arithmetic): InverseMelScale and Griffin-Lim return the same bits - the power of two commutes with every
    rounding, which is the whole argument of the numeric-range contract."""
    Tn = 40
    mel = _mel(1e6, C=1, Tn=Tn)  # (largest amplitude below the hint's 30e6-style default on purpose)
    spec0 = torch.rand(1, Tn, plan.n_stft, generator=torch.Generator().manual_seed(2)).cuda()
    outs = [plan.inverse_mel(mel.cuda(), 1, spec0=spec0, magnitude_hint=h) for h in (0.0, 1e6, 1e6 * 2.0 ** 10, 1e6 * 2.0 ** 20)]
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32))
    S = outs[0]
    a0 = plan.pack_complex(torch.view_as_complex(torch.rand(1, plan.n_stft, Tn, 2, generator=torch.Generator().manual_seed(3))).cuda())
    waves = [plan.griffinlim(S, 1, Tn, 6, 0.99, angles0_slots=a0, magnitude_hint=h) for h in (0.0, 1e6, 1e6 * 2.0 ** 10, 1e6 * 2.0 ** 20)]
    for w in waves[1:]:
        assert torch.equal(w.view(torch.int32), waves[0].view(torch.int32))
    fused = [plan.waveform_from_mel(mel.cuda(), 1, 6, 0.99, seed=5, magnitude_hint=h) for h in (0.0, 30e6)]
    assert torch.equal(fused[0].view(torch.int32), fused[1].view(torch.int32))


@pytest.mark.parametrize("max_value", MAX_VALUES + [1e30])
def test_griffinlim_against_oracle_over_the_range(plan, O, max_value):
    """Griffin-Lim(4), injected phases, magnitudes of scale max_value: >= 95 dB vs the oracle (torch on the CPU takes |a| with
    hypot: no overflow there either).  Until round 5 the device returned silence at 1e20 (|a|^2 = inf)."""
    op = O.OracleParams()
    B, Tn = 2, 40
    g = torch.Generator().manual_seed(21)
    mag = torch.rand(B, op.n_stft, Tn, generator=g) * float(max_value)
    a0 = torch.view_as_complex(torch.rand(B, op.n_stft, Tn, 2, generator=g))
    want = O.griffinlim(mag, op, angles0=a0, n_iter=4)
    assert bool(torch.isfinite(want).all()) and float(want.abs().max()) > 0
    for hint in (0.0, float(max_value)):
        got = plan.griffinlim(plan.pack_magnitudes(mag.cuda()), B, Tn, 4, 0.99, angles0_slots=plan.pack_complex(a0.cuda()), magnitude_hint=hint).cpu()
        s = snr_db(want, got)
        print(f"max_value {max_value:g} (hint {hint:g}): Griffin-Lim(4) {s:.1f} dB vs oracle")
        assert s >= 95.0


@pytest.mark.parametrize("form", ["runs", "frames"])
@pytest.mark.parametrize("k", [-20, 33, 80])
def test_griffinlim_is_exactly_scale_equivariant_for_powers_of_two(form, k):
    """|S| * 2^k gives waveform * 2^k bit for bit, in both device forms (the analysis input is brought back to the same units by
    the row's power of two, the synthesis is linear).  (Not at 2^-70: magnitudes of 1e-14 sit in the regime of the reference's own
    `+ 1e-16` guard, where its update is not scale-equivariant either - see include/rfx.h, "Numeric range".)"""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    pl = _hip.get_plan(SpectrogramParams(), "cuda", gl_form=form)
    B, Tn = 3, 50
    g = torch.Generator(device="cuda").manual_seed(4)
    mag = torch.rand(B, pl.n_stft, Tn, device="cuda", generator=g) * 3e7
    a0 = pl.pack_complex(torch.view_as_complex(torch.rand(B, pl.n_stft, Tn, 2, device="cuda", generator=g)))
    base = pl.griffinlim(pl.pack_magnitudes(mag), B, Tn, 5, 0.99, angles0_slots=a0)
    f = float(2.0 ** k)
    scaled = pl.griffinlim(pl.pack_magnitudes(mag * f), B, Tn, 5, 0.99, angles0_slots=a0)
    assert torch.equal((base * f).view(torch.int32), scaled.view(torch.int32))


@pytest.mark.parametrize("rate", [48000, 11025])
def test_other_engines_take_huge_magnitudes_too(O, rate):
    """The row-family (48 kHz) and generic (11.025 kHz) Griffin-Lim engines at magnitudes of 1e20 against the oracle."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(sample_rate=rate)
    op = O.params_from(p)
    pl = _hip.get_plan(p, "cuda")
    B, Tn = 1, 40
    g = torch.Generator().manual_seed(2)
    mag = torch.rand(B, op.n_stft, Tn, generator=g) * 1e20
    a0 = torch.view_as_complex(torch.rand(B, op.n_stft, Tn, 2, generator=g))
    want = O.griffinlim(mag, op, angles0=a0, n_iter=3)
    got = pl.griffinlim(pl.pack_magnitudes(mag.cuda()), B, Tn, 3, 0.99, angles0_slots=pl.pack_complex(a0.cuda())).cpu()
    s = snr_db(want, got)
    print(f"{rate} Hz ({pl.griffinlim_engine}): Griffin-Lim(3) at 1e20: {s:.1f} dB vs oracle")
    assert s >= 90.0


@pytest.mark.parametrize("max_value", MAX_VALUES)
def test_image_path_yields_audio_at_every_max_value(max_value):
    """audio_from_spectrogram_images(max_value=...): full-scale, finite PCM at every max_value (round 5: silence at 1e20)."""
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    conv = SpectrogramImageConverter(SpectrogramParams(num_griffin_lim_iters=8), device="cuda")
    tiles = np.random.default_rng(1).integers(0, 256, size=(2, 512, 64, 3), dtype=np.uint8)
    pcm = conv.audio_from_spectrogram_images(tiles, max_value=max_value, seed=3)
    assert pcm.shape == (2, 441 * 63, 1) and int(np.abs(pcm.astype(np.int32)).max(axis=(1, 2)).min()) >= 32766
    # above the noise floor of the untouched bins (U[0, 1) in the reference's units) the audio is that of the default max_value
    # up to rounding in the noise floor's relative weight: a power of two keeps the image's part of the problem bit-identical
    if max_value == 30e6:
        again = conv.audio_from_spectrogram_images(tiles, max_value=30e6 * 2.0 ** 20, seed=3)
        diff = np.abs(again.astype(np.int32) - pcm.astype(np.int32))
        s = snr_db(torch.from_numpy(pcm.astype(np.float32)), torch.from_numpy(again.astype(np.float32)))
        print(f"max_value 30e6 vs 30e6 * 2^20: PCM differs by at most {int(diff.max())} steps, {s:.1f} dB (the U[0,1) floor weighs 2^-20 of what it did)")
        assert s >= 40.0


def test_unsupported_hints_are_refused(plan):
    from riffusion import _hip

    mel = _mel(30e6).cuda()
    for bad in (-1.0, float("nan"), float("inf")):
        with pytest.raises(_hip.RfxError):
            plan.inverse_mel(mel, 1, magnitude_hint=bad)
