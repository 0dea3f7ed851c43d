"""
GPU parity tests of the mel projection, InverseMelScale, the image / PCM codecs and the drop-in
classes, through the C ABI against the CPU oracle and the reference's golden fixtures.
"""
import os

import numpy as np
import pytest
import torch

from helpers import snr_db, synthetic_tiles_u8, synthetic_wave

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def params():
    from riffusion.spectrogram_params import SpectrogramParams

    return SpectrogramParams()


@pytest.fixture(scope="module")
def plan(params):
    from riffusion import _hip

    return _hip.get_plan(params, "cuda")


@pytest.fixture(scope="module")
def O():
    import riffusion_oracle

    return riffusion_oracle


# ---- forward: STFT -> |.| -> mel (MFMA) ------------------------------------------------------------
@pytest.mark.parametrize("length,batch", [(8821, 1), (441 * 130 + 5, 3), (250400, 2)])
def test_mel_amplitudes_match_oracle(plan, params, O, length, batch):
    wave = synthetic_wave(batch, length, seed=length + 1)
    ref = O.mel_amplitudes_from_waveform(wave, O.params_from(params))
    got = plan.mel_from_waveform(wave.cuda()).cpu()
    assert got.shape == ref.shape
    # SURVEY 8(d) gate: max |diff| <= 1e-4 * max(ref) per tile and rel-L2 <= 1e-4
    for b in range(batch):
        assert (got[b] - ref[b]).abs().max() <= 1e-4 * ref[b].max()
    assert torch.linalg.norm(got - ref) / torch.linalg.norm(ref) <= 1e-4


def test_fused_forward_equals_stft_then_mfma_melscale(plan, params, O):
    """`mel_from_waveform` runs the fused kernel (framed transform -> |X| -> banded projection on chip); the standalone
    members Spectrogram -> abs -> MelScale (MFMA GEMM over all slot positions) must give the same amplitudes."""
    wave = synthetic_wave(3, 441 * 200 + 17, seed=9)
    fused = plan.mel_from_waveform(wave.cuda())
    mag, _, Tn = plan.stft(wave.cuda(), want_mag=True, want_spec=False)
    dense = plan.mel_scale(plan.unpack_magnitudes(mag, 3, Tn))
    assert fused.shape == dense.shape == (3, 512, Tn)
    assert float((fused - dense).abs().max() / dense.abs().max()) <= 2e-6
    ref = O.mel_amplitudes_from_waveform(wave, O.params_from(params))
    assert torch.linalg.norm(fused.cpu() - ref) / torch.linalg.norm(ref) <= 1e-4


def test_mel_other_frequency_range(O):
    """20 Hz .. 20 kHz parameters of the reference's own round-trip test (spectrogram_converter_test.py:46-53)."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(min_frequency=20, max_frequency=20000)
    plan = _hip.get_plan(p, "cuda")
    wave = synthetic_wave(2, 441 * 60, seed=3)
    ref = O.mel_amplitudes_from_waveform(wave, O.params_from(p))
    got = plan.mel_from_waveform(wave.cuda()).cpu()
    assert torch.linalg.norm(got - ref) / torch.linalg.norm(ref) <= 1e-4


def test_golden_png_from_clip(plan, params, golden_dir):
    """The reference's own fixture: clip_2 wav -> stereo PNG with EXIF MAX_VALUE 46801012.0."""
    from PIL import Image
    from scipy.io import wavfile

    from riffusion.util import image_util

    sr, w = wavfile.read(os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms.wav"))
    assert sr == 44100
    wave = torch.from_numpy(w.T.astype(np.float32))
    mel = plan.mel_from_waveform(wave.cuda())
    thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
    img, mx = plan.image_encode(mel, True, thr)
    gold = np.array(Image.open(os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms_stereo.png")).convert("RGB"))
    got = img[0].cpu().numpy()
    assert got.shape == gold.shape == (512, 568, 3)
    assert float(mx[0]) == pytest.approx(46801012.0, rel=2e-6)
    diff = np.abs(got.astype(int) - gold.astype(int))
    assert diff.max() <= 1
    assert (diff == 0).mean() > 0.998
    assert np.all(got[:, :, 0] == 0)  # stereo => R plane empty (audio_to_image_test.py:83)


# ---- image codec --------------------------------------------------------------------------------------
@pytest.mark.parametrize("stereo", [False, True])
def test_image_decode_bit_exact(plan, O, stereo):
    from riffusion.util import image_util

    tiles = synthetic_tiles_u8(3, 512, 512)
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).cuda()
    got = plan.image_decode(torch.from_numpy(tiles).cuda(), stereo, lut).cpu().numpy()
    C = 2 if stereo else 1
    for n in range(3):
        ref = O.spectrogram_from_image_u8(tiles[n], 0.25, stereo, 30e6)
        assert np.array_equal(got[n * C : (n + 1) * C], ref)


@pytest.mark.parametrize("stereo", [False, True])
def test_image_encode_bit_exact(plan, O, stereo):
    from riffusion.util import image_util

    rng = np.random.default_rng(5)
    C = 2 if stereo else 1
    N = 3
    mel = (rng.random((N * C, 512, 97), dtype=np.float32) ** 4 * 3e7).astype(np.float32)
    # plant exact threshold values (and their float32 neighbours) to hit every decision boundary
    thr = image_util.encode_thresholds(0.25)
    bits = thr.view(np.uint32).astype(np.int64)
    cand = np.unique(np.clip(np.concatenate([bits - 1, bits, bits + 1]), 0, 0x3F800000)).astype(np.uint32).view(np.float32)
    mel[0, 0, :] = 0
    mel[0].reshape(-1)[: cand.size] = cand * np.float32(2.0**20)
    mel[0, -1, -1] = np.float32(2.0**20)  # the max of clip 0: ratios are exact
    got, mx = plan.image_encode(torch.from_numpy(mel).cuda(), stereo, torch.from_numpy(thr).cuda())
    got = got.cpu().numpy()
    for n in range(N):
        clip = mel[n * C : (n + 1) * C]
        assert float(mx[n]) == float(clip.max())
        assert np.array_equal(got[n], O.image_u8_from_spectrogram(clip, 0.25))


def test_pcm16_bit_exact(plan, O):
    rng = np.random.default_rng(9)
    for C in (1, 2):
        wave = (rng.standard_normal((3 * C, 5000)) * rng.uniform(0.1, 5)).astype(np.float32)
        pcm, peak = plan.pcm16(torch.from_numpy(wave).cuda(), channels=C, normalize=True)
        pcm = pcm.cpu().numpy()
        for n in range(3):
            ref = O.pcm16_from_waveform(wave[n * C : (n + 1) * C], normalize=True)
            assert np.array_equal(pcm[n], ref)


# ---- InverseMelScale -------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,T", [(1, 24), (2, 16)])
def test_inverse_mel_sgd_matches_oracle(plan, params, O, C, T):
    op = O.params_from(params)
    g = torch.Generator().manual_seed(77)
    tiles = synthetic_tiles_u8(1, 512, T, seed=4)
    mel = torch.from_numpy(O.spectrogram_from_image_u8(tiles[0], 0.25, C == 2, 30e6))  # (C, 512, T)
    spec0 = torch.rand(C, T, op.n_stft, generator=g)
    ref, steps = O.inverse_mel_scale_sgd(mel, op, spec0=spec0, return_iters=True)
    assert steps == 200  # never stops early at this scale
    slots = plan.inverse_mel(mel.cuda(), C, spec0=spec0.cuda())
    got = plan.unpack_magnitudes(slots, C, T).cpu()
    fb = O.mel_filterbank(op)
    active = fb.abs().sum(1) > 0
    rel = torch.linalg.norm(got[:, active] - ref[:, active]) / torch.linalg.norm(ref[:, active])
    assert rel <= 1e-3, rel
    # untouched bins pass the init through bit-for-bit
    assert torch.equal(got[:, ~active], spec0.transpose(1, 2)[:, ~active])
    # duplicate slots carry the same value as their primary
    repacked = plan.pack_magnitudes(got.cuda())
    assert torch.equal(repacked, slots)


def test_inverse_mel_clip_coupling(plan, params, O):
    """The 1/(C*T) factor: solving two channels as one clip differs from solving them separately."""
    op = O.params_from(params)
    T = 12
    tiles = synthetic_tiles_u8(1, 512, T, seed=8)
    mel = torch.from_numpy(O.spectrogram_from_image_u8(tiles[0], 0.25, True, 30e6))
    spec0 = torch.rand(2, T, op.n_stft, generator=torch.Generator().manual_seed(1))
    together = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), 2, spec0=spec0.cuda()), 2, T).cpu()
    apart = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), 1, spec0=spec0.cuda()), 2, T).cpu()
    ref_together = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
    ref_apart = torch.cat([O.inverse_mel_scale_sgd(mel[i : i + 1], op, spec0=spec0[i : i + 1]) for i in range(2)])
    assert torch.linalg.norm(together - ref_together) / torch.linalg.norm(ref_together) < 1e-3
    assert torch.linalg.norm(apart - ref_apart) / torch.linalg.norm(ref_apart) < 1e-3
    assert torch.linalg.norm(together - apart) / torch.linalg.norm(apart) > 1e-2


def test_inverse_mel_early_stop(O):
    """A zero target reaches loss < 1e-5 at once: the reference stops after its first step."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(max_mel_iters=40)
    op = O.params_from(p)
    plan = _hip.get_plan(p, "cuda")
    T = 10
    mel = torch.zeros(1, 512, T)
    spec0 = torch.rand(1, T, op.n_stft, generator=torch.Generator().manual_seed(2)) * 1e-4
    ref, steps = O.inverse_mel_scale_sgd(mel, op, spec0=spec0, return_iters=True)
    assert steps < 40
    got = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), 1, spec0=spec0.cuda()), 1, T).cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-9)


# ---- drop-in classes ------------------------------------------------------------------------------------------
def test_converter_round_trip_api(params, O):
    """mel -> waveform with both random initialisations injected, against the oracle end to end."""
    from riffusion.spectrogram_converter import SpectrogramConverter

    conv = SpectrogramConverter(params, device="cuda")
    op = O.params_from(params)
    T = 32
    tiles = synthetic_tiles_u8(1, 512, T, seed=21)
    mel = torch.from_numpy(O.spectrogram_from_image_u8(tiles[0], 0.25, False, 30e6))
    g = torch.Generator().manual_seed(1234)
    spec0 = torch.rand(1, T, op.n_stft, generator=g)
    angles0 = torch.rand(1, op.n_stft, T, dtype=torch.complex64, generator=g)
    ref = O.waveform_from_mel_amplitudes(mel, op, spec0=spec0, angles0=angles0)
    got = conv.waveform_from_mel_amplitudes(mel, spec0=spec0, angles0=angles0).cpu()
    assert got.shape == ref.shape == (1, 441 * (T - 1))
    assert snr_db(ref, got) >= 50.0  # 32 chaotic iterations on top of a 1e-3-class SGD difference
    # members exist and behave like the reference's modules
    lin = conv.inverse_mel_scaler(mel, spec0=spec0)
    assert lin.shape == (1, op.n_stft, T)
    spec = conv.spectrogram_func(got)
    assert spec.shape == (1, op.n_stft, T) and spec.dtype == torch.complex64
    # the four members chain exactly like the reference's forward path (:179-185)
    mel2 = conv.mel_scaler(torch.abs(spec))
    fused = conv.mel_amplitudes_from_waveform(got)
    assert mel2.shape == fused.shape and torch.linalg.norm(mel2 - fused) / torch.linalg.norm(fused) < 1e-5
    ref_mel = O.mel_scale(torch.abs(spec).cpu(), O.mel_filterbank(op))
    assert torch.linalg.norm(mel2.cpu() - ref_mel) / torch.linalg.norm(ref_mel) < 1e-5


def test_image_converter_batch_and_single(params, golden_dir):
    from PIL import Image

    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.util import audio_util

    ic = SpectrogramImageConverter(params, device="cuda")
    img = Image.open(os.path.join(golden_dir, "og_beat_64.png"))
    seg = ic.audio_from_spectrogram_image(img, apply_filters=False)
    assert seg.frame_rate == 44100 and seg.channels == 1 and seg.sample_width == 2
    assert abs(seg.duration_seconds - 441 * (img.size[0] - 1) / 44100) < 0.01
    # audio -> image: size, mode, EXIF round trip (audio_to_image_test.py:68-99)
    clip = audio_util.PcmSegment.from_wav(os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms.wav"))
    out = ic.spectrogram_image_from_audio(clip)
    assert out.mode == "RGB" and out.size == (568, 512)
    arr = np.array(out)
    assert np.array_equal(arr[..., 0], arr[..., 1]) and np.array_equal(arr[..., 0], arr[..., 2])
    from riffusion.spectrogram_params import SpectrogramParams

    assert SpectrogramParams.from_exif(out.getexif()) == params


def test_stereo_tiles_64_iterations_end_to_end(O):
    """BASELINE.json configs[3] in miniature: stereo tiles, Griffin-Lim 64, batch of clips through the
    image-level batch entry point; parity with injected initialisations at the 64-iteration floor."""
    from riffusion.spectrogram_converter import SpectrogramConverter
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(stereo=True, num_griffin_lim_iters=64)
    op = O.params_from(p)
    T, N = 24, 3
    tiles = synthetic_tiles_u8(N, 512, T, seed=31)
    ic = SpectrogramImageConverter(p, device="cuda")
    pcm = ic.audio_from_spectrogram_images(tiles, seed=7)
    assert pcm.shape == (N, 441 * (T - 1), 2) and pcm.dtype == np.int16
    assert np.abs(pcm.astype(int)).max(axis=(1, 2)).min() >= 32766  # every clip peak-normalised on its own
    assert not np.array_equal(pcm[0], pcm[1])
    # injected-init parity of one stereo clip (its two channels share the SGD loss mean)
    conv = SpectrogramConverter(p, device="cuda")
    mel = torch.from_numpy(O.spectrogram_from_image_u8(tiles[0], 0.25, True, 30e6))
    g = torch.Generator().manual_seed(5)
    spec0 = torch.rand(2, T, op.n_stft, generator=g)
    angles0 = torch.rand(2, op.n_stft, T, dtype=torch.complex64, generator=g)
    ref = O.waveform_from_mel_amplitudes(mel, op, spec0=spec0, angles0=angles0)
    got = conv.waveform_from_mel_amplitudes(mel, spec0=spec0, angles0=angles0).cpu()
    assert snr_db(ref, got) >= 40.0  # >= 40 dB at 64 iterations


@pytest.mark.parametrize("kw", [dict(min_frequency=20, max_frequency=20000), dict(num_frequencies=256), dict(num_frequencies=128)])
def test_inverse_mel_other_parameter_sets_use_fallback_kernels(O, kw):
    """Mel parameters outside the default bank's register budgets: 20 Hz .. 20 kHz takes the line-form group kernel since round 5
    (tests/test_gpu_round5.py), so do 256 filters; 128 filters (groups of up to 90 bins) the general banded kernel."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(max_mel_iters=60, **kw)
    op = O.params_from(p)
    plan = _hip.get_plan(p, "cuda")
    T, C = 6, 1
    g = torch.Generator().manual_seed(17)
    mel = torch.rand(C, p.num_frequencies, T, generator=g) ** 3 * 2e7
    spec0 = torch.rand(C, T, op.n_stft, generator=g)
    ref = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
    got = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), C, spec0=spec0.cuda()), C, T).cpu()
    active = O.mel_filterbank(op).abs().sum(1) > 0
    assert torch.linalg.norm(got[:, active] - ref[:, active]) / torch.linalg.norm(ref[:, active]) <= 1e-3
    assert torch.equal(got[:, ~active], spec0.transpose(1, 2)[:, ~active])
    # and the forward projection with the same parameters
    wave = synthetic_wave(1, 441 * 40, seed=2)
    fwd = plan.mel_from_waveform(wave.cuda()).cpu()
    fref = O.mel_amplitudes_from_waveform(wave, op)
    assert torch.linalg.norm(fwd - fref) / torch.linalg.norm(fref) <= 1e-4


@pytest.mark.parametrize("n_mels", [1024, 96])
def test_other_filter_counts_forward_and_inverse(O, n_mels):
    """num_frequencies other than 512 (the reference exposes it: cli.py:27, spectrogram_params.py:31): 1024 filters take the
    unfused STFT + MFMA GEMM path forward and the general SGD kernel inverse, 96 the fused / group paths' fallbacks."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(num_frequencies=n_mels, max_mel_iters=40)
    op = O.params_from(p)
    plan = _hip.get_plan(p, "cuda")
    wave = synthetic_wave(2, 441 * 40 + 7, seed=n_mels)
    ref = O.mel_amplitudes_from_waveform(wave, op)
    got = plan.mel_from_waveform(wave.cuda()).cpu()
    assert got.shape == ref.shape == (2, n_mels, 41)
    assert torch.linalg.norm(got - ref) / torch.linalg.norm(ref) <= 1e-4
    g = torch.Generator().manual_seed(1)
    spec0 = torch.rand(2, 41, op.n_stft, generator=g)
    want = O.inverse_mel_scale_sgd(ref, op, spec0=spec0)
    lin = plan.unpack_magnitudes(plan.inverse_mel(ref.cuda(), 2, spec0=spec0.cuda()), 2, 41).cpu()
    act = O.mel_filterbank(op).abs().sum(1) > 0
    rel = float(torch.linalg.norm(lin[:, act] - want[:, act]) / torch.linalg.norm(want[:, act]))
    print(f"n_mels {n_mels}: forward ok, InverseMelScale rel-L2 {rel:.2e}")
    assert rel <= 1e-3 and torch.equal(lin[:, ~act], want[:, ~act])
