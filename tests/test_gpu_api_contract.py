"""GPU tests of the boundary contract: errors, re-entrancy, plan cache, edge shapes."""
import ctypes
import time
from multiprocessing.pool import ThreadPool

import numpy as np
import pytest
import torch

from helpers import synthetic_tiles_u8, synthetic_wave

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def params():
    from riffusion.spectrogram_params import SpectrogramParams

    return SpectrogramParams()


def test_plan_cache_and_cheap_construction(params):
    from riffusion import _hip
    from riffusion.spectrogram_converter import SpectrogramConverter
    from riffusion.spectrogram_params import SpectrogramParams

    SpectrogramConverter(params, device="cuda")._plan()
    assert _hip.get_plan(params, "cuda") is _hip.get_plan(SpectrogramParams(), "cuda")
    t0 = time.perf_counter()
    for _ in range(50):  # the reference's server builds a converter per request (server.py:159)
        SpectrogramConverter(SpectrogramParams(), device="cuda")._plan()
    assert (time.perf_counter() - t0) / 50 < 2e-3


def test_errors_match_reference_behaviour(params):
    from riffusion import _hip
    from riffusion.spectrogram_converter import SpectrogramConverter
    from riffusion.spectrogram_params import SpectrogramParams

    conv = SpectrogramConverter(params, device="cuda")
    with pytest.raises(RuntimeError, match="Padding size should be less than"):  # torch.stft's reflect-pad error
        conv.mel_amplitudes_from_waveform(torch.zeros(1, 8820))
    conv.mel_amplitudes_from_waveform(torch.zeros(1, 8821))  # smallest legal input
    with pytest.raises(ValueError, match="mel bins"):  # torchaudio InverseMelScale's check
        conv.waveform_from_mel_amplitudes(torch.zeros(1, 100, 16))
    # other sample rates run on the generic engine; only FFT lengths with a prime factor above 13 are refused, with a reason
    assert SpectrogramConverter(SpectrogramParams(sample_rate=48000), device="cuda").mel_amplitudes_from_waveform(torch.zeros(1, 30000)).shape == (1, 512, 63)
    with pytest.raises(_hip.RfxError, match="prime factor"):
        SpectrogramConverter(SpectrogramParams(sample_rate=42570), device="cuda").mel_amplitudes_from_waveform(torch.zeros(1, 30000))
    plan = _hip.get_plan(params, "cuda")
    with pytest.raises(_hip.RfxError, match="multiple of channels_per_clip"):
        plan.inverse_mel(torch.zeros(3, 512, 8, device="cuda"), 2)
    # too small a workspace is an error code, not a crash
    S = torch.zeros(2 * 24, plan.frame_stride, device="cuda")
    out = torch.empty(2, 441 * 23, device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = plan.lib.rfx_griffinlim(plan.handle, S.data_ptr(), None, 0, 2, 24, 2, ctypes.c_float(0.99), out.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _hip.current_stream())
    assert rc == -3 and b"workspace" in plan.lib.rfx_last_error()


def test_edge_shapes(params):
    """Shortest legal clips (22 frames: reflect padding needs > 8820 samples), odd batches, and the
    reference's own failure below that."""
    import riffusion_oracle as O
    from riffusion import _hip

    plan = _hip.get_plan(params, "cuda")
    op = O.params_from(params)
    with pytest.raises(_hip.RfxError, match="Padding size should be less than"):
        plan.griffinlim(torch.zeros(10, plan.frame_stride, device="cuda"), 1, 10, 2, 0.99)
    with pytest.raises(RuntimeError, match="Padding size should be less than"):
        O.griffinlim(torch.rand(1, op.n_stft, 10), op, n_iter=2)
    # zero iterations = one ISTFT: legal for any T >= 2
    g = torch.Generator().manual_seed(0)
    mag = torch.rand(1, op.n_stft, 2, generator=g)
    a0 = torch.rand(1, op.n_stft, 2, dtype=torch.complex64, generator=g)
    ref0 = O.griffinlim(mag, op, angles0=a0, n_iter=0)
    got0 = plan.griffinlim(plan.pack_magnitudes(mag.cuda()), 1, 2, 0, 0.99, angles0_slots=plan.pack_complex(a0.cuda())).cpu()
    assert got0.shape == ref0.shape == (1, 441) and (got0 - ref0).abs().max() / ref0.abs().max() < 1e-5
    for T, B in ((22, 1), (23, 3), (31, 2)):
        g = torch.Generator().manual_seed(T)
        mag = torch.rand(B, op.n_stft, T, generator=g) * 100
        a0 = torch.rand(B, op.n_stft, T, dtype=torch.complex64, generator=g)
        ref = O.griffinlim(mag, op, angles0=a0, n_iter=2)
        got = plan.griffinlim(plan.pack_magnitudes(mag.cuda()), B, T, 2, 0.99, angles0_slots=plan.pack_complex(a0.cuda())).cpu()
        assert got.shape == ref.shape == (B, 441 * (T - 1))
        assert (got - ref).abs().max() / ref.abs().max() < 1e-4


def test_shared_converter_is_reentrant(params):
    """One converter used from a ThreadPool, as the reference's batch CLI does (cli.py:172-204)."""
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter

    ic = SpectrogramImageConverter(params, device="cuda")
    tiles = synthetic_tiles_u8(6, 512, 24, seed=3)

    def job(i):
        return ic.audio_from_spectrogram_images(tiles[i : i + 1], seed=100 + i)

    serial = [job(i) for i in range(6)]
    with ThreadPool(3) as pool:
        threaded = pool.map(job, range(6))
    for a, b in zip(serial, threaded):
        assert np.array_equal(a, b)
    # and seeds matter / repeat
    assert not np.array_equal(serial[0], ic.audio_from_spectrogram_images(tiles[0:1], seed=999))
    assert np.array_equal(serial[0], ic.audio_from_spectrogram_images(tiles[0:1], seed=100))


def test_forward_batch_equals_per_clip(params):
    from riffusion import _hip

    plan = _hip.get_plan(params, "cuda")
    wave = synthetic_wave(5, 441 * 37 + 100, seed=12).cuda()
    full = plan.mel_from_waveform(wave)
    for i in range(5):
        assert torch.equal(full[i : i + 1], plan.mel_from_waveform(wave[i : i + 1]))


def test_cli_round_trips_like_the_reference_tests(golden_dir, tmp_path):
    """reference test/image_to_audio_test.py:40-67 and test/audio_to_image_test.py:57-99 through our CLI."""
    import os

    from PIL import Image

    from riffusion import cli
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import audio_util

    png = os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms_stereo.png")
    wav_out = str(tmp_path / "out.wav")
    cli.main(["image-to-audio", "--image", png, "--audio", wav_out, "--device", "cuda"])
    seg = audio_util.PcmSegment.from_wav(wav_out)
    assert seg.frame_rate == 44100 and seg.channels == 2
    assert abs(seg.duration_seconds - 5.678) < 0.010 + 1.0 / 100  # duration within one hop of the clip (reference: 10 ms)
    png_out = str(tmp_path / "out.png")
    cli.main(["audio-to-image", "--audio", os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms.wav"),
              "--image", png_out, "--stereo"])
    im = Image.open(png_out)
    assert im.mode == "RGB" and im.size == (568, 512) and np.all(np.array(im)[..., 0] == 0)
    assert SpectrogramParams.from_exif(im.getexif()) == SpectrogramParams(stereo=True)
    gold = np.array(Image.open(png).convert("RGB"))
    assert np.abs(np.array(im).astype(int) - gold.astype(int)).max() <= 1
    # batch command: three copies of a tile decode in one GPU call
    tiles = tmp_path / "tiles"
    tiles.mkdir()
    for i in range(3):
        Image.open(os.path.join(golden_dir, "og_beat_64.png")).save(str(tiles / f"t{i}.png"))
    cli.main(["images-to-audio-batch", "--image-dir", str(tiles), "--output-dir", str(tmp_path / "wavs")])
    assert sorted(os.listdir(tmp_path / "wavs")) == ["t0.wav", "t1.wav", "t2.wav"]


def test_cli_other_sample_rates_and_batch_flags(golden_dir, tmp_path):
    """The reference takes the sample rate from the input file (cli.py:43): a 48 kHz clip goes through the generic engine, and
    `audio-to-images-batch` resamples mixed-rate files to --sample-rate, skips what it cannot read and defaults to stereo jpg
    (cli.py:134-204)."""
    import os

    from PIL import Image

    from riffusion import cli
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import audio_util

    src = audio_util.PcmSegment.from_wav(os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms.wav"))
    wav48 = str(tmp_path / "clip48.wav")
    src.set_frame_rate(48000).export(wav48, format="wav")
    png48 = str(tmp_path / "clip48.png")
    cli.main(["audio-to-image", "--audio", wav48, "--image", png48])
    im = Image.open(png48)
    p = SpectrogramParams.from_exif(im.getexif())
    assert p.sample_rate == 48000 and p.n_fft == 19200 and im.size[1] == 512 and im.size[0] == 1 + int(48000 * 250400 / 44100) // 480
    back = str(tmp_path / "back48.wav")
    cli.main(["image-to-audio", "--image", png48, "--audio", back])
    seg = audio_util.PcmSegment.from_wav(back)
    assert seg.frame_rate == 48000 and seg.channels == 1 and abs(seg.duration_seconds - 5.678) < 0.02

    clips = tmp_path / "clips"
    clips.mkdir()
    src.export(str(clips / "a.wav"), format="wav")
    src.set_frame_rate(22050).set_channels(1).export(str(clips / "b.wav"), format="wav")
    (clips / "broken.wav").write_bytes(b"not a wav file")
    out = tmp_path / "images"
    cli.main(["audio-to-images-batch", "--audio-dir", str(clips), "--output-dir", str(out)])
    assert sorted(os.listdir(out)) == ["a.jpg", "b.jpg"]  # the unreadable file is skipped, like the reference does
    for name in ("a.jpg", "b.jpg"):
        with Image.open(str(out / name)) as im:
            q = SpectrogramParams.from_exif(im.getexif())
            assert q.stereo is True and q.sample_rate == 44100 and im.size[1] == 512 and abs(im.size[0] - 568) <= 1
    cli.main(["audio-to-images-batch", "--audio-dir", str(clips), "--output-dir", str(tmp_path / "mono"), "--mono",
              "--image-extension", "png", "--limit", "1"])
    assert os.listdir(tmp_path / "mono") == ["a.png"]
