"""CPU tests of the host-side mirror of the reference interface and of the C-ABI library's surface."""
import ctypes
import os
import re
import sys
import warnings

import numpy as np
import pytest
import torch

from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import audio_util, image_util


@pytest.fixture(scope="module")
def vectors(golden_dir):
    return np.load(os.path.join(golden_dir, "image_codec_vectors.npz"))


# ---- SpectrogramParams ---------------------------------------------------------------------------------
def test_params_defaults_and_derived_lengths(vectors):
    p = SpectrogramParams()
    assert (p.n_fft, p.win_length, p.hop_length) == (17640, 4410, 441)
    assert (p.num_griffin_lim_iters, p.max_mel_iters, p.power_for_image) == (32, 200, 0.25)
    sets = ({}, {"stereo": True}, {"sample_rate": 48000}, {"sample_rate": 22050, "step_size_ms": 5},
            {"min_frequency": 20, "max_frequency": 20000, "num_frequencies": 256})
    for kw, row in zip(sets, vectors["params_rows"]):  # rows produced by the reference's own class
        q = SpectrogramParams(**kw)
        assert [q.n_fft, q.win_length, q.hop_length] == list(row[:3])
        assert [float(v) for v in q.to_exif().values()] == list(row[3:])


def test_params_exif_round_trip_and_hash():
    p = SpectrogramParams(stereo=True, min_frequency=20, max_frequency=20000)
    exif = p.to_exif()
    assert set(exif) == {11000, 11005, 11010, 11020, 11030, 11040, 11050, 11060, 11070}
    assert SpectrogramParams.from_exif(exif) == p
    assert hash(p) == hash(SpectrogramParams(stereo=True, min_frequency=20, max_frequency=20000))
    assert SpectrogramParams.ExifTags.MAX_VALUE.value == 11080
    with pytest.raises(KeyError):
        SpectrogramParams.from_exif({11000: 44100})
    with pytest.raises(Exception):
        p.stereo = False  # frozen


# ---- image codec host side ----------------------------------------------------------------------------
def test_image_util_equals_reference_vectors(vectors):
    from PIL import Image

    for tag, C in (("mono", 1), ("stereo", 2)):
        spec = vectors[f"enc_{tag}_in"]
        img = image_util.image_from_spectrogram(spec, 0.25)
        assert img.mode == "RGB" and np.array_equal(np.array(img), vectors[f"enc_{tag}_out"])
        got = image_util.spectrogram_from_image(img, 0.25, C == 2, 30e6)
        assert got.dtype == np.float32 and np.array_equal(got, vectors[f"dec_{tag}_30e6"])
    gray = Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16), mode="L")
    assert np.array_equal(image_util.spectrogram_from_image(gray, 0.25, False, 30e6), vectors["dec_all256"])
    ramp = np.linspace(0.0, 1.0, 65536, dtype=np.float32).reshape(1, 256, 256)
    assert np.array_equal(np.array(image_util.image_from_spectrogram(ramp, 0.25))[..., 0], vectors["enc_ramp_out"])
    with pytest.raises(NotImplementedError):
        image_util.image_from_spectrogram(np.ones((3, 4, 4), np.float32))


def test_threshold_table_is_consistent():
    thr = image_util.encode_thresholds(0.25)
    assert thr.shape == (255,) and np.all(np.diff(thr) <= 0) and thr[0] < 1.0 and thr[-1] > 0.0
    # each threshold is the first float32 that quantises to <= v: its predecessor quantises to v + 1
    bits = thr.view(np.uint32)
    below = (bits - 1).view(np.float32)
    q_at = image_util._quantise_ratio(thr, 0.25).astype(int)
    q_below = image_util._quantise_ratio(below, 0.25).astype(int)
    assert np.all(q_at <= np.arange(255)) and np.all(q_below > np.arange(255))


def test_exif_from_image(golden_dir):
    from PIL import Image

    im = Image.open(os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms_stereo.png"))
    tags = image_util.exif_from_image(im)
    assert tags["MAX_VALUE"] == 46801012.0 and tags["STEREO"] == 1 and tags["SAMPLE_RATE"] == 44100
    assert SpectrogramParams.from_exif(im.getexif()) == SpectrogramParams(stereo=True)
    assert image_util.exif_from_image(Image.open(os.path.join(golden_dir, "og_beat_64.png"))) == {}


# ---- PCM tail -------------------------------------------------------------------------------------------------
def test_pcm16_from_waveform_matches_oracle_and_reference_semantics():
    import riffusion_oracle as O

    rng = np.random.default_rng(0)
    x = (rng.standard_normal((2, 1000)) * 3.7).astype(np.float32)
    a = audio_util.pcm16_from_waveform(x, normalize=True)
    assert a.dtype == np.int16 and a.shape == (1000, 2)
    assert np.array_equal(a, O.pcm16_from_waveform(x, normalize=True))
    assert np.abs(a).max() in (32766, 32767)
    assert np.array_equal(audio_util.pcm16_from_waveform(np.array([[1.9, -1.9, 0.4]], np.float32)), [[1], [-1], [0]])


def test_pcm_segment_interface(golden_dir, tmp_path):
    seg = audio_util.PcmSegment.from_wav(os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms.wav"))
    assert (seg.frame_rate, seg.channels, seg.sample_width) == (44100, 2, 2)
    assert abs(seg.duration_seconds - 5.678) < 0.001 and abs(len(seg) - 5678) <= 1
    monos = seg.split_to_mono()
    assert len(monos) == 2 and len(monos[0].get_array_of_samples()) == 250400
    assert seg.set_channels(1).channels == 1 and monos[0].set_channels(2).channels == 2
    out = tmp_path / "x.wav"
    seg.export(str(out), format="wav")
    back = audio_util.PcmSegment.from_wav(str(out))
    assert np.array_equal(back.get_array_of_samples(), seg.get_array_of_samples())
    made = audio_util.audio_from_waveform(np.ones((1, 50), np.float32), 44100, normalize=True)
    assert made.frame_rate == 44100 and made.channels == 1


# ---- converter surface without a GPU -----------------------------------------------------------------
def test_converter_constructs_and_refuses_cpu_compute():
    from riffusion.spectrogram_converter import SpectrogramConverter
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter

    if torch.cuda.is_available():
        pytest.skip("CPU-only behaviour")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        conv = SpectrogramConverter(SpectrogramParams(), device="cuda")
    assert conv.device == "cpu" and any("not available" in str(x.message) for x in w)
    for member in ("spectrogram_func", "inverse_spectrogram_func", "mel_scaler", "inverse_mel_scaler"):
        assert callable(getattr(conv, member))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        conv.mel_amplitudes_from_waveform(torch.zeros(1, 20000))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        conv.waveform_from_mel_amplitudes(torch.zeros(1, 512, 16))
    ic = SpectrogramImageConverter(SpectrogramParams(), device="cpu")
    assert ic.p == SpectrogramParams() and ic.device == "cpu" and isinstance(ic.converter, SpectrogramConverter)
    seg = audio_util.PcmSegment(np.zeros((1000, 1), np.int16), 22050)
    with pytest.raises(AssertionError):
        ic.spectrogram_image_from_audio(seg)  # sample-rate mismatch asserts like the reference


def test_filterbank_matches_oracle_bitwise():
    import riffusion_oracle as O
    from riffusion import _hip

    for kw in ({}, {"min_frequency": 20, "max_frequency": 20000}, {"num_frequencies": 256},
               # the two mel parameters the reference exposes besides the defaults (spectrogram_params.py:34-35)
               {"mel_scale_type": "slaney"}, {"mel_scale_norm": "slaney"}, {"mel_scale_type": "slaney", "mel_scale_norm": "slaney"},
               {"mel_scale_type": "slaney", "sample_rate": 48000, "max_frequency": 16000}):
        p = SpectrogramParams(**kw)
        fb = _hip.mel_filterbank(p.n_fft // 2 + 1, float(p.min_frequency), float(p.max_frequency), p.num_frequencies,
                                 p.sample_rate, p.mel_scale_norm, p.mel_scale_type)
        assert torch.equal(fb, O.mel_filterbank(O.params_from(p)))
    assert torch.equal(_hip.hann_window(4410), O.hann_window(O.OracleParams()))
    # slaney normalisation rescales the columns of the HTK bank without touching its sparsity pattern (same SGD kernel);
    # the slaney scale moves the filter edges (linear below 1 kHz, logarithmic above)
    htk = _hip.mel_filterbank(8821, 0.0, 10000.0, 512, 44100, None, "htk")
    htk_n = _hip.mel_filterbank(8821, 0.0, 10000.0, 512, 44100, "slaney", "htk")
    sl = _hip.mel_filterbank(8821, 0.0, 10000.0, 512, 44100, None, "slaney")
    assert torch.equal(htk != 0, htk_n != 0) and not torch.equal(htk != 0, sl != 0)
    assert float(htk.max()) <= 1.0 and float(htk_n.max()) < 0.5
    for bad in ({"norm": "l2", "mel_scale": "htk"}, {"norm": None, "mel_scale": "bark"}):
        with pytest.raises(ValueError):
            _hip.mel_filterbank(8821, 0.0, 10000.0, 512, 44100, bad["norm"], bad["mel_scale"])


# ---- C ABI surface ------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(repo_root):
    from riffusion import _hip

    header = open(os.path.join(repo_root, "include", "rfx.h")).read()
    declared = set(re.findall(r"\b(rfx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    lib_path = _hip.library_path()
    if not os.path.exists(lib_path):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name
    lib.rfx_frame_stride.restype = ctypes.c_int
    lib.rfx_num_bins.restype = ctypes.c_int
    assert lib.rfx_frame_stride() == 9408 and lib.rfx_num_bins() == 8821  # no GPU needed for these


def test_product_never_imports_the_oracle(repo_root):
    pkg = os.path.join(repo_root, "riffusion-hobby_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "riffusion_oracle" not in text and "oracle/" not in text.replace("the oracle/", ""), f


# ---- CLI surface ------------------------------------------------------------------------------------------------
def test_cli_print_exif_and_parser(golden_dir, capsys):
    from riffusion import cli

    cli.main(["print-exif", "--image", os.path.join(golden_dir, "clip_2_start_103694_ms_duration_5678_ms.png")])
    out = capsys.readouterr().out
    # the two formatted lines the reference's print_exif_test.py:29-32 looks for
    assert "NUM_FREQUENCIES      =             512" in out
    assert "SAMPLE_RATE          =           44100" in out
    ns = cli.build_parser().parse_args(["audio-to-image", "--audio", "a.wav", "--image", "b.png", "--stereo", "--max-frequency", "20000"])
    assert ns.stereo is True and ns.max_frequency == 20000 and ns.step_size_ms == 10 and ns.device == "cuda"
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["image-to-audio", "--image", "x.png"])  # --audio is required


def test_pcm_segment_gain_filters():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((5000, 2)) * 3000).astype(np.int16)
    seg = audio_util.PcmSegment(x, 44100)
    assert seg.max == int(np.abs(x.astype(int)).max())
    assert seg.rms == int(np.sqrt(np.mean(x.astype(np.float64) ** 2)))
    out = audio_util.apply_filters(seg, compression=False)
    # normalised to 0.1 dB below full scale: peak = floor(32768 * 10^(-0.1/20)) up to the floor of the product
    assert abs(out.max - 32768 * 10 ** (-0.1 / 20)) <= 2
    assert out.channels == 2 and out.frame_rate == 44100
    # gain maths: +6.0206 dB doubles, floor toward -inf, clipping at the int16 rails
    g = audio_util.PcmSegment(np.array([[100], [-101], [20000], [-20000]], np.int16), 8000).apply_gain(20 * np.log10(2.0))
    assert g.get_array_of_samples().tolist() in ([200, -202, 32767, -32768], [199, -203, 32767, -32768], [200, -203, 32767, -32768])
    assert audio_util.PcmSegment(np.zeros((10, 1), np.int16), 8000).dBFS == float("-inf")
    pydub = audio_util._pydub()
    if pydub is not None:  # pin against the real thing whenever it is installed
        ref = pydub.AudioSegment(x.tobytes(), frame_rate=44100, sample_width=2, channels=2)
        ref = pydub.effects.normalize(ref.apply_gain(-12 - ref.dBFS), headroom=0.1)
        assert np.array_equal(np.array(ref.get_array_of_samples()), out.get_array_of_samples())


def test_header_is_plain_c_and_links_from_c(repo_root, tmp_path):
    """include/rfx.h is the whole boundary: it must compile as strict C99 (no C++ / torch types) and a C program
    must link against librfx.so and call the entry points that need no GPU."""
    import shutil
    import subprocess

    from riffusion import _hip

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include "rfx.h"\n'
        "int main(void) {\n"
        "  rfx_params p = {44100, 17640, 4410, 441, 512, 200};\n"
        "  rfx_plan* plan = 0;\n"
        "  int rc = rfx_plan_create(&p, 0, 0, 0, &plan); /* null window: must be refused, not crash */\n"
        '  printf("%d %d %d %d %s\\n", rfx_version() > 0, rfx_frame_stride(), rfx_num_bins(), rc, rfx_last_error());\n'
        "  return 0;\n}\n"
    )
    exe = tmp_path / "abi"
    lib_dir = os.path.dirname(_hip.library_path())
    subprocess.run(
        ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(repo_root, "include"), str(src),
         "-o", str(exe), "-L", lib_dir, "-lrfx", f"-Wl,-rpath,{lib_dir}"],
        check=True,
    )
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(maxsplit=4)
    assert out[:3] == ["1", "9408", "8821"]
    assert int(out[3]) < 0 and "rfx_plan_create" in out[4]


# ---- segment arithmetic pinned on CPython's audioop (the C module pydub.AudioSegment itself calls) ----------------
try:
    import audioop as _ao
except ImportError:  # Python >= 3.13
    _ao = None


@pytest.mark.skipif(_ao is None, reason="audioop removed from this interpreter")
def test_numpy_restatements_equal_audioop_bit_for_bit():
    """PcmSegment runs on audioop when it exists; its numpy fall-backs (interpreters without audioop) are pinned here."""
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((20000, 2)) * 9000).clip(-32768, 32767).astype(np.int16)
    x[:4] = [[32767, -32768], [-32768, 32767], [1, -1], [0, 0]]
    for fac in (0.0, 0.001, 0.25, 0.5, 0.7071, 1.0, 1.9999, 2.2387211385683394, 10.0, 1e6):
        want = np.frombuffer(_ao.mul(x.tobytes(), 2, fac), dtype=np.int16).reshape(x.shape)
        assert np.array_equal(audio_util.PcmSegment._mul_np(x, fac), want), fac
    assert np.array_equal(audio_util.PcmSegment._tomono_np(x), np.frombuffer(_ao.tomono(x.tobytes(), 2, 0.5, 0.5), dtype=np.int16))
    seg = audio_util.PcmSegment(x, 44100)
    assert seg.rms == _ao.rms(x.tobytes(), 2) == int(np.sqrt(np.sum(x.astype(np.float64) ** 2) / x.size))
    assert seg.max == _ao.max(x.tobytes(), 2) == 32768
    st = audio_util.PcmSegment(x[:, :1].copy(), 44100).set_channels(2)
    assert np.array_equal(st._data[:, 0], x[:, 0]) and np.array_equal(st._data[:, 1], x[:, 0])


@pytest.mark.skipif(_ao is None, reason="audioop removed from this interpreter")
def test_apply_filters_equals_pydub_call_sequence_on_audioop():
    """Reference audio_util.py:64-70 written out as the audioop calls pydub 0.25.1 makes for it (apply_gain = audioop.mul by
    10**(dB/20); dBFS = 20*log10(audioop.rms / 32768); normalize = gain by 20*log10(32768*10**(-0.1/20) / audioop.max))."""
    import math

    rng = np.random.default_rng(9)
    x = (rng.standard_normal((30000, 2)) * 2500).astype(np.int16)
    raw = x.tobytes()
    dbfs = 20 * math.log(_ao.rms(raw, 2) / 32768.0, 10)
    raw = _ao.mul(raw, 2, 10 ** ((-12 - dbfs) / 20))
    peak = _ao.max(raw, 2)
    raw = _ao.mul(raw, 2, 10 ** ((20 * math.log(32768.0 * (10 ** (-0.1 / 20)) / peak, 10)) / 20))
    got = audio_util.apply_filters(audio_util.PcmSegment(x, 44100), compression=False)
    assert np.array_equal(got._data, np.frombuffer(raw, dtype=np.int16).reshape(-1, 2))


def test_stitch_and_overlay_segments():
    """Reference audio_util.py:75-100 without pydub: append with crossfade, overlay."""
    rate = 8000
    a = audio_util.PcmSegment(np.full((rate, 1), 1000, np.int16), rate)  # 1 s
    b = audio_util.PcmSegment(np.full((rate // 2, 1), -3000, np.int16), rate)  # 0.5 s
    out = audio_util.stitch_segments([a, b], crossfade_s=0.25)
    n_x = rate // 4
    assert out.frame_count() == a.frame_count() + b.frame_count() - n_x and out.frame_rate == rate
    y = out.get_array_of_samples()
    assert np.all(y[: rate - n_x] == 1000) and np.all(y[rate:] == -3000)  # untouched outside the crossfade
    mid = y[rate - n_x : rate].astype(int)
    assert mid[0] in (1000, 999) and abs(mid[-1] + 3000) <= 30  # a fades out, b fades in (coarse 1 ms gain steps)
    assert np.all(np.diff(mid) <= 0)
    assert audio_util.stitch_segments([a], 0.1) is a
    with pytest.raises(ValueError):
        audio_util.stitch_segments([a, b], crossfade_s=0.75)  # longer than the appended segment, as pydub raises
    plain = a.append(b, crossfade=0)
    assert plain.frame_count() == a.frame_count() + b.frame_count()
    # overlay: saturating sum over the overlap, length of the first segment
    loud = audio_util.PcmSegment(np.full((rate, 1), 30000, np.int16), rate)
    ov = audio_util.overlay_segments([loud, loud, b])
    z = ov.get_array_of_samples()
    assert ov.frame_count() == rate and z[0] == 32767 - 3000 and z[-1] == 32767


def test_stitch_of_clips_that_are_not_whole_milliseconds():
    """pydub slices by MILLISECONDS on the length rounded to whole ms (AudioSegment.__getitem__ / _parse_position) and rebuilds
    a faded segment from one-millisecond pieces: a clip of 5110.4 ms loses the frames past ms 5110 of its head when it is
    sliced `[:-crossfade]`, and its crossfade tail `[-crossfade:]` starts at (5110 - crossfade) ms."""
    rate = 44100
    n = 441 * 511 + 17  # 5110.385 ms: len() == 5110
    a = audio_util.PcmSegment(np.arange(n, dtype=np.int64).astype(np.int16).reshape(-1, 1), rate)
    assert len(a) == 5110
    r = rate / 1000.0
    # seg[a:b] = frames [int(a_ms * r), int(b_ms * r)); None bounds mean 0 / len(self) in ms
    assert a._slice_ms(None, None).frame_count() == int(5110 * r) == n - 17
    assert a._slice_ms(-100, None).frame_count() == int(5110 * r) - int(5010 * r)
    assert a._slice_ms(None, -100).frame_count() == int(5010 * r)
    assert np.array_equal(a._slice_ms(7, 9)._data, a._data[int(7 * r) : int(9 * r)])
    # a length that rounds UP (x.6 ms): the slice to len(self) is padded with silence like pydub does (<= 2 ms)
    up = audio_util.PcmSegment(np.ones((int(10.6 * r), 1), np.int16), rate)
    assert len(up) == 11 and up._slice_ms(None, None).frame_count() == int(11 * r)
    assert up._slice_ms(None, None)._data[-1, 0] == 0 and up._slice_ms(None, None)._data[int(10.6 * r) - 1, 0] == 1
    # fade: rebuilt from 1-ms pieces + the (empty) remainder past len(self) ms
    faded = a._slice_ms(-200, None)._fade(to_gain=-120)
    tail = a._slice_ms(-200, None)
    assert faded.frame_count() == int(len(tail) * r)
    b = audio_util.PcmSegment(np.full((n, 1), 500, np.int16), rate)
    out = a.append(b, crossfade=200)
    head = int(4910 * r)
    xf = int(len(tail) * r)
    rest = int(5110 * r) - int(200 * r)
    assert out.frame_count() == head + xf + rest
    assert np.array_equal(out._data[:head], a._data[:head])
    assert np.all(out._data[head + xf :] == 500)


@pytest.mark.skipif(_ao is None, reason="audioop removed from this interpreter")
def test_set_frame_rate_is_audioop_ratecv():
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((4800, 2)) * 5000).astype(np.int16)
    seg = audio_util.PcmSegment(x, 48000).set_frame_rate(44100)
    want, _ = _ao.ratecv(x.tobytes(), 2, 2, 48000, 44100, None)
    assert seg.frame_rate == 44100 and np.array_equal(seg._data, np.frombuffer(want, dtype=np.int16).reshape(-1, 2))
    assert audio_util.PcmSegment(x, 44100).set_frame_rate(44100)._data is not None


def test_cli_batch_flags_mirror_the_reference():
    """reference cli.py:134-149: stereo tiles by default (--mono switches), --sample-rate, --image-extension jpg, --limit."""
    from riffusion import cli

    ns = cli.build_parser().parse_args(["audio-to-images-batch", "--audio-dir", "a", "--output-dir", "b"])
    assert ns.mono is False and ns.sample_rate == 44100 and ns.image_extension == "jpg" and ns.limit == -1
    assert ns.num_frequencies == 512 and ns.max_frequency == 10000 and ns.power_for_image == 0.25 and ns.step_size_ms == 10
    ns = cli.build_parser().parse_args(["images-to-audio-batch", "--image-dir", "a", "--output-dir", "b", "--no-filters"])
    assert ns.no_filters is True and ns.batch_size == 64
    # file lists split over the ranks of a one-process-per-GPU launch
    old = {k: os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        os.environ.update(WORLD_SIZE="4", RANK="1", LOCAL_RANK="1")
        assert list(cli._rank_slice(list(range(10)))) == [3, 4, 5] and cli._rank_device("cuda") == "cuda:1"
        assert cli._rank_device("cuda:3") == "cuda:3"
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    assert list(cli._rank_slice([1, 2, 3])) == [1, 2, 3]


def test_quantize_pipeline_images_matches_numpy_to_pil():
    """riffusion_pipeline.py:433-434 -> diffusers numpy_to_pil: (images * 255).round().astype('uint8')."""
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter as C

    rng = np.random.default_rng(0)
    x = rng.random((3, 16, 8, 3), dtype=np.float32)
    x.reshape(-1)[:6] = [0.0, 1.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255]
    want = (x * 255).round().astype("uint8")
    got = C.quantize_pipeline_images(torch.from_numpy(x))
    assert got.dtype == torch.uint8 and np.array_equal(got.numpy(), want)
    with pytest.raises(ValueError):
        C.quantize_pipeline_images(torch.zeros(1, 3, 8, 8))  # NCHW: not what the pipeline hands over


def test_bench_gpus_flag_relaunches_n_ranks(repo_root, monkeypatch):
    """bench.py --gpus N outside a distributed launch starts N ranks under torch.distributed.run on 127.0.0.1."""
    import importlib
    import subprocess

    monkeypatch.syspath_prepend(repo_root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.relaunch_distributed(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # and main() takes that route only when no RANK is set
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(bench, "relaunch_distributed", lambda n: 17)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 17


def test_plan_options_are_validated_before_any_device_call(repo_root):
    """rfx_plan_options as the Python layer fills it: unknown gl_form / frame_engine names are refused on the host, and the
    structure keeps the layout include/rfx.h declares (struct_size first, int32 fields after it)."""
    import ctypes

    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    assert [f[0] for f in _hip.RfxPlanOptions._fields_] == ["struct_size", "gl_form", "gl_frames_per_slot", "frame_engine", "plan_layout",
                                                                "imel_form"]
    assert ctypes.sizeof(_hip.RfxPlanOptions) == 24
    assert _hip.FRAME_ENGINES == {"auto": 0, "generic": 1} and _hip.GL_FORMS == {"auto": 0, "runs": 1, "frames": 2}
    assert _hip.PLAN_LAYOUTS == {"auto": 0, "generic": 1} and _hip.IMEL_FORMS == {"auto": 0, "groups": 1}
    header = open(os.path.join(repo_root, "include", "rfx.h")).read()
    assert "RFX_ENGINE_AUTO = 0" in header and "RFX_ENGINE_GENERIC = 1" in header and "int32_t frame_engine;" in header
    assert "RFX_LAYOUT_AUTO = 0" in header and "RFX_LAYOUT_GENERIC = 1" in header and "int32_t plan_layout;" in header
    assert "RFX_IMEL_FORM_AUTO = 0" in header and "RFX_IMEL_FORM_GROUPS = 1" in header and "int32_t imel_form;" in header
    # the release library reads no environment variable: every getenv of the C++ side sits behind RFX_ABLATION / RFX_TIMING
    api = open(os.path.join(repo_root, "riffusion-hobby_amd", "csrc", "rfx_api.hip")).read()
    assert api.count("getenv(") == 2 and "#ifdef RFX_ABLATION\n  return getenv(name);" in api and 'getenv("RFX_TIMING_PTR")' in api
    for kw in ({"gl_form": "sometimes"}, {"frame_engine": "fastest"}, {"plan_layout": "dense"}, {"imel_form": "fast"}):
        with pytest.raises(ValueError):
            _hip.Plan(SpectrogramParams(sample_rate=48000), "cpu", **kw)


def test_c_abi_refuses_bad_plan_options_before_any_device_call(repo_root):
    """rfx_plan_create_ex validates rfx_plan_options first (no GPU needed to get the refusal): an imel_form / plan_layout outside
    the enums and a struct_size larger than the library's own struct are RFX_ERR_INVALID with a message that names the field."""
    import ctypes

    from riffusion import _hip

    lib = _hip.load_library()
    win = (ctypes.c_float * 4410)()
    cp = _hip.RfxParams(44100, 17640, 4410, 441, 512, 200)
    handle = ctypes.c_void_p()
    size = ctypes.sizeof(_hip.RfxPlanOptions)
    for opt, word in ((_hip.RfxPlanOptions(size, 0, 0, 0, 0, 7), b"imel_form"), (_hip.RfxPlanOptions(size, 0, 0, 0, 5, 0), b"plan_layout"),
                      (_hip.RfxPlanOptions(size + 4, 0, 0, 0, 0, 0), b"struct_size"), (_hip.RfxPlanOptions(4, 0, 0, 0, 0, 0), b"struct_size")):
        rc = lib.rfx_plan_create_ex(ctypes.byref(cp), ctypes.cast(win, ctypes.c_void_p), None, 0, ctypes.byref(opt), ctypes.byref(handle))
        assert rc != 0 and not handle.value and word in lib.rfx_last_error(), (rc, lib.rfx_last_error())


def test_c_abi_image_from_waveform_refuses_null_arguments_before_any_device_call():
    """rfx_image_from_waveform (spectrogram_image_converter.py:30-51 in one call) checks its arguments first: a null plan / buffer is
    RFX_ERR_INVALID with a message naming the entry point, and the workspace query answers 0 for a null plan - no GPU needed."""
    import ctypes

    from riffusion import _hip

    lib = _hip.load_library()
    assert lib.rfx_image_from_waveform_workspace_bytes(None, 4, 0, 44100) == 0
    buf = (ctypes.c_float * 16)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.rfx_image_from_waveform(None, ptr, 1, 0, 44100, ptr, ptr, ptr, ptr, 1 << 20, None)
    assert rc != 0 and b"rfx_image_from_waveform" in lib.rfx_last_error(), (rc, lib.rfx_last_error())


def test_c_abi_waveform_from_mel_refuses_null_arguments_before_any_device_call():
    """rfx_waveform_from_mel (spectrogram_converter.py:187-204 in one call): a null plan is RFX_ERR_INVALID, the workspace query 0."""
    import ctypes

    from riffusion import _hip

    lib = _hip.load_library()
    assert lib.rfx_waveform_from_mel_workspace_bytes(None, 2, 64) == 0
    buf = (ctypes.c_float * 16)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.rfx_waveform_from_mel(None, ptr, 1, 64, 1, 0, 4, ctypes.c_float(0.99), ptr, ptr, 1 << 20, None)
    assert rc != 0 and b"rfx_waveform_from_mel" in lib.rfx_last_error(), (rc, lib.rfx_last_error())


def test_c_abi_audio_from_image_refuses_null_arguments_before_any_device_call():
    """rfx_audio_from_image_u8 (spectrogram_image_converter.py:54-91 in one call): a null plan is RFX_ERR_INVALID, the workspace query 0."""
    import ctypes

    from riffusion import _hip

    lib = _hip.load_library()
    assert lib.rfx_audio_from_image_workspace_bytes(None, 2, 0, 64) == 0
    buf = (ctypes.c_float * 16)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.rfx_audio_from_image_u8(None, ptr, 1, 64, 0, ptr, 0, 4, ctypes.c_float(0.99), 1, ptr, ptr, ptr, 1 << 20, None)
    assert rc != 0 and b"rfx_audio_from_image_u8" in lib.rfx_last_error(), (rc, lib.rfx_last_error())
