"""
Round 5: the Griffin-Lim run partition.  rfx_griffinlim cuts the batch's B x T frames, counted clip after clip, into equal runs -
one per resident workgroup slot, never more - so a run may cross a clip boundary (csrc/rfx_gl.hip: one segment per clip).  Until
round 5 every clip had ceil(slots / B) runs of its own and B = 65 or 100 launched more workgroups than the chip holds.
Everything goes through the C ABI; the oracle is the checker.
"""
import numpy as np
import pytest
import torch

from helpers import snr_db

pytestmark = pytest.mark.gpu

T = 512


@pytest.fixture(scope="module")
def plan():
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    return _hip.get_plan(SpectrogramParams(), "cuda")


@pytest.fixture(scope="module")
def O():
    import riffusion_oracle

    return riffusion_oracle


@pytest.mark.parametrize("B", [65, 100])
def test_batches_that_do_not_divide_the_slots_equal_the_64_tile_partition(plan, O, B):
    """cli.py:172-204 hands over whatever number of files a directory holds.  B = 65 / 100 at the headline tile size: every clip
    must come out as it does inside a batch of 64 (the partition rounds 1-4 measured and tested: 64-frame runs that never cross
    a clip) up to the summation order at the run seams, and one clip of the odd-sized batch is checked against the oracle."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(500 + B)
    S = torch.rand(B * T, plan.frame_stride, device=dev, generator=g) * 1000.0
    mag = plan.unpack_magnitudes(S, B, T)          # what the slots mean as (B, n_stft, T) ...
    S = plan.pack_magnitudes(mag)                  # ... with duplicate slots consistent
    a0_bft = torch.view_as_complex(torch.rand(B, plan.n_stft, T, 2, device=dev, generator=g))
    A = plan.pack_complex(a0_bft)
    assert plan.lib.rfx_griffinlim_form(plan.handle, B, T) == 1  # the run-based kernel
    whole = plan.griffinlim(S, B, T, 3, 0.99, angles0_slots=A)
    assert whole.shape == (B, 441 * (T - 1)) and bool(torch.isfinite(whole).all())
    snrs = []
    for lo in range(0, B, 64):
        hi = min(B, lo + 64)
        n = hi - lo
        if n < 64:  # pad the last chunk to 64 clips so that it, too, takes the 64-frame runs of the old partition
            idx = torch.arange(lo, lo + 64, device=dev) % B
        else:
            idx = torch.arange(lo, hi, device=dev)
        rows = (idx[:, None] * T + torch.arange(T, device=dev)[None, :]).reshape(-1)
        part = plan.griffinlim(S[rows].contiguous(), 64, T, 3, 0.99, angles0_slots=A[rows].contiguous())
        for i in range(n):
            snrs.append(snr_db(part[i], whole[lo + i]))
    print(f"B = {B}: every clip vs the same clip inside a 64-tile batch after 3 iterations: median {np.median(snrs):.1f} dB, worst {min(snrs):.1f} dB")
    # measured: median 112, worst of 100 clips 103.4 dB (rounding at other seams, three iterations of growth); a wrong seam gives < 30 dB
    assert float(np.median(snrs)) >= 105.0 and min(snrs) >= 95.0
    b = B - 1  # the last clip: its runs start in the clip before it
    want = O.griffinlim(mag[b : b + 1].cpu(), O.OracleParams(), angles0=a0_bft[b : b + 1].cpu(), n_iter=3)
    s = snr_db(want, whole[b : b + 1].cpu())
    print(f"B = {B}: clip {b} vs the oracle after 3 iterations: {s:.1f} dB")
    assert s >= 95.0


@pytest.mark.parametrize("B,Tn", [(100, 60), (37, 150), (9, 57), (523, 22)])
def test_runs_that_cross_clip_boundaries_against_the_oracle(plan, O, B, Tn):
    """Short clips, many of them: runs of ~12 frames cut through clips of 22 - 150 frames, most runs hold parts of two clips and
    some clips are shared by three runs.  All clips against the oracle (injected initial angles, 3 iterations)."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    pl = _hip.get_plan(SpectrogramParams(), "cuda", gl_form="runs")
    op = O.OracleParams()
    g = torch.Generator().manual_seed(31 * B + Tn)
    mag = torch.rand(B, op.n_stft, Tn, generator=g) * 1000.0
    a0 = torch.rand(B, op.n_stft, Tn, dtype=torch.complex64, generator=g)
    want = O.griffinlim(mag, op, angles0=a0, n_iter=3)
    got = pl.griffinlim(pl.pack_magnitudes(mag.cuda()), B, Tn, 3, 0.99, angles0_slots=pl.pack_complex(a0.cuda())).cpu()
    per_clip = [snr_db(want[i], got[i]) for i in range(B)]
    print(f"B = {B}, T = {Tn}: Griffin-Lim 3 vs oracle: all clips {snr_db(want, got):.1f} dB, worst clip {min(per_clip):.1f} dB")
    # a seam handled wrongly leaves a clip at 10 - 30 dB; one ill-conditioned bin (tests/helpers.py::mask_ill_conditioned_bins) can
    # cost a 22-frame clip up to 10 log10(F T / 12) = 42 dB, so the per-clip floor is 45 dB and the tight gates are the batch and the median
    assert got.shape == want.shape and snr_db(want, got) >= 95.0 and float(np.median(per_clip)) >= 100.0 and min(per_clip) >= 45.0


@pytest.mark.parametrize("kw,unit", [(dict(min_frequency=20, max_frequency=20000), 1), (dict(max_frequency=22050), 1), (dict(max_frequency=16000, mel_scale_norm="slaney"), 0),
                                     (dict(max_frequency=20000, mel_scale_type="slaney"), 1), (dict(num_frequencies=384), 1),
                                     # 256 filters: every group is a long one, and group 0 holds the first filter's rising edge next to its
                                     # falling one - not a line: it sits in its thread's table-form slot (and the unit form is off)
                                     (dict(num_frequencies=256), 0), (dict(num_frequencies=300, max_frequency=12000), 1),
                                     # 200 filters: 56 thread roles own nothing; 48 kHz: the generic plan's plain bin-ordered frames
                                     (dict(num_frequencies=200), 0), (dict(sample_rate=48000, max_frequency=20000), 1)])
def test_line_form_group_kernel_on_banks_with_long_groups(O, kw, unit):
    """Banks whose groups are too long for the group kernels' register budgets (512 filters up to 16 / 20 / 22.05 kHz - the reference's
    own round-trip test runs 20 Hz .. 20 kHz, test/spectrogram_converter_test.py:46-53 - or 384 filters) ran on the general LDS kernel
    until round 5 (169 ms per 64 tiles); they take the line-form group kernel now (rfx_plan_imel_kernel 5).  Against the oracle after
    the reference's 200 steps, injected start, stereo coupling and a frame count that fills no workgroup slot evenly; the same bank on
    the general kernel (rfx_plan_options.imel_form has no switch for it: the uniform variant of an ablation build would be needed) is
    covered by test_inverse_mel_other_parameter_sets_use_fallback_kernels' history: here the oracle is the judge."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(**kw)
    op = O.params_from(p)
    plan = _hip.get_plan(p, "cuda")
    assert plan.lib.rfx_plan_imel_kernel(plan.handle) == 5 and plan.lib.rfx_plan_imel_unit_form(plan.handle) == unit
    T, C = 9, 2
    g = torch.Generator().manual_seed(len(str(kw)))
    mel = torch.rand(C, p.num_frequencies, T, generator=g) ** 3 * 2e7
    spec0 = torch.rand(C, T, op.n_stft, generator=g)
    ref = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
    got = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), C, spec0=spec0.cuda()), C, T).cpu()
    active = O.mel_filterbank(op).abs().sum(1) > 0
    rel = float(torch.linalg.norm(got[:, active] - ref[:, active]) / torch.linalg.norm(ref[:, active]))
    print(f"{kw}: line-form group kernel vs oracle after 200 steps: rel-L2 {rel:.2e} on {int(active.sum())} active bins")
    assert rel <= 1e-5  # measured ~1e-7; the family-wide gate is 1e-3
    assert torch.equal(got[:, ~active], spec0.transpose(1, 2)[:, ~active])
    # drawn start: finite, non-negative, and a different seed gives different magnitudes
    a = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), C, seed=1), C, T)
    b = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), C, seed=2), C, T)
    assert bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0 and not torch.equal(a, b)


# ---- round 5: spectrogram_image_from_audio in one call (rfx_image_from_waveform) and the forward kernel's packed tables
FWD_CASES = [
    # (SpectrogramParams keywords, stereo, clips, frames): engines and banks the forward path knows
    ({}, False, 3, 512),                                                   # 44.1 kHz, default bank: product form, packed tables
    ({}, True, 2, 130),                                                    # stereo: the maximum spans an image's two channels
    ({}, False, 2, 101),                                                   # T % 4 != 0: the byte-wise tail of the encoder
    ({"max_frequency": 20000, "min_frequency": 20}, False, 2, 64),         # a bank over every kb (plain tables)
    ({"num_frequencies": 700, "max_frequency": 10000}, False, 1, 40),      # wider than the product form takes: table form, maximum by a pass
    ({"sample_rate": 48000, "max_frequency": 10000}, True, 2, 72),         # row family
    ({"sample_rate": 11025, "max_frequency": 5000}, False, 2, 50),         # generic engine
]


@pytest.mark.parametrize("kw,stereo,N,Tn", FWD_CASES)
def test_image_from_waveform_equals_the_two_calls_byte_for_byte(kw, stereo, N, Tn):
    """spectrogram_image_converter.py:30-51.  rfx_image_from_waveform leaves out the (N*C, n_mels, T) tensor and the pass for its
    maximum; what comes out must be what rfx_mel_from_waveform + rfx_image_encode_u8 give: same bytes, same MAX_VALUE bits."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    p = SpectrogramParams(stereo=stereo, **kw)
    plan = _hip.get_plan(p, "cuda")
    C = 2 if stereo else 1
    g = torch.Generator(device="cuda").manual_seed(77 + Tn)
    wave = torch.randn(N * C, p.hop_length * (Tn - 1), device="cuda", generator=g) * 6000.0
    wave[0, : wave.shape[1] // 3] *= 1e-3  # a quiet stretch: small ratios, the far end of the threshold table
    thr = torch.from_numpy(image_util.encode_thresholds(float(p.power_for_image))).cuda()
    mel = plan.mel_from_waveform(wave)
    img2, mx2 = plan.image_encode(mel, stereo, thr)
    img1, mx1 = plan.image_from_waveform(wave, stereo, thr)
    assert img1.shape == img2.shape == (N, p.num_frequencies, Tn, 3)
    assert torch.equal(mx1.view(torch.int32), mx2.view(torch.int32)), (mx1, mx2)
    assert torch.equal(img1, img2), f"{int((img1 != img2).sum())} bytes differ"
    assert int(img1.max()) > 0 and len(torch.unique(img1)) > 50  # (a real picture, not a constant)


def test_forward_kernel_with_packed_tables_against_the_dense_definition(plan):
    """The default-bank forward kernel reads its product positions as packed 16-bit LDS addresses, derives nine of its twenty
    twiddles and keeps a sliding input window (csrc/rfx_stft.hip, round 5): |STFT| through torch.stft times the dense filterbank
    is the definition (spectrogram_converter.py:165-185); clips of different lengths exercise runs of 1 .. 64 frames (a clip needs more than n_fft / 2 samples: 22 frames)."""
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams()
    win = torch.hann_window(p.win_length, device="cuda")
    for B, Tn in ((1, 23), (2, 77), (5, 512)):
        g = torch.Generator(device="cuda").manual_seed(B * 1000 + Tn)
        wave = torch.randn(B, p.hop_length * (Tn - 1), device="cuda", generator=g) * 8000.0
        mel = plan.mel_from_waveform(wave)
        ref = torch.stft(wave.double(), p.n_fft, p.hop_length, p.win_length, win.double(), center=True, pad_mode="reflect", return_complex=True).abs()
        ref_mel = (ref.transpose(1, 2) @ plan.melfb.cuda().double()).transpose(1, 2)
        rel = float(torch.linalg.norm(mel.double() - ref_mel) / torch.linalg.norm(ref_mel))
        worst = float(((mel.double() - ref_mel).abs() / (ref_mel.abs() + 1e-3 * ref_mel.abs().max())).max())
        print(f"forward kernel, B = {B}, T = {Tn}: rel-L2 {rel:.2e} against the float64 dense definition, worst relative entry {worst:.2e}")
        assert rel < 5e-7 and worst < 1e-4


@pytest.mark.parametrize("kw,B,cpc,Tn", [({}, 4, 2, 96), ({}, 1, 1, 24), ({"sample_rate": 48000, "max_frequency": 10000}, 2, 1, 40),
                                         ({"sample_rate": 11025, "max_frequency": 5000}, 2, 2, 30)])
def test_waveform_from_mel_equals_the_two_calls_bit_for_bit(kw, B, cpc, Tn):
    """spectrogram_converter.py:187-204.  rfx_waveform_from_mel keeps the linear magnitudes inside its workspace; what comes out must be
    rfx_inverse_mel (seed) followed by rfx_griffinlim (seed + 1), to the bit, on every engine."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(num_griffin_lim_iters=5, **kw)
    plan = _hip.get_plan(p, "cuda")
    g = torch.Generator(device="cuda").manual_seed(9 + Tn)
    mel = torch.rand(B, p.num_frequencies, Tn, device="cuda", generator=g) ** 3 * 2e6
    lin = plan.inverse_mel(mel, cpc, seed=1234)
    two = plan.griffinlim(lin, B, Tn, 5, 0.99, seed=1235)
    one = plan.waveform_from_mel(mel, cpc, 5, 0.99, seed=1234)
    assert one.shape == two.shape == (B, p.hop_length * (Tn - 1)) and bool(torch.isfinite(one).all())
    assert torch.equal(one.view(torch.int32), two.view(torch.int32))
    assert float(one.abs().max()) > 0


@pytest.mark.parametrize("kw,stereo,N,Tn", [({}, False, 3, 64), ({}, True, 2, 40), ({"sample_rate": 48000, "max_frequency": 10000}, True, 1, 30)])
def test_audio_from_image_equals_the_three_calls_byte_for_byte(kw, stereo, N, Tn):
    """spectrogram_image_converter.py:54-91.  rfx_audio_from_image_u8 keeps the mel amplitudes, the linear magnitudes and the float
    waveform inside its workspace; the PCM must be what rfx_image_decode_u8 + rfx_inverse_mel + rfx_griffinlim + rfx_pcm16 give."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    p = SpectrogramParams(stereo=stereo, num_griffin_lim_iters=4, **kw)
    plan = _hip.get_plan(p, "cuda")
    C = 2 if stereo else 1
    g = torch.Generator(device="cuda").manual_seed(5 + Tn)
    img = torch.randint(0, 256, (N, p.num_frequencies, Tn, 3), dtype=torch.uint8, device="cuda", generator=g)
    lut = torch.from_numpy(image_util.decode_lut(float(p.power_for_image), 30e6)).cuda()
    mel = plan.image_decode(img, stereo, lut)
    lin = plan.inverse_mel(mel, C, seed=77)
    wave = plan.griffinlim(lin, N * C, Tn, 4, 0.99, seed=78)
    pcm3, peak3 = plan.pcm16(wave, channels=C, normalize=True)
    pcm1, peak1 = plan.audio_from_image(img, stereo, lut, 4, 0.99, seed=77)
    assert pcm1.shape == pcm3.shape == (N, p.hop_length * (Tn - 1), C) and pcm1.dtype == torch.int16
    assert torch.equal(peak1.view(torch.int32), peak3.view(torch.int32))
    assert torch.equal(pcm1, pcm3)
    assert int(pcm1.abs().max()) > 1000
