"""
Round 3 parity: the gaps that do not need torchaudio.

* one oracle tile taken from INSIDE the headline batch (B = 64, T = 512, same injected initial values): one hop from the
  oracle to the batch the bench times;
* `mel_scale_type="slaney"` / `mel_scale_norm="slaney"` (spectrogram_params.py:34-35, passed at spectrogram_converter.py:82-83
  and :97-98): forward <= 1e-4, InverseMelScale <= 1e-3, and which SGD kernel each bank selects;
* the realistic set of SURVEY 8(d): the reference's other seed images and its two test PNGs through the inverse path with
  injected initial values, its other two test clips through the forward path;
* og_beat.png with production RNG on both sides: MEAN spectral convergence over 8 seeds per side within 1 %.

Everything goes through the C ABI (librfx.so); the oracle is the checker.
"""
import os

import numpy as np
import pytest
import torch
from PIL import Image
from scipy.io import wavfile

from helpers import snr_db, synthetic_tiles_u8, synthetic_wave

pytestmark = pytest.mark.gpu

T_FULL = 512


@pytest.fixture(scope="module")
def O():
    import riffusion_oracle

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return riffusion_oracle


def _plan(params, **kw):
    from riffusion import _hip

    return _hip.get_plan(params, "cuda", **kw)


def _active_rows(O, op):
    return O.mel_filterbank(op).abs().sum(1) > 0


def test_oracle_tile_from_inside_the_headline_batch(O):
    """configs[1] as the bench runs it (64 tiles, 512 frames, SGD-200, Griffin-Lim 32 in ONE batch, run-based kernel): tiles 0, 37
    and 63 of the batch - first, middle, last: 3 / 64 of the headline output - carry initial values drawn on the host, the oracle
    gets the same ones (13 s of oracle per tile; one tile until round 5).  The other 61 tiles are checked against the device
    itself (tests/test_gpu_full_size.py: every clip equals the clip converted alone; since round 6 bit for bit,
    tests/test_gpu_round6.py)."""
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    params = SpectrogramParams()
    op = O.params_from(params)
    plan = _plan(params)
    B, checked = 64, (0, 37, 63)
    dev = torch.device("cuda")
    tiles_np = synthetic_tiles_u8(B)
    tiles = torch.from_numpy(tiles_np).to(dev)
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)
    g = torch.Generator().manual_seed(1234)
    gd = torch.Generator(device=dev).manual_seed(99)
    spec0 = torch.rand(B, T_FULL, op.n_stft, device=dev, generator=gd)
    angles0 = torch.view_as_complex(torch.rand(B, op.n_stft, T_FULL, 2, device=dev, generator=gd))
    host_init = {}
    for b in checked:
        spec0_b = torch.rand(1, T_FULL, op.n_stft, generator=g)
        angles0_b = torch.rand(1, op.n_stft, T_FULL, dtype=torch.complex64, generator=g)
        spec0[b] = spec0_b[0].to(dev)
        angles0[b] = angles0_b[0].to(dev)
        host_init[b] = (spec0_b, angles0_b)
    a0_slots = plan.pack_complex(angles0)
    del angles0

    assert plan.lib.rfx_griffinlim_form(plan.handle, B, T_FULL) == 1  # the run-based kernel of the headline
    mel = plan.image_decode(tiles, False, lut)
    lin = plan.inverse_mel(mel, 1, spec0=spec0)
    del spec0
    act = _active_rows(O, op)
    lin_sub = lin.clone()
    wants = {}
    for b in checked:
        spec0_b, angles0_b = host_init[b]
        mel_b = torch.from_numpy(O.spectrogram_from_image_u8(tiles_np[b], 0.25, False, 30e6))
        assert torch.equal(mel[b : b + 1].cpu(), mel_b)
        want_lin = O.inverse_mel_scale_sgd(mel_b, op, spec0=spec0_b)
        got_lin = plan.unpack_magnitudes(lin[b * T_FULL : (b + 1) * T_FULL].contiguous(), 1, T_FULL).cpu()
        rel = float(torch.linalg.norm(got_lin[:, act] - want_lin[:, act]) / torch.linalg.norm(want_lin[:, act]))
        assert rel <= 1e-3 and torch.equal(got_lin[:, ~act], want_lin[:, ~act])
        # Griffin-Lim 32 of the whole batch; the tile's magnitudes replaced by the oracle's so that only the iteration is compared
        wants[b] = (O.griffinlim(want_lin, op, angles0=angles0_b, n_iter=32), rel)
        lin_sub[b * T_FULL : (b + 1) * T_FULL] = plan.pack_magnitudes(want_lin.to(dev))
    wave = plan.griffinlim(lin_sub, B, T_FULL, 32, 0.99, angles0_slots=a0_slots)
    # and the batch exactly as the bench runs it: device SGD result -> device Griffin-Lim
    wave_full = plan.griffinlim(lin, B, T_FULL, 32, 0.99, angles0_slots=a0_slots)
    for b in checked:
        want, rel = wants[b]
        s_gl = snr_db(want, wave[b : b + 1].cpu())
        s_full = snr_db(want, wave_full[b : b + 1].cpu())
        print(f"tile {b} inside the B = 64 batch vs oracle: InverseMelScale rel-L2 {rel:.2e}; Griffin-Lim 32 {s_gl:.1f} dB on identical "
              f"magnitudes, {s_full:.1f} dB for device SGD -> device Griffin-Lim")
        # SURVEY 8(d)'s floor is 60 dB; measured 89.3 - 92.5 dB (identical magnitudes) and 86.5 - 97.2 dB (device SGD feeding the device
        # Griffin-Lim: its 1e-7-level differences grow through 32 chaotic iterations) over rounds 3 - 5 and two run partitions.  The
        # gates sit ~15 dB under the lowest figure seen, not at the floor (they were 60 / 40 until round 5)
        assert s_gl >= 75.0
        assert s_full >= 70.0


@pytest.mark.parametrize("scale,norm,kernel", [("htk", "slaney", 4), ("slaney", None, None), ("slaney", "slaney", None)])
def test_slaney_scale_and_norm(O, scale, norm, kernel):
    """The two mel parameters the reference exposes besides the defaults.  The normalised HTK bank has the default bank's
    sparsity pattern and takes the wave kernel with both weight lines in every chunk (its two weights per bin do not sum to one); the slaney SCALE moves the filter edges (its largest group may exceed
    the default budgets) and takes whichever kernel rfx_plan_imel_kernel reports."""
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams(mel_scale_type=scale, mel_scale_norm=norm, max_mel_iters=60, num_griffin_lim_iters=4)
    op = O.params_from(p)
    plan = _plan(p)
    assert torch.equal(plan.melfb, O.mel_filterbank(op))
    which = plan.lib.rfx_plan_imel_kernel(plan.handle)
    assert which >= 0 and (kernel is None or which == kernel)
    # the gradient's unit form needs w0 + w1 == 1 per bin: true of the triangles, not of area-normalised ones
    unit = plan.lib.rfx_plan_imel_unit_form(plan.handle)
    assert unit == (1 if norm is None and which >= 2 else 0)
    T = 48
    wave = synthetic_wave(2, 441 * (T - 1), seed=5)
    mel_ref = O.mel_amplitudes_from_waveform(wave, op)
    mel = plan.mel_from_waveform(wave.cuda()).cpu()
    assert (mel - mel_ref).abs().max() <= 1e-4 * mel_ref.max()
    rel_fwd = float(torch.linalg.norm(mel - mel_ref) / torch.linalg.norm(mel_ref))
    assert rel_fwd <= 1e-4
    # standalone MelScale member (MFMA GEMM over the dense bank)
    lin_in = O.stft_complex(wave, op).abs()
    ms = plan.mel_scale(lin_in.cuda()).cpu()
    assert torch.linalg.norm(ms - O.mel_scale(lin_in, O.mel_filterbank(op))) / torch.linalg.norm(mel_ref) <= 1e-4
    g = torch.Generator().manual_seed(3)
    spec0 = torch.rand(2, T, op.n_stft, generator=g)
    want = O.inverse_mel_scale_sgd(mel_ref, op, spec0=spec0)
    got = plan.unpack_magnitudes(plan.inverse_mel(mel_ref.cuda(), 2, spec0=spec0.cuda()), 2, T).cpu()
    act = _active_rows(O, op)
    rel = float(torch.linalg.norm(got[:, act] - want[:, act]) / torch.linalg.norm(want[:, act]))
    print(f"mel_scale_type={scale!r} mel_scale_norm={norm!r}: forward rel-L2 {rel_fwd:.2e}, InverseMelScale-60 rel-L2 {rel:.2e}, "
          f"SGD kernel {which} (4 = wave, 2 / 3 = per-wave groups, 1 = uniform groups, 0 = general), unit-form gradient {unit}")
    assert rel <= 1e-3 and torch.equal(got[:, ~act], want[:, ~act])


SEED_IMAGES = ["agile.png", "marim.png", "motorway.png", "vibes.png",
               "clip_2_start_103694_ms_duration_5678_ms.png", "clip_2_start_103694_ms_duration_5678_ms_stereo.png"]


@pytest.mark.parametrize("name", SEED_IMAGES)
def test_reference_images_through_the_inverse_path(O, golden_dir, name):
    """SURVEY 8(d) realistic set: real spectrogram images (sparse, structured spectra instead of white noise) -> InverseMelScale
    200 -> Griffin-Lim 32 with injected initial values, device vs oracle stage by stage; the stereo test PNG couples its two
    channels in the SGD loss mean (C = 2), the 568-frame test PNGs exercise T != 512."""
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    stereo = name.endswith("_stereo.png")
    with Image.open(os.path.join(golden_dir, name)) as im:
        rgb = np.asarray(image_util.rgb_array_from_image(im))
    params = SpectrogramParams(stereo=stereo)
    op = O.params_from(params)
    plan = _plan(params)
    C, T = (2 if stereo else 1), rgb.shape[1]
    mel = torch.from_numpy(O.spectrogram_from_image_u8(rgb, 0.25, stereo, 30e6))
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).cuda()
    assert torch.equal(plan.image_decode(torch.from_numpy(rgb)[None].cuda(), stereo, lut).cpu(), mel)
    g = torch.Generator().manual_seed(len(name))
    spec0 = torch.rand(C, T, op.n_stft, generator=g)
    angles0 = torch.rand(C, op.n_stft, T, dtype=torch.complex64, generator=g)
    want_lin = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
    got_lin = plan.unpack_magnitudes(plan.inverse_mel(mel.cuda(), C, spec0=spec0.cuda()), C, T).cpu()
    act = _active_rows(O, op)
    rel = float(torch.linalg.norm(got_lin[:, act] - want_lin[:, act]) / torch.linalg.norm(want_lin[:, act]))
    want = O.griffinlim(want_lin, op, angles0=angles0, n_iter=32)
    out = {}
    for form in ("frames", "runs"):
        pl = _plan(params, gl_form=form)
        got = pl.griffinlim(pl.pack_magnitudes(want_lin.cuda()), C, T, 32, 0.99, angles0_slots=pl.pack_complex(angles0.cuda())).cpu()
        assert got.shape == want.shape == (C, 441 * (T - 1))
        out[form] = snr_db(want, got)
    print(f"{name}: T = {T}, C = {C}: InverseMelScale rel-L2 {rel:.2e}; Griffin-Lim 32 vs oracle {out['frames']:.1f} dB (per-frame form) / "
          f"{out['runs']:.1f} dB (run-based form)")
    assert rel <= 1e-3 and torch.equal(got_lin[:, ~act], want_lin[:, ~act])
    # >= 60 dB after 32 iterations.  The iteration is chaotic and how fast rounding noise grows depends on the
    # spectrogram (the stereo test PNG sits at 60.6 dB, the seed images at 79-91 dB): within 5 dB of the floor the figure that
    # matters is the oracle's OWN fp32-vs-fp64 distance on this input - the device must not be further from the fp32 oracle
    # than the fp32 oracle is from exact arithmetic (minus 3 dB).  Round 5: no fixed escape below that - the floor is
    # min(60, own - 3), and an input on which it falls under 60 says so (own < 63 is printed and asserted with it).
    floor = 60.0
    if min(out.values()) < 65.0:
        want64 = O.griffinlim(want_lin, op, angles0=angles0, n_iter=32, dtype=torch.float64)
        own = snr_db(want64, want)
        floor = min(60.0, own - 3.0)
        print(f"{name}: fp32 oracle vs fp64 oracle {own:.1f} dB -> floor {floor:.1f} dB")
        assert floor == 60.0 or own < 63.0
    assert min(out.values()) >= floor


@pytest.mark.parametrize("name", ["clip_0_start_15795_ms_duration_5678_ms.wav", "clip_1_start_860_ms_duration_5678_ms.wav"])
def test_reference_clips_through_the_forward_path(O, golden_dir, name):
    """The reference's other two test clips (stereo int16, no rescaling, spectrogram_converter.py:117-121): mel amplitudes and the
    uint8 image against the oracle."""
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    sr, w = wavfile.read(os.path.join(golden_dir, name))
    assert sr == 44100 and w.ndim == 2 and w.shape[1] == 2
    wave = torch.from_numpy(np.ascontiguousarray(w.T).astype(np.float32))
    params = SpectrogramParams(stereo=True)
    op = O.params_from(params)
    plan = _plan(params)
    ref = O.mel_amplitudes_from_waveform(wave, op)
    got = plan.mel_from_waveform(wave.cuda())
    assert got.shape == ref.shape == (2, 512, 1 + wave.shape[1] // 441)
    d = (got.cpu() - ref).abs().max() / ref.max()
    rel = torch.linalg.norm(got.cpu() - ref) / torch.linalg.norm(ref)
    thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
    img, mx = plan.image_encode(got, True, thr)
    ref_img = O.image_u8_from_spectrogram(ref.numpy(), 0.25)
    diff = np.abs(img[0].cpu().numpy().astype(int) - ref_img.astype(int))
    print(f"{name}: max|d|/max {float(d):.2e}, rel-L2 {float(rel):.2e}, image pixels equal {float((diff == 0).mean()):.5f}, max diff {diff.max()}")
    assert d <= 1e-4 and rel <= 1e-4
    assert diff.max() <= 1 and (diff == 0).mean() >= 0.999
    # bit-exact given the SAME float input (the codec contract)
    img_same, _ = plan.image_encode(ref.cuda(), True, thr)
    assert np.array_equal(img_same[0].cpu().numpy(), ref_img)


def test_og_beat_mean_spectral_convergence_over_8_seeds(O, golden_dir):
    """configs[0] with production RNG on both sides.  One draw moves the figure by 1-2 % (both sides), so SURVEY 8(d)'s 1 % gate
    is applied to the MEAN over 8 independent initialisations per side."""
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    params = SpectrogramParams()
    op = O.params_from(params)
    plan = _plan(params)
    with Image.open(os.path.join(golden_dir, "og_beat.png")) as im:
        rgb = np.asarray(image_util.rgb_array_from_image(im))
    mel = torch.from_numpy(O.spectrogram_from_image_u8(rgb, 0.25, False, 30e6))
    n = 8
    sc_o, sc_d = [], []
    for s in range(n):
        torch.manual_seed(1000 + s)  # the reference draws both inits from torch's global generator
        lin_o = O.inverse_mel_scale_sgd(mel, op)
        sc_o.append(O.spectral_convergence(O.griffinlim(lin_o, op), lin_o, op))
    # device: the 8 draws as one batch of 8 independent clips (channels_per_clip = 1), seeds from the counter RNG
    mel8 = mel.cuda().repeat(n, 1, 1).contiguous()
    lin_slots = plan.inverse_mel(mel8, 1, seed=4242)
    waves = plan.griffinlim(lin_slots, n, T_FULL, 32, 0.99, seed=4243).cpu()
    lin_d = plan.unpack_magnitudes(lin_slots, n, T_FULL).cpu()
    for s in range(n):
        sc_d.append(O.spectral_convergence(waves[s : s + 1], lin_d[s : s + 1], op))
    mo, md = float(np.mean(sc_o)), float(np.mean(sc_d))
    print(f"og_beat spectral convergence over {n} seeds: oracle mean {mo:.5f} (std {np.std(sc_o):.5f}), device mean {md:.5f} "
          f"(std {np.std(sc_d):.5f}), relative difference of the means {abs(md - mo) / mo:.4f}")
    assert abs(md - mo) <= 0.01 * mo
    assert len(set(sc_d)) >= n - 1  # eight different draws, not one repeated (two float32 figures may coincide by chance)
