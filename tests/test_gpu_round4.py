"""
Round 4 on the GPU: the parity gaps the round-3 review named, and the entry point's input side.

* 48 kHz (the rate cli.py:43 meets most; row-family kernels): one oracle tile from INSIDE a B = 64, T = 512 batch -
  InverseMelScale-200 and Griffin-Lim-32 with injected initial values - and the forward path of the same batch size;
* silent inputs (image_util.py:28-41 with max = 0, audio_util.py:23-28 with peak = 0): what numpy does, the kernels do;
* og_beat.png with production RNG: the device's mean spectral convergence over 32 FRESH seeds per run against the oracle's
  mean over 32 seeds (tests/golden/og_beat_oracle_sc.json, tools/make_og_beat_oracle_sc.py);
* banks wider than 512 filters through the fused forward path;
* host tiles in -> host audio out equals device tiles in; float-range validation without a host sync; the bounded plan cache
  returns its device memory.
"""
import json
import os
import time

import numpy as np
import pytest
import torch
from PIL import Image

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


def test_48k_oracle_tile_from_inside_a_full_batch(O):
    """The row-family kernels (fam_gl_kernel<2, 24, 20>, the generic plan's InverseMelScale) at the size the bench line's
    `other_sample_rates` reports: 64 tiles of 512 frames at 48 kHz in ONE batch; tile 21 carries host-drawn initial values."""
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    params = SpectrogramParams(sample_rate=48000)
    op = O.params_from(params)
    plan = _plan(params)
    assert plan.griffinlim_engine == "row-family" and plan.n_stft == 9601
    B, b = 64, 21
    dev = torch.device("cuda")
    tiles_np = synthetic_tiles_u8(B, seed=48000)
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)
    g = torch.Generator().manual_seed(4321)
    spec0_b = torch.rand(1, T_FULL, op.n_stft, generator=g)
    angles0_b = torch.rand(1, op.n_stft, T_FULL, dtype=torch.complex64, generator=g)
    gd = torch.Generator(device=dev).manual_seed(98)
    spec0 = torch.rand(B, T_FULL, op.n_stft, device=dev, generator=gd)
    spec0[b] = spec0_b[0].to(dev)
    angles0 = torch.view_as_complex(torch.rand(B, op.n_stft, T_FULL, 2, device=dev, generator=gd))
    angles0[b] = angles0_b[0].to(dev)
    a0_slots = plan.pack_complex(angles0)
    del angles0

    mel = plan.image_decode(torch.from_numpy(tiles_np).to(dev), False, lut)
    lin = plan.inverse_mel(mel, 1, spec0=spec0)
    del spec0
    mel_b = torch.from_numpy(O.spectrogram_from_image_u8(tiles_np[b], 0.25, False, 30e6))
    assert torch.equal(mel[b : b + 1].cpu(), mel_b)
    want_lin = O.inverse_mel_scale_sgd(mel_b, op, spec0=spec0_b)
    fs = plan.frame_stride
    got_lin = plan.unpack_magnitudes(lin[b * T_FULL : (b + 1) * T_FULL].contiguous(), 1, T_FULL).cpu()
    act = O.mel_filterbank(op).abs().sum(1) > 0
    rel = float(torch.linalg.norm(got_lin[:, act] - want_lin[:, act]) / torch.linalg.norm(want_lin[:, act]))
    assert rel <= 1e-3 and torch.equal(got_lin[:, ~act], want_lin[:, ~act])

    # Griffin-Lim 32 of the whole batch, tile b on the oracle's magnitudes so that only the iteration is compared; the floor is
    # the oracle's own fp32-vs-fp64 distance on this tile (the iteration is chaotic, how chaotic depends on the geometry)
    want = O.griffinlim(want_lin, op, angles0=angles0_b, n_iter=32)
    want64 = O.griffinlim(want_lin, op, angles0=angles0_b, n_iter=32, dtype=torch.float64)
    own = snr_db(want64, want)
    lin_sub = lin.clone()
    lin_sub[b * T_FULL : (b + 1) * T_FULL] = plan.pack_magnitudes(want_lin.to(dev))
    wave = plan.griffinlim(lin_sub, B, T_FULL, 32, 0.99, angles0_slots=a0_slots)
    s_gl = snr_db(want, wave[b : b + 1].cpu())
    print(f"48 kHz, tile {b} inside the B = 64 batch vs oracle: InverseMelScale rel-L2 {rel:.2e}; Griffin-Lim 32 {s_gl:.1f} dB "
          f"(the oracle's own fp32-vs-fp64 distance on this tile: {own:.1f} dB)")
    assert s_gl >= min(60.0, own)
    assert fs == plan.frame_stride


def test_48k_forward_of_a_full_batch_matches_oracle(O):
    """mel_from_waveform at B = 64, T = 512, 48 kHz (fam_fwd_kernel + the banded projection): four clips of the batch through
    the oracle (the reference's 1e-4 gates)."""
    from riffusion.spectrogram_params import SpectrogramParams

    params = SpectrogramParams(sample_rate=48000)
    op = O.params_from(params)
    plan = _plan(params)
    B = 64
    wave = synthetic_wave(B, params.hop_length * (T_FULL - 1), seed=4848)
    mel = plan.mel_from_waveform(wave.cuda()).cpu()
    assert mel.shape == (B, 512, T_FULL)
    worst = 0.0
    for c in (0, 21, 42, 63):
        ref = O.mel_amplitudes_from_waveform(wave[c : c + 1], op)
        assert (mel[c : c + 1] - ref).abs().max() <= 1e-4 * ref.max()
        r = float(torch.linalg.norm(mel[c : c + 1] - ref) / torch.linalg.norm(ref))
        worst = max(worst, r)
        assert r <= 1e-4
    print(f"48 kHz forward, B = 64 x T = 512: worst rel-L2 of four clips vs oracle {worst:.2e}")


def test_silent_inputs_do_what_numpy_does(O):
    """image_util.py:28-41 on an all-zero spectrogram: max = 0, 0/0 = NaN through the power curve, and NaN -> uint8 is 0 on the
    reference's platform (x86-64 cvttss2si; numpy warns "invalid value encountered in cast"): a black-is-zero image, MAX_VALUE 0.
    audio_util.py:23-28 on an all-zero waveform: 32767 / 0 = inf, 0 * inf = NaN, NaN -> int16 is 0.  The kernels produce the same
    bytes, no NaN-dependent garbage."""
    import warnings

    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    plan = _plan(SpectrogramParams())
    thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
    for stereo in (False, True):
        C = 2 if stereo else 1
        zeros = torch.zeros(2 * C, 512, 64, device="cuda")
        zeros[C:] = torch.rand(C, 512, 64, device="cuda") * 1e6  # the second clip is ordinary: per-clip max
        img, mx = plan.image_encode(zeros, stereo, thr)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want0 = O.image_u8_from_spectrogram(zeros[:C].cpu().numpy(), 0.25)
            want1 = O.image_u8_from_spectrogram(zeros[C:].cpu().numpy(), 0.25)
        assert float(mx[0]) == 0.0 and np.array_equal(img[0].cpu().numpy(), want0) and not img[0].any()
        assert np.array_equal(img[1].cpu().numpy(), want1)
        wave = torch.zeros(2 * C, 4410, device="cuda")
        wave[C:] = torch.randn(C, 4410, device="cuda") * 100
        pcm, peak = plan.pcm16(wave, channels=C, normalize=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w0 = O.pcm16_from_waveform(wave[:C].cpu().numpy().copy(), normalize=True)
            w1 = O.pcm16_from_waveform(wave[C:].cpu().numpy().copy(), normalize=True)
        assert float(peak[0]) == 0.0 and not pcm[0].any() and np.array_equal(pcm[0].cpu().numpy(), w0)
        assert np.array_equal(pcm[1].cpu().numpy(), w1)


def test_og_beat_mean_spectral_convergence_32_fresh_seeds(O, golden_dir):
    """configs[0] with production RNG on both sides: 32 oracle draws, 64 device draws.  The oracle's 32 values are a committed table
    (pure CPU, deterministic per seed); the device draws 64 NEW initialisations on every run (seed from the clock, printed), so a
    generator whose statistics were off would show as a shifted mean run after run.  Gate: 1 % on the means; the
    means have sigma 0.17 % (oracle, 32) and 0.18 % (device, 64), the device sat +0.4 % above the table in four sessions."""
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    table = json.load(open(os.path.join(golden_dir, "og_beat_oracle_sc.json")))
    mo, so = float(table["mean"]), float(table["std"])
    assert len(table["spectral_convergence"]) >= 32
    params = SpectrogramParams()
    op = O.params_from(params)
    plan = _plan(params)
    with Image.open(os.path.join(golden_dir, "og_beat.png")) as im:
        rgb = np.asarray(image_util.rgb_array_from_image(im))
    mel = torch.from_numpy(O.spectrogram_from_image_u8(rgb, 0.25, False, 30e6))
    n = 64  # twice the oracle's count: the device's fresh draws are the cheap side, and the gate should not flake on them
    seed = int(time.time() * 1000) & 0x7FFFFFFF
    mel_n = mel.cuda().repeat(n, 1, 1).contiguous()
    lin_slots = plan.inverse_mel(mel_n, 1, seed=seed)
    waves = plan.griffinlim(lin_slots, n, T_FULL, 32, 0.99, seed=seed + 1).cpu()
    lin_d = plan.unpack_magnitudes(lin_slots, n, T_FULL).cpu()
    sc_d = [O.spectral_convergence(waves[s : s + 1], lin_d[s : s + 1], op) for s in range(n)]
    md, sd = float(np.mean(sc_d)), float(np.std(sc_d))
    print(f"og_beat spectral convergence, 32 oracle / {n} device seeds (device seed {seed}): oracle mean {mo:.5f} (std {so:.5f}), device mean {md:.5f} "
          f"(std {sd:.5f}), relative difference of the means {(md - mo) / mo:+.4f}")
    assert abs(md - mo) <= 0.01 * mo
    # different draws, not one repeated.  (Not "== n": the figure is a float32 with ~5e5 equally likely values around its mean - two
    # of 64 draws coincide in one run of 270, and did in this round's last visit; rounded to six digits they collide in one run of 4.)
    assert len(set(sc_d)) >= n - 2
    # the exact form of the same property: no two clips share a waveform or a starting spectrogram
    assert len({waves[s].numpy().tobytes() for s in range(n)}) == n and len({lin_d[s].numpy().tobytes() for s in range(n)}) == n


def test_fused_forward_with_more_than_512_filters(O):
    """num_frequencies = 768: the product-form kernel's sum phase covers 512 filters, wider banks must take the table form and
    fill every row of the result (rows >= 512 were never written)."""
    from riffusion.spectrogram_params import SpectrogramParams

    params = SpectrogramParams(num_frequencies=768)
    op = O.params_from(params)
    plan = _plan(params)
    wave = synthetic_wave(2, 441 * 40, seed=768)
    ref = O.mel_amplitudes_from_waveform(wave, op)
    torch.cuda.empty_cache()
    junk = torch.full((64, 1024, 1024), float("nan"), device="cuda")  # poison what the allocator hands out next
    del junk
    mel = plan.mel_from_waveform(wave.cuda()).cpu()
    assert mel.shape == ref.shape == (2, 768, 41) and bool(torch.isfinite(mel).all())
    assert (mel - ref).abs().max() <= 1e-4 * ref.max()
    assert torch.linalg.norm(mel - ref) / torch.linalg.norm(ref) <= 1e-4
    assert torch.linalg.norm(mel[:, 512:] - ref[:, 512:]) / torch.linalg.norm(ref[:, 512:]) <= 1e-4


def test_host_tiles_in_equal_device_tiles_in():
    """The reference's API is host images in, host audio out (spectrogram_image_converter.py:65-91): a pageable numpy batch goes
    through pinned staging and side-stream uploads chunk by chunk (batch_shard.ChunkSource) and must give the bytes a resident
    device tensor gives - ragged last chunk, mono and stereo, also from an already pinned tensor."""
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    for stereo in (False, True):
        conv = SpectrogramImageConverter(SpectrogramParams(stereo=stereo, num_griffin_lim_iters=4, max_mel_iters=20), device="cuda")
        tiles = synthetic_tiles_u8(5, width=64, seed=11 + stereo)
        on_dev = conv.audio_from_spectrogram_images(torch.from_numpy(tiles).cuda(), seed=5, tiles_per_call=2)
        from_host = conv.audio_from_spectrogram_images(tiles, seed=5, tiles_per_call=2)
        pinned = torch.from_numpy(tiles).pin_memory()
        from_pinned = conv.audio_from_spectrogram_images(pinned, seed=5, tiles_per_call=2)
        one_chunk = conv.audio_from_spectrogram_images(tiles, seed=5, tiles_per_call=64)
        assert on_dev.dtype == np.int16 and on_dev.shape[0] == 5
        assert np.array_equal(on_dev, from_host) and np.array_equal(on_dev, from_pinned)
        # chunking changes the seeds per chunk start only: the same chunking from one upload must agree with itself
        assert np.array_equal(one_chunk, conv.audio_from_spectrogram_images(torch.from_numpy(tiles).cuda(), seed=5, tiles_per_call=64))


def test_float_range_check_modes():
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    conv = SpectrogramImageConverter(SpectrogramParams(num_griffin_lim_iters=2, max_mel_iters=10), device="cuda")
    good = torch.rand(1, 512, 32, 3)
    bad, nan = good.clone(), good.clone()
    bad[0, 3, 3, 1] = 1.5
    nan[0, 7, 7, 0] = float("nan")
    ref = conv.audio_from_spectrogram_images(conv.quantize_pipeline_images(good), seed=1)
    assert np.array_equal(ref, conv.audio_from_spectrogram_images(good, seed=1))          # host float: checked at once
    assert np.array_equal(ref, conv.audio_from_spectrogram_images(good.cuda(), seed=1))   # device float: deferred check passes
    for x in (bad, nan):
        with pytest.raises(ValueError, match="0, 1"):
            conv.audio_from_spectrogram_images(x, seed=1)                                  # host: at once
        with pytest.raises(ValueError, match="0, 1"):
            conv.audio_from_spectrogram_images(x.cuda(), seed=1)                           # device: at the result's sync
        with pytest.raises(ValueError, match="0, 1"):
            conv.audio_from_spectrogram_images(x.cuda(), seed=1, validate=True, return_device=True)
        out = conv.audio_from_spectrogram_images(x.cuda(), seed=1, validate=False)         # the server path: no check
        assert out.shape == ref.shape
        # round 5: return_device + device input never synchronises, so it cannot raise - a failed check gives silence and a flag
        dev = conv.audio_from_spectrogram_images(x.cuda(), seed=1, return_device=True)
        assert dev.is_cuda and dev.shape == ref.shape and not bool(dev.range_ok) and int(dev.abs().max()) == 0
    dev = conv.audio_from_spectrogram_images(good.cuda(), seed=1, return_device=True)
    assert bool(dev.range_ok) and np.array_equal(dev.cpu().numpy(), ref)
    # round 6: the flag as a return value of its own (an attribute is lost by the first slice), and always there
    out, ok = conv.audio_from_spectrogram_images(bad.cuda(), seed=1, return_device=True, return_range_flag=True)
    assert not bool(ok) and int(out.abs().max()) == 0 and not hasattr(out[:1], "range_ok")
    out, ok = conv.audio_from_spectrogram_images(conv.quantize_pipeline_images(good).cuda(), seed=1, return_device=True, return_range_flag=True)
    assert bool(ok) and bool(out.range_ok) and np.array_equal(out.cpu().numpy(), ref)   # uint8 input: nothing to check, flag True
    with pytest.raises(ValueError, match="return_device"):
        conv.audio_from_spectrogram_images(good, seed=1, return_range_flag=True)


def test_plan_cache_returns_device_memory():
    """_hip.get_plan keeps at most PLAN_CACHE_SIZE plans (least recently used out); an evicted plan nobody holds is destroyed and
    its tables (>= 18 MB: the dense filterbank) go back to the device."""
    import gc

    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    cap = _hip.PLAN_CACHE_SIZE
    for i in range(cap + 1):  # fill the cache with throw-away parameter sets
        _hip.get_plan(SpectrogramParams(max_mel_iters=150 + i), "cuda")
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(cap + 1, 3 * cap + 1):
        _hip.get_plan(SpectrogramParams(max_mel_iters=150 + i), "cuda")
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert _hip.cached_plans() == cap
    print(f"plan cache: {cap} plans resident, free memory moved by {(free0 - free1) / 1e6:.1f} MB while {2 * cap} more plans came and went")
    assert free0 - free1 < 18e6  # less than ONE plan's tables: nothing accumulates
    # a plan that a caller still holds survives its eviction
    held = _hip.get_plan(SpectrogramParams(max_mel_iters=99), "cuda")
    for i in range(cap + 1):
        _hip.get_plan(SpectrogramParams(max_mel_iters=300 + i), "cuda")
    assert held.handle and held.mel_from_waveform(synthetic_wave(1, 441 * 45).cuda()).shape == (1, 512, 46)


def test_inverse_mel_wave_kernel_against_group_kernels_and_oracle(O):
    """The two InverseMelScale kernel families on the default bank: the wave kernel (rfx_plan_imel_kernel 4: one wave per frame,
    weights as a line per group, neighbours through DPP wave shifts) and the group kernels it replaced as the default
    (rfx_plan_options.imel_form = GROUPS, kernel 2).  Same injected start: both within the gate of the oracle and much closer to
    each other; same seed and no injected start: both draw the same initial values per (frame, bin), so they stay as close;
    untouched bins and duplicate slots bit for bit; stereo clips couple through the loss scale only."""
    from riffusion.spectrogram_params import SpectrogramParams

    params = SpectrogramParams()
    op = O.params_from(params)
    wave_plan, group_plan = _plan(params), _plan(params, imel_form="groups")
    assert wave_plan.lib.rfx_plan_imel_kernel(wave_plan.handle) == 4 and group_plan.lib.rfx_plan_imel_kernel(group_plan.handle) == 2
    act = O.mel_filterbank(op).abs().sum(1) > 0
    for C, T in ((1, 40), (2, 24)):
        g = torch.Generator().manual_seed(100 + C)
        mel = (torch.rand(C, 512, T, generator=g) ** 3 * 2e7)
        spec0 = torch.rand(C, T, op.n_stft, generator=g)
        want = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
        slots_w = wave_plan.inverse_mel(mel.cuda(), C, spec0=spec0.cuda())
        got_w = wave_plan.unpack_magnitudes(slots_w, C, T).cpu()
        got_g = group_plan.unpack_magnitudes(group_plan.inverse_mel(mel.cuda(), C, spec0=spec0.cuda()), C, T).cpu()
        nrm = torch.linalg.norm(want[:, act])
        rel_w, rel_g = float(torch.linalg.norm(got_w[:, act] - want[:, act]) / nrm), float(torch.linalg.norm(got_g[:, act] - want[:, act]) / nrm)
        rel_wg = float(torch.linalg.norm(got_w[:, act] - got_g[:, act]) / nrm)
        print(f"InverseMelScale C={C} T={T}: wave kernel vs oracle {rel_w:.2e}, group kernels vs oracle {rel_g:.2e}, wave vs group {rel_wg:.2e}")
        assert rel_w <= 1e-3 and rel_g <= 1e-3 and rel_wg <= 1e-5
        assert torch.equal(got_w[:, ~act], spec0.transpose(1, 2)[:, ~act])  # untouched bins: the injected start, bit for bit
        assert torch.equal(wave_plan.pack_magnitudes(got_w.cuda()), slots_w)  # duplicate slots carry their primary's value
        assert float(got_w.min()) >= 0.0 and bool(torch.isfinite(got_w).all())
    # seeded start: the same (seed, frame, bin) stream in both kernels
    mel = (torch.rand(3, 512, 16, generator=torch.Generator().manual_seed(9)) ** 3 * 2e7).cuda()
    a = wave_plan.unpack_magnitudes(wave_plan.inverse_mel(mel, 1, seed=77), 3, 16)
    b = group_plan.unpack_magnitudes(group_plan.inverse_mel(mel, 1, seed=77), 3, 16)
    assert torch.equal(a[:, ~act.cuda()], b[:, ~act.cuda()])  # pass-through bins ARE the drawn values
    assert float(torch.linalg.norm(a - b) / torch.linalg.norm(b)) <= 1e-5
    assert not torch.equal(a, wave_plan.unpack_magnitudes(wave_plan.inverse_mel(mel, 1, seed=78), 3, 16))
