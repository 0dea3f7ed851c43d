"""Throughput of the generic engine: B synthetic 512-frame tiles at a given sample rate (tiles -> audio, audio -> mel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util

B, T = int(os.environ.get("B", 64)), 512
for rate in [int(r) for r in os.environ.get("RATES", "48000,22050,44100").split(",")]:
    p = SpectrogramParams(sample_rate=rate, max_frequency=min(10000, rate // 2))
    plan = _hip.get_plan(p, "cuda", gl_form=os.environ.get("GLFORM", "auto"), frame_engine=os.environ.get("ENGINE", "auto"))
    tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(B, 512, T, 3), dtype=np.uint8)).cuda()
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).cuda()
    def decode(seed):
        mel = plan.image_decode(tiles, False, lut)
        lin = plan.inverse_mel(mel, 1, seed=seed)
        wave = plan.griffinlim(lin, B, T, 32, 0.99, seed=seed + 1)
        return plan.pcm16(wave, channels=1, normalize=True)[0], lin
    wave_in = (torch.randn(B, p.hop_length * (T - 1), device="cuda") * 8000)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        pcm, lin = decode(rep)
        torch.cuda.synchronize(); t1 = time.time()
        mel = plan.mel_from_waveform(wave_in)
        torch.cuda.synchronize(); t2 = time.time()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); l2 = plan.inverse_mel(plan.image_decode(tiles, False, lut), 1, seed=9); e[1].record()
        plan.griffinlim(l2, B, T, 32, 0.99, seed=3); e[2].record(); torch.cuda.synchronize()
    print(f"{rate} Hz ({'generic' if plan.generic else 'specialised'} plan, Griffin-Lim on the {plan.griffinlim_engine} engine, n_fft {p.n_fft}): decode {B} tiles {1e3*(t1-t0):.1f} ms = {B/(t1-t0):.0f} tiles/s "
          f"(InverseMelScale {e[0].elapsed_time(e[1]):.1f} ms, Griffin-Lim 32 {e[1].elapsed_time(e[2]):.1f} ms); "
          f"forward {1e3*(t2-t1):.2f} ms = {B/(t2-t1):.0f} images/s; finite={bool(torch.isfinite(mel).all())}")
