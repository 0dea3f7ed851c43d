"""Forward path (waveforms -> mel amplitudes) at a given sample rate, B clips of 512 frames, ten calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B, T = int(os.environ.get("B", 64)), 512
rate = int(os.environ.get("RATE", 48000))
p = SpectrogramParams(sample_rate=rate, max_frequency=min(10000, rate // 2))
plan = _hip.get_plan(p, "cuda", frame_engine=os.environ.get("ENGINE", "auto"))
wave = torch.randn(B, p.hop_length * (T - 1), device="cuda") * 8000
for _ in range(10):
    mel = plan.mel_from_waveform(wave)
torch.cuda.synchronize()
print(rate, plan.griffinlim_engine, tuple(mel.shape), bool(torch.isfinite(mel).all()))
