"""Batch-of-B vs clip-alone Griffin-Lim (injected init): per-clip SNR, to expose clip-boundary effects."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import snr_db
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

B = int(os.environ.get("B", 64)); T = int(os.environ.get("T", 512))
plan = _hip.get_plan(SpectrogramParams(), "cuda")
g = torch.Generator(device="cuda").manual_seed(7)
lin = plan.pack_magnitudes(torch.rand(B, plan.n_stft, T, device="cuda", generator=g) * 1e6)
ang = plan.pack_complex(torch.view_as_complex(torch.rand(B, plan.n_stft, T, 2, device="cuda", generator=g)))
for n_iter in (0, 1, 2, 3, 8, 32):
    wave = plan.griffinlim(lin, B, T, n_iter, 0.99, angles0_slots=ang)
    res = []
    for b in range(B):
        w1 = plan.griffinlim(lin[b * T:(b + 1) * T].contiguous(), 1, T, n_iter, 0.99, angles0_slots=ang[b * T:(b + 1) * T].contiguous())
        res.append(snr_db(w1, wave[b:b + 1]))
    r = torch.tensor(res)
    print(f"n_iter={n_iter}: min {r.min():.1f} dB at clip {int(r.argmin())}, median {r.median():.1f}, max {r.max():.1f};  last clip {res[-1]:.1f}, first {res[0]:.1f}")
    if n_iter == 32:
        print("  sorted lowest:", sorted((round(v, 1), i) for i, v in enumerate(res))[:8])
