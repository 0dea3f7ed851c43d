"""Timing probe of the InverseMelScale kernel on B synthetic mono tiles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B = int(os.environ.get("B", 64)); T = 512
plan = _hip.get_plan(SpectrogramParams(), "cuda")
mel = torch.rand(B, 512, T, device="cuda") ** 4 * 3e7
for rep in range(5):
    torch.cuda.synchronize(); t = time.time()
    out = plan.inverse_mel(mel, 1, seed=rep)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"inverse_mel B={B}: {dt*1e3:.1f} ms  ({B/dt:.0f} tiles/s)  finite={bool(torch.isfinite(out).all())}")
