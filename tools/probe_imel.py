"""Timing probe of the InverseMelScale kernel on B synthetic mono tiles (torch events on the launch stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B = int(os.environ.get("B", 64)); T = 512
plan = _hip.get_plan(SpectrogramParams(), "cuda")
mel = torch.rand(B, 512, T, device="cuda") ** 4 * 3e7
ts = []
for rep in range(14):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = plan.inverse_mel(mel, 1, seed=rep)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
print(f"inverse_mel B={B}: min {ts[0]:.2f} ms, median {ts[len(ts)//2]:.2f} ms  ({B/ts[len(ts)//2]*1e3:.0f} tiles/s)  finite={bool(torch.isfinite(out).all())}")
