"""Timing probe: Griffin-Lim 32 iterations on B synthetic mono tiles (slot-layout magnitudes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

B = int(os.environ.get("B", 64)); T = 512; n_iter = int(os.environ.get("ITERS", 32))
plan = _hip.get_plan(SpectrogramParams(), "cuda")
S = torch.rand(B * T, plan.frame_stride, device="cuda") * 1e6
ws = torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, B, T), dtype=torch.uint8, device="cuda")
for rep in range(int(os.environ.get("REPS", 3))):
    torch.cuda.synchronize(); t = time.time()
    out = plan.griffinlim(S, B, T, n_iter, 0.99, seed=rep, workspace=ws)
    torch.cuda.synchronize(); dt = time.time() - t
    alg = (20 * n_iter + 4) * 8821 * T * B
    print(f"B={B} iters={n_iter}: {dt*1e3:.1f} ms  {B/dt:.1f} tiles/s  {alg/dt/1e9:.0f} GB/s algorithmic  finite={bool(torch.isfinite(out).all())}")
