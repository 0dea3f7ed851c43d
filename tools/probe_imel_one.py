"""One InverseMelScale configuration, a few calls (for rocprofv3 --kernel-trace --stats): ITERS steps on B synthetic mono tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B, T, iters = int(os.environ.get("B", 64)), 512, int(os.environ.get("ITERS", 200))
mel = torch.rand(B, 512, T, device="cuda") ** 4 * 3e7
plan = _hip.get_plan(SpectrogramParams(max_mel_iters=iters), "cuda")
for rep in range(6):
    out = plan.inverse_mel(mel, 1, seed=rep)
torch.cuda.synchronize()
print("done", iters, bool(torch.isfinite(out).all()))
