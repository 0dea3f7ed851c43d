"""Batch-shape sweep of the decode step (uint8 tiles -> int16 PCM, Griffin-Lim 32) and of Griffin-Lim alone: tiles/s for every B
of BATCHES (default 16,48,64,65,96,100,128).  The run partition of rfx_griffinlim must not fall off a cliff when B does not
divide the resident workgroup slots.  RFX_LIB_PATH selects the library (A/B against a variant)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util

T, ITERS = 512, 32
dev = torch.device("cuda", 0)
plan = _hip.get_plan(SpectrogramParams(), dev)
lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)
rng = np.random.default_rng(1)
rows = {}
for B in [int(b) for b in os.environ.get("BATCHES", "16,48,64,65,96,100,128").split(",")]:
    tiles = torch.from_numpy(rng.integers(0, 256, size=(B, 512, T, 3), dtype=np.uint8)).to(dev)
    ws = torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, B, T), dtype=torch.uint8, device=dev)
    def step(seed):
        mel = plan.image_decode(tiles, False, lut)
        lin = plan.inverse_mel(mel, 1, seed=seed)
        wave = plan.griffinlim(lin, B, T, ITERS, 0.99, seed=seed + 1, workspace=ws)
        return plan.pcm16(wave, channels=1, normalize=True)[0], lin
    _, lin = step(0)
    torch.cuda.synchronize()
    n = 4
    t0 = time.perf_counter()
    for k in range(n):
        step(10 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    best = 1e9
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.griffinlim(lin, B, T, ITERS, 0.99, seed=r, workspace=ws)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    rows[B] = (dt * 1e3, B / dt, best * 1e3)
    print(f"B={B:4d}: step {dt*1e3:8.3f} ms  {B/dt:8.1f} tiles/s   griffinlim-32 {best*1e3:8.3f} ms  ({best*1e3/B:.4f} ms per tile)", flush=True)
if 64 in rows and 128 in rows:
    for B in rows:
        if 64 < B < 128:
            lin_ms = rows[64][0] + (rows[128][0] - rows[64][0]) * (B - 64) / 64.0
            print(f"B={B}: step is {100 * (rows[B][0] / lin_ms - 1):+.1f} % off the line between B=64 and B=128")
