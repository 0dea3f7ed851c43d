"""Latency probe: one 512x512 tile (and small batches) through the full uint8 -> int16 path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util

out = {}
for stereo in (False, True):
    p = SpectrogramParams(stereo=stereo)
    plan = _hip.get_plan(p, "cuda")
    C = 2 if stereo else 1
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).cuda()
    for B in (1, 2, 4, 8, 16, 32):
        tiles = torch.from_numpy(np.random.default_rng(B).integers(0, 256, size=(B, 512, 512, 3), dtype=np.uint8)).cuda()
        def step(seed):
            mel = plan.image_decode(tiles, stereo, lut)
            lin = plan.inverse_mel(mel, C, seed=seed)
            wave = plan.griffinlim(lin, B * C, 512, 32, 0.99, seed=seed + 1)
            return plan.pcm16(wave, channels=C, normalize=True)[0]
        for i in range(2): step(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        K = 5
        for i in range(K): step(10 + i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
        out[f"{'stereo' if stereo else 'mono'}_B{B}"] = {"ms": round(dt * 1e3, 2), "tiles_per_s": round(B / dt, 1)}
print(json.dumps(out))
