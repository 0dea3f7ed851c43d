"""Round 6 probe: where do the run form and the per-frame form of Griffin-Lim differ (block, sample in block, ulps)?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

p = SpectrogramParams()
runs, frames = _hip.get_plan(p, "cuda", gl_form="runs"), _hip.get_plan(p, "cuda", gl_form="frames")
for B, Tn in ((1, 64), (3, 512)):
    g = torch.Generator(device="cuda").manual_seed(9 * B + Tn)
    mag = torch.rand(B, runs.n_stft, Tn, device="cuda", generator=g) * 1000.0
    S = runs.pack_magnitudes(mag)
    for n_iter in (0, 1):
        a = runs.griffinlim(S, B, Tn, n_iter, 0.99, seed=31).cpu().numpy()
        b = frames.griffinlim(S, B, Tn, n_iter, 0.99, seed=31).cpu().numpy()
        d = a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)
        bad = np.argwhere(d != 0)
        print(f"B={B} T={Tn} n_iter={n_iter}: {len(bad)} of {a.size} differ; max |ulps| {np.abs(d).max()}; rel err max {np.abs(a-b).max()/np.abs(a).max():.2e}")
        if len(bad):
            blk = bad[:, 1] // 441
            n = bad[:, 1] % 441
            hist = np.bincount(blk, minlength=Tn - 1)
            print("  per block:", " ".join(f"{i}:{c}" for i, c in enumerate(hist[:70]) if c))
            print("  n' histogram (by k1 = n'//21):", np.bincount(n // 21, minlength=21).tolist())
            print("  ulp histogram:", {int(k): int(v) for k, v in zip(*np.unique(np.clip(d[d != 0], -4, 4), return_counts=True))})
