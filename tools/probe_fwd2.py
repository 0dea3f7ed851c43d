"""Forward path probe: fused mel kernel timing (HIP events, drained per launch) and a parity check of the product form against the table form."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B, L = int(os.environ.get("B", 64)), 441 * 511
plan = _hip.get_plan(SpectrogramParams(), "cuda")
rng = np.random.default_rng(20240807)
wave = torch.from_numpy((rng.standard_normal((B, L)) * 8000).astype(np.float32)).cuda()
for _ in range(3): mel = plan.mel_from_waveform(wave)
ts = []
for _ in range(8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mel = plan.mel_from_waveform(wave); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"mel_from_waveform B={B}: min {min(ts):.3f} ms, median {sorted(ts)[len(ts)//2]:.3f} ms  sum={float(mel.double().sum()):.6e} finite={bool(torch.isfinite(mel).all())}")
if os.environ.get("REF"):
    torch.save(mel.cpu(), os.environ["REF"])
elif os.environ.get("CMP") and os.path.exists(os.environ["CMP"]):
    ref = torch.load(os.environ["CMP"])
    d = (mel.cpu() - ref)
    print(f"  vs table form: max|d|/max {float(d.abs().max() / ref.max()):.2e}, rel-L2 {float(torch.linalg.norm(d) / torch.linalg.norm(ref)):.2e}")
