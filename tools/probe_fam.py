"""Griffin-Lim 32 alone on B synthetic 512-frame tiles (random magnitudes in the plan's layout), torch events, best of 4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B, T = int(os.environ.get("B", 64)), 512
out = []
for rate in [int(r) for r in os.environ.get("RATES", "48000").split(",")]:
    p = SpectrogramParams(sample_rate=rate, max_frequency=min(10000, rate // 2))
    plan = _hip.get_plan(p, "cuda", frame_engine=os.environ.get("ENGINE", "auto"))
    g = torch.Generator(device="cuda").manual_seed(1)
    S = torch.rand(B * T, plan.frame_stride, device="cuda", generator=g) * 1000
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w = plan.griffinlim(S, B, T, 32, 0.99, seed=3); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    out.append(f"{rate}: {min(ts[1:]):.1f} ms [{plan.griffinlim_engine}] sum|w| {float(w.abs().double().mean()):.6g}")
print(os.environ.get("TAG", "default") + "  " + " | ".join(out), flush=True)
