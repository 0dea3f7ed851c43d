"""Round 6 probe: Griffin-Lim 32 by batch size in both device forms (run-based kernel, per-frame kernel + fold): where is the crossover
now that runs are whole groups of 16 frames (a batch below 16 tiles cannot give every workgroup slot a run)?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

p = SpectrogramParams()
plans = {f: _hip.get_plan(p, "cuda", gl_form=f) for f in ("runs", "frames", "auto")}
T = 512
for B in [int(x) for x in os.environ.get("BATCHES", "1,2,3,4,5,6,7,8,10,12,14,16,20,24,32").split(",")]:
    S = torch.rand(B * T, plans["runs"].frame_stride, device="cuda") * 1000.0
    row = [f"B={B:3d}"]
    for f, plan in plans.items():
        ws = torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, B, T), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            plan.griffinlim(S, B, T, 32, 0.99, seed=1, workspace=ws)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            plan.griffinlim(S, B, T, 32, 0.99, seed=1, workspace=ws)
        torch.cuda.synchronize()
        row.append(f"{f} {(time.perf_counter() - t) / 5 * 1e3:7.3f} ms")
    row.append("auto takes " + ("frames" if plans["auto"].lib.rfx_griffinlim_form(plans["auto"].handle, B, T) == 2 else "runs"))
    print("  ".join(row), flush=True)
