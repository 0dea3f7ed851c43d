"""InverseMelScale on 64 synthetic mono tiles at several step counts (SpectrogramParams.max_mel_iters): the slope is the SGD loop,
the intercept what the kernel spends loading a frame (mel targets T floats apart, the random start) and storing it (4-byte stores
into slot order), plus the scan / fix-up launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B, T = int(os.environ.get("B", 64)), 512
mel = torch.rand(B, 512, T, device="cuda") ** 4 * 3e7
rows = []
for iters in (1, 50, 100, 200, 400):
    plan = _hip.get_plan(SpectrogramParams(max_mel_iters=iters), "cuda")
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = plan.inverse_mel(mel, 1, seed=rep); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    rows.append((iters, sorted(ts[2:])[len(ts[2:]) // 2]))
    print(f"{os.environ.get('TAG', 'default')}  max_mel_iters = {iters:4d}: {rows[-1][1]:.3f} ms per {B} tiles", flush=True)
slope = (rows[-1][1] - rows[1][1]) / (rows[-1][0] - rows[1][0])
print(f"{os.environ.get('TAG', 'default')}  per step {slope * 1e3:.2f} us, intercept {rows[3][1] - 200 * slope:.3f} ms of the {rows[3][1]:.3f} ms at 200 steps")
