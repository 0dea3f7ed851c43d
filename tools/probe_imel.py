"""Timing probe of the InverseMelScale kernels on B synthetic mono tiles (torch events on the launch stream): the plan's default
kernel (the wave kernel on the default bank) and the group kernels (rfx_plan_options.imel_form = GROUPS), alternating."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B = int(os.environ.get("B", 64)); T = 512
forms = os.environ.get("FORMS", "auto,groups").split(",")
plans = {f: _hip.get_plan(SpectrogramParams(), "cuda", imel_form=f) for f in forms}
mel = torch.rand(B, 512, T, device="cuda") ** 4 * 3e7
ts = {f: [] for f in forms}
outs = {}
for rep in range(int(os.environ.get("REPS", 12))):
    for f in forms:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs[f] = plans[f].inverse_mel(mel, 1, seed=rep)
        e1.record(); torch.cuda.synchronize()
        ts[f].append(e0.elapsed_time(e1))
for f in forms:
    t = sorted(ts[f][2:])
    print(f"inverse_mel B={B} imel_form={f} (kernel {plans[f].lib.rfx_plan_imel_kernel(plans[f].handle)}): min {t[0]:.2f} ms, median {t[len(t)//2]:.2f} ms  "
          f"({B/t[len(t)//2]*1e3:.0f} tiles/s)  finite={bool(torch.isfinite(outs[f]).all())}")
if len(forms) == 2:
    a, b = outs[forms[0]], outs[forms[1]]
    print(f"same seed, two kernels: rel-L2 {float(torch.linalg.norm(a - b) / torch.linalg.norm(b)):.2e}")
