import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/riffusion-hobby_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
B, T = 64, 512
for kw in (dict(), dict(min_frequency=20, max_frequency=20000), dict(max_frequency=20000), dict(num_frequencies=256), dict(mel_scale_type="slaney"), dict(mel_scale_norm="slaney"), dict(max_frequency=16000)):
    p = SpectrogramParams(**kw)
    plan = _hip.get_plan(p, "cuda")
    mel = torch.rand(B, p.num_frequencies, T, device="cuda") * 3e7
    plan.inverse_mel(mel, 1, seed=1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): plan.inverse_mel(mel, 1, seed=1)
    e1.record(); torch.cuda.synchronize()
    print(f"{str(kw):60s} imel kernel {plan.lib.rfx_plan_imel_kernel(plan.handle)}  {e0.elapsed_time(e1)/3:8.2f} ms per 64 tiles")
