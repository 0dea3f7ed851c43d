"""Phase timeline of the Griffin-Lim kernel (RFX_TIMING build): wall_clock64 (100 MHz) per phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
B, T = 64, 512
tim = torch.zeros(B * 16 * 7 * 12, dtype=torch.int64, device="cuda")
os.environ["RFX_TIMING_PTR"] = str(tim.data_ptr())
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
plan = _hip.get_plan(SpectrogramParams(), "cuda")
S = torch.rand(B * T, plan.frame_stride, device="cuda") * 1e6
out = plan.griffinlim(S, B, T, 6, 0.99, seed=1)
torch.cuda.synchronize()
nb = int(os.environ.get("NBLK", 512))
t = tim[: nb * 7 * 12].view(nb, 7, 12).double().cpu()
frames = T * B / nb
names = ["in+P1 (to pre-barrier)", "barrier B1", "P2+P3", "projection", "tw1/d issue (after P2')", "barrier B2", "P1'+OLA",
         "wait for |S| (vmcnt 0)", "P3'", "wave sync + P2'"]
tot = 0
for i, n in enumerate(names):
    us = t[:, :, i].mean().item() / frames / 100.0  # 100 MHz ticks -> us
    tot += us
    print(f"{n:28s} {us:7.2f} us/frame   (min wave {t[:,:,i].min().item()/frames/100:.2f}, max {t[:,:,i].max().item()/frames/100:.2f})")
print("sum", round(tot, 2), "us per WG-frame")
# per-workgroup totals: how far apart do the workgroups of one launch finish?  (the tail a launch boundary exposes)
tot_wg = t[:, :, :10].sum(dim=2).max(dim=1).values / 100.0 / 6.0   # us per launch (6 iterations ran; modes 1 + 2: 6 launches with timers)
print(f"per-workgroup time per launch: mean {tot_wg.mean().item():.1f} us, min {tot_wg.min().item():.1f}, max {tot_wg.max().item():.1f} "
      f"-> the slowest workgroup is {100 * (tot_wg.max().item() / tot_wg.mean().item() - 1):.1f} % behind the mean")
