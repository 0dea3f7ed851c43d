"""
Is the tail of a Griffin-Lim launch the same workgroups every time?  (RFX_WGCLOCK build of librfx: every workgroup of every launch
records start / end on the 100 MHz wall clock and its XCD / SE / CU.)

    bash tools/build_variants.sh wgclock:"-DRFX_WGCLOCK"        (here, no GPU needed)
    RFX_LIB_PATH=build_var/librfx_wgclock.so python tools/probe_wgclock.py      (on the GPU box)

A launch ends when its slowest workgroup does.  If the slow workgroups were a different set each launch, one persistent launch over
all iterations with neighbour-to-neighbour hand-off (a run needs only its two neighbours' previous iteration) would average the
tail away; if they are the same set - a place on the chip - it would not, and only a different split of the frames could.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np
import torch

B, T, NIT = 64, 512, 32
NB = 512
tim = torch.zeros((NIT + 1) * NB * 4, dtype=torch.int64, device="cuda")
os.environ["RFX_TIMING_PTR"] = str(tim.data_ptr())
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

plan = _hip.get_plan(SpectrogramParams(), "cuda", gl_form="runs")
S = torch.rand(B * T, plan.frame_stride, device="cuda") * 1e6
for rep in range(3):  # the last repetition's records stay
    plan.griffinlim(S, B, T, NIT, 0.99, seed=1)
torch.cuda.synchronize()
rec = tim.view(NIT + 1, NB, 4).cpu().numpy()
if os.environ.get("WGCLOCK_DUMP"):
    np.save(os.environ["WGCLOCK_DUMP"], rec)
t0, t1, hw, xcc, cyc = rec[..., 0], rec[..., 1], rec[..., 2], rec[..., 3] & 15, rec[..., 3] >> 8
dur = (t1 - t0) / 100.0  # us
it = slice(2, NIT + 1)   # MODE 2 launches
d = dur[it]
launch_len = (t1[it].max(axis=1) - t0[it].min(axis=1)) / 100.0
start_skew = (t0[it].max(axis=1) - t0[it].min(axis=1)) / 100.0
print(f"MODE 2 launches: {d.shape[0]}; launch length (first start to last end) mean {launch_len.mean():.1f} us; workgroup duration mean {d.mean():.1f} us, "
      f"mean of the per-launch maxima {d.max(axis=1).mean():.1f} us ({100 * (d.max(axis=1).mean() / d.mean() - 1):.1f} % above the mean); start skew {start_skew.mean():.1f} us")
rel = d / d.mean(axis=1, keepdims=True)          # every launch normalised by its own mean (the clock drifts from launch to launch)
per_wg = rel.mean(axis=0)                         # systematic part: a workgroup's average over the launches
resid = rel - per_wg[None, :]
print(f"relative duration: spread of the per-workgroup averages (systematic) {per_wg.std():.4f}, min {per_wg.min():.3f}, max {per_wg.max():.3f}; "
      f"spread of what is left (launch to launch) {resid.std():.4f}")
print(f"if every launch ended with its own slowest workgroup: {rel.max(axis=1).mean():.4f} of the mean; if only the systematic part counted: {per_wg.max():.4f}; "
      f"a perfect hand-off pipeline would approach the slowest workgroup's average = {per_wg.max():.4f} (vs {rel.max(axis=1).mean():.4f} now)")
first, second = slice(0, NB // 2), slice(NB // 2, NB)
print(f"first-dispatched workgroups (blocks 0..{NB // 2 - 1}): {d[:, first].mean():.1f} us, the ones that join them: {d[:, second].mean():.1f} us; "
      f"counter of s_memtime per us of wall clock (MHz if it counts shader cycles): first {(cyc[it][:, first] / d[:, first]).mean():.1f}, second {(cyc[it][:, second] / d[:, second]).mean():.1f}")
# where the slow ones sit
cu = (hw[NIT] >> 8) & 15; se = (hw[NIT] >> 13) & 7; x = xcc[NIT]
same_place = all((xcc[i] == x).all() and (((hw[i] >> 8) & 15) == cu).all() for i in range(2, NIT + 1))
print(f"workgroup -> (XCD, SE, CU) identical in every launch: {same_place}")
for name, key in (("XCD", x), ("SE", se)):
    print(f"by {name}: " + "  ".join(f"{k}: {per_wg[key == k].mean():.4f}" for k in sorted(set(key.tolist()))))
order = np.argsort(-per_wg)[:12]
print("slowest workgroups (block: relative duration, XCD/SE/CU): " + "  ".join(f"{b}: {per_wg[b]:.3f} {x[b]}/{se[b]}/{cu[b]}" for b in order))
pair = {}
for b in range(NB):
    pair.setdefault((int(x[b]), int(se[b]), int(cu[b])), []).append(b)
sizes = sorted(len(v) for v in pair.values())
print(f"distinct (XCD, SE, CU) places: {len(pair)}; workgroups per place: min {sizes[0]}, max {sizes[-1]}")
blk = np.arange(NB)
print(f"correlation of the systematic part with blockIdx parity: even {per_wg[blk % 2 == 0].mean():.4f} odd {per_wg[blk % 2 == 1].mean():.4f}; "
      f"first / last run of a clip (block % 8 == 0 / 7): {per_wg[blk % 8 == 0].mean():.4f} / {per_wg[blk % 8 == 7].mean():.4f}, others {per_wg[(blk % 8 != 0) & (blk % 8 != 7)].mean():.4f}")
