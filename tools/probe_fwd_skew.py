"""Round 6 probe: the forward kernel's run skew by dispatch order (RFX_FWD_SKEW per mille, -DRFX_ABLATION build via RFX_LIB_PATH).
One process per setting (the plan reads the switch at creation): 64 waveforms x 512 frames, rfx_image_from_waveform and
rfx_mel_from_waveform, best and mean of REPS blocks of 20 back-to-back calls."""
import os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import os, sys, time
sys.path.insert(0, os.path.join(sys.argv[1], "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util
plan = _hip.get_plan(SpectrogramParams(), "cuda")
wave = torch.from_numpy((np.random.default_rng(1).standard_normal((64, 441 * 511)) * 8000).astype(np.float32)).cuda()
thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
def block(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for fn, name in ((lambda: plan.image_from_waveform(wave, False, thr), "image_from_waveform"), (lambda: plan.mel_from_waveform(wave), "mel_from_waveform")):
    block(fn, 5)
    xs = [block(fn) for _ in range(int(os.environ.get("REPS", "6")))]
    print(f"  {name}: best {min(xs):.4f} ms, mean {sum(xs) / len(xs):.4f} ms per 64 waveforms")
"""
for skew in [int(x) for x in os.environ.get("SKEWS", "0,30,60,90,120,150,0,60,90").split(",")]:
    print(f"RFX_FWD_SKEW={skew}", flush=True)
    env = dict(os.environ, RFX_FWD_SKEW=str(skew))
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
    print(out.stdout.rstrip() or out.stderr[-800:], flush=True)
