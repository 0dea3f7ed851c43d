import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
plan = _hip.get_plan(SpectrogramParams(), "cuda")
rng = np.random.default_rng(1)
wave = torch.from_numpy((rng.standard_normal((2, 441 * 63)) * 8000).astype(np.float32)).cuda()
mel = plan.mel_from_waveform(wave).cpu()
ref = torch.load("/tmp/mel_ref_small.pt") if os.path.exists("/tmp/mel_ref_small.pt") and not os.environ.get("RFX_FWD_V1") else None
if os.environ.get("RFX_FWD_V1"):
    torch.save(mel, "/tmp/mel_ref_small.pt"); print("saved ref", mel.shape)
else:
    r = (mel / ref)
    print("shape", mel.shape)
    for m in [0, 1, 2, 3, 10, 63, 64, 100, 255, 256, 400, 447, 448, 449, 500, 511]:
        print(m, "ratio frames 0,1,30,63:", [round(float(r[0, m, t]), 4) for t in (0, 1, 30, 63)], " got", float(mel[0, m, 30]), "want", float(ref[0, m, 30]))
    bad = ((r - 1).abs() > 1e-3)
    print("bad fraction", float(bad.float().mean()), "bad per clip", bad.float().mean((1, 2)).tolist())
    print("bad by frame (clip 0):", bad[0].float().mean(0)[:16].tolist())
    print("bad by mel (clip 0) first 32:", bad[0].float().mean(1)[:32].tolist())
