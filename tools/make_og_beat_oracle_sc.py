"""
Oracle side of the og_beat production-RNG comparison (tests/test_gpu_round3_parity.py): spectral convergence of
InverseMelScale-200 -> Griffin-Lim-32 on seed_images/og_beat.png for N independent initialisations drawn like the reference
draws them (torch's global generator).  Pure CPU, deterministic per seed: run once in the build container, commit the table
(tests/golden/og_beat_oracle_sc.json); the GPU test compares the device's mean over freshly seeded draws with this mean.

    python tools/make_og_beat_oracle_sc.py [--seeds 32] [--device-rng]

--device-rng draws the two initialisations with the DEVICE's counter RNG (rfx_core.h::rand_unit / rand_unit_pair through the
host emulator) instead of torch.rand: same oracle arithmetic, the device's random stream - separates "the generator" from
"the kernels" when the two means differ.
"""
import argparse, ctypes, json, os, subprocess, sys, tempfile, time
import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("riffusion-hobby_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import riffusion_oracle as O
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=32)
ap.add_argument("--first-seed", type=int, default=1000)
ap.add_argument("--device-rng", action="store_true")
ap.add_argument("--out", default=None)
args = ap.parse_args()
torch.set_num_threads(min(16, os.cpu_count() or 1))
op = O.params_from(SpectrogramParams())
with Image.open(os.path.join(ROOT, "tests", "golden", "og_beat.png")) as im:
    rgb = np.asarray(image_util.rgb_array_from_image(im))
mel = torch.from_numpy(O.spectrogram_from_image_u8(rgb, 0.25, False, 30e6))
T = mel.shape[-1]
emu = None
if args.device_rng:
    so = os.path.join(tempfile.mkdtemp(), "librfx_emu.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emu", "rfx_emu.cpp")], check=True)
    emu = ctypes.CDLL(so)
    FP = ctypes.POINTER(ctypes.c_float)

def device_inits(seed):
    spec0 = np.zeros((1, T, op.n_stft), np.float32)
    ang = np.zeros((op.n_stft, T, 2), np.float32)
    row = np.zeros(op.n_stft, np.float32); pair = np.zeros(2 * op.n_stft, np.float32)
    for t in range(T):
        emu.emu_rand_unit(ctypes.c_ulonglong(seed), ctypes.c_ulonglong(t), op.n_stft, row.ctypes.data_as(FP)); spec0[0, t] = row
        emu.emu_rand_unit_pair(ctypes.c_ulonglong(seed + 1), ctypes.c_ulonglong(t), op.n_stft, pair.ctypes.data_as(FP)); ang[:, t] = pair.reshape(-1, 2)
    return torch.from_numpy(spec0), torch.view_as_complex(torch.from_numpy(ang))[None]

vals = []
t0 = time.time()
for s in range(args.first_seed, args.first_seed + args.seeds):
    if emu is None:
        torch.manual_seed(s)  # the reference draws both initialisations from torch's global generator
        lin = O.inverse_mel_scale_sgd(mel, op)
        wave = O.griffinlim(lin, op)
    else:
        spec0, ang0 = device_inits(s * 7919)
        lin = O.inverse_mel_scale_sgd(mel, op, spec0=spec0)
        wave = O.griffinlim(lin, op, angles0=ang0)
    vals.append(O.spectral_convergence(wave, lin, op))
    print(f"seed {s}: {vals[-1]:.6f}   ({time.time() - t0:.0f} s)", flush=True)
res = {"image": "tests/golden/og_beat.png", "pipeline": "InverseMelScale SGD-200 -> Griffin-Lim 32 (oracle/riffusion_oracle.py, torch CPU fp32)",
       "init": "device counter RNG via tests/emu" if emu else "torch.manual_seed(seed); torch.rand (as the reference)",
       "seeds": list(range(args.first_seed, args.first_seed + args.seeds)), "spectral_convergence": vals,
       "mean": float(np.mean(vals)), "std": float(np.std(vals)), "torch": torch.__version__}
out = args.out or os.path.join(ROOT, "tests", "golden", "og_beat_oracle_sc_device_rng.json" if emu else "og_beat_oracle_sc.json")
json.dump(res, open(out, "w"), indent=1)
print(f"mean {res['mean']:.6f} std {res['std']:.6f} -> {out}")
