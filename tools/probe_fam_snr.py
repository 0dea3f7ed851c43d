"""Griffin-Lim SNR against the fp32 oracle after n iterations, row-family vs generic engine, several random inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import riffusion_oracle as O
from helpers import snr_db
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
torch.set_num_threads(16)
rate = int(os.environ.get("RATE", 48000))
p = SpectrogramParams(sample_rate=rate, max_frequency=min(10000, rate // 2))
op = O.params_from(p)
fam, gen = _hip.get_plan(p, "cuda"), _hip.get_plan(p, "cuda", frame_engine="generic")
B, T = 2, 46
for seed in range(int(os.environ.get("SEEDS", 5))):
    g = torch.Generator().manual_seed(1000 + seed)
    mag = torch.rand(B, op.n_stft, T, generator=g) * 1000
    a0 = torch.rand(B, op.n_stft, T, dtype=torch.complex64, generator=g)
    S, A = fam.pack_magnitudes(mag.cuda()), fam.pack_complex(a0.cuda())
    row = []
    for n in (1, 2, 3, 4, 8):
        want = O.griffinlim(mag, op, angles0=a0, n_iter=n)
        f = fam.griffinlim(S, B, T, n, 0.99, angles0_slots=A).cpu()
        q = gen.griffinlim(S, B, T, n, 0.99, angles0_slots=A).cpu()
        row.append(f"n={n}: fam {snr_db(want, f):6.1f} gen {snr_db(want, q):6.1f} f-g {snr_db(q, f):6.1f}")
    print(f"seed {seed}: " + " | ".join(row), flush=True)
