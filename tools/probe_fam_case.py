"""One Griffin-Lim input, iteration by iteration: device vs fp32 oracle, and the oracle's own fp32-vs-fp64 distance; with the
largest per-bin deviation of the final spectra (is an outlier one bin's phase, or everywhere?).  RATE, SEED, B, T from the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import riffusion_oracle as O
from helpers import snr_db
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
torch.set_num_threads(16)
rate, seed = int(os.environ.get("RATE", 24000)), int(os.environ.get("SEED", 24001))
B, T = int(os.environ.get("B", 3)), int(os.environ.get("T", 47))
p = SpectrogramParams(sample_rate=rate, max_frequency=min(10000, rate // 2))
op = O.params_from(p)
plan = _hip.get_plan(p, "cuda")
g = torch.Generator().manual_seed(seed)
mag = torch.rand(B, op.n_stft, T, generator=g) * 1000
a0 = torch.rand(B, op.n_stft, T, dtype=torch.complex64, generator=g)
S, A = plan.pack_magnitudes(mag.cuda()), plan.pack_complex(a0.cuda())
for n in (1, 2, 3, 4):
    want = O.griffinlim(mag, op, angles0=a0, n_iter=n)
    w64 = O.griffinlim(mag, op, angles0=a0, n_iter=n, dtype=torch.float64)
    got = plan.griffinlim(S, B, T, n, 0.99, angles0_slots=A).cpu()
    err = (got.double() - want.double())
    per_clip = [snr_db(want[b], got[b]) for b in range(B)]
    # where in time does the error sit?  energy of the error per hop block of the worst clip
    wb = min(range(B), key=lambda b: per_clip[b])
    e = err[wb].reshape(-1, p.hop_length).pow(2).sum(1)
    top = torch.topk(e, 3)
    print(f"{os.environ.get('TAG','')} {rate} Hz seed {seed} n={n}: device vs oracle {snr_db(want, got):6.1f} dB (own {snr_db(w64, want):6.1f}); per clip {', '.join(f'{x:.1f}' for x in per_clip)}; "
          f"worst clip {wb}: error energy share of its top-3 hop blocks {float(top.values.sum() / e.sum()):.2f} at blocks {top.indices.tolist()}", flush=True)
