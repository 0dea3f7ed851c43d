"""Parity probe for a librfx variant (RFX_LIB_PATH): Griffin-Lim SNR vs the CPU oracle at B=2, T=48 with injected
angles (oracle results cached under /tmp so that several variants share one oracle run), plus the STFT error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import riffusion_oracle as O
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

p = SpectrogramParams(); op = O.params_from(p)
plan = _hip.get_plan(p, "cuda")
g = torch.Generator().manual_seed(1234)
B, T = 2, 48
mag = torch.rand(B, op.n_stft, T, generator=g) * 1000
a0 = torch.rand(B, op.n_stft, T, dtype=torch.complex64, generator=g)
cache = "/tmp/rfx_parity_cache.pt"
if os.path.exists(cache):
    want = torch.load(cache)
else:
    torch.set_num_threads(16)
    want = {n: O.griffinlim(mag, op, angles0=a0, n_iter=n) for n in (0, 1, 4, 32)}
    wave = torch.randn(2, 441 * 60) * 8000
    want["wave"], want["stft"] = wave, O.stft_complex(wave, op)
    torch.save(want, cache)
S, A = plan.pack_magnitudes(mag.cuda()), plan.pack_complex(a0.cuda())
out = []
for n in (0, 1, 4, 32):
    have = plan.griffinlim(S, B, T, n, 0.99, angles0_slots=A).cpu()
    w = want[n].double()
    out.append(f"snr@{n}={float(10 * torch.log10(w.pow(2).sum() / (w - have.double()).pow(2).sum())):.1f}")
_, spec, Tn = plan.stft(want["wave"].cuda(), want_mag=False, want_spec=True)
got = plan.unpack_complex(spec, 2, Tn).cpu()
out.append(f"stft_rel={float((got - want['stft']).abs().max() / want['stft'].abs().max()):.2e}")
print(os.environ.get("RFX_LIB_PATH", "default"), " ".join(out))
