"""The encoder on the forward kernel's frame-major scratch (rfx_image_from_waveform minus rfx_mel_from_waveform's kernel): time of the
one-call path against the mel kernel alone, 64 waveforms x 512 frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util
p = SpectrogramParams()
plan = _hip.get_plan(p, "cuda")
wave = torch.randn(64, p.hop_length * 511, device="cuda") * 8000
thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
def timed(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
one = timed(lambda: plan.image_from_waveform(wave, False, thr))
mel = timed(lambda: plan.mel_from_waveform(wave))
m = plan.mel_from_waveform(wave)
enc = timed(lambda: plan.image_encode(m, False, thr))
print(f"{os.environ.get('TAG', 'default')}  image_from_waveform {one:.3f} ms, mel_from_waveform (kernel + transpose) {mel:.3f} ms, image_encode (two-call form) {enc:.3f} ms")
