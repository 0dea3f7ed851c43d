"""Forward path (waveforms -> mel amplitudes -> uint8 images) at the given sample rates: B clips of 512 frames, images per second
(torch events around ten calls) and a spot check against the dense definition (|STFT| through torch.stft, times the filterbank)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util
B, T = int(os.environ.get("B", 64)), 512
out = []
for rate in [int(r) for r in os.environ.get("RATES", "48000").split(",")]:
    p = SpectrogramParams(sample_rate=rate, max_frequency=min(10000, rate // 2))
    plan = _hip.get_plan(p, "cuda", frame_engine=os.environ.get("ENGINE", "auto"))
    wave = torch.randn(B, p.hop_length * (T - 1), device="cuda") * 8000
    thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
    def step():
        mel = plan.mel_from_waveform(wave)
        return mel, plan.image_encode(mel, False, thr)[0]
    for _ in range(3): mel, img = step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    e0.record()
    for _ in range(20): plan.mel_from_waveform(wave)
    e1.record(); torch.cuda.synchronize()
    ms_mel = e0.elapsed_time(e1) / 20
    # spot check of clip 0 against torch.stft on the device (the reference's definition, dense filterbank)
    win = torch.hann_window(p.win_length, device="cuda")
    ref = torch.stft(wave[:1], p.n_fft, p.hop_length, p.win_length, win, center=True, pad_mode="reflect", return_complex=True).abs()
    ref_mel = (ref.transpose(1, 2) @ plan.melfb.cuda()).transpose(1, 2)
    rel = float(torch.linalg.norm(mel[:1] - ref_mel) / torch.linalg.norm(ref_mel))
    out.append(f"{rate}: {ms:.3f} ms per {B} clips (mel alone {ms_mel:.3f}) = {B / ms * 1e3:.0f} images/s [{plan.griffinlim_engine}] rel-L2 vs torch.stft {rel:.1e}")
print(os.environ.get("TAG", "default") + "  " + " | ".join(out), flush=True)
