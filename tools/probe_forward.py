"""Forward path probe (BASELINE.json configs[2]): batch of 64 waveforms -> mel (STFT + MFMA GEMM) -> uint8 images."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util

B, L = int(os.environ.get("B", 64)), 441 * 511
plan = _hip.get_plan(SpectrogramParams(), "cuda")
rng = np.random.default_rng(20240807)
wave = torch.from_numpy((rng.standard_normal((B, L)) * 8000).astype(np.float32)).cuda()
thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
T = 1 + L // 441
def step():
    mel = plan.mel_from_waveform(wave)
    img, mx = plan.image_encode(mel, False, thr)
    return img
for _ in range(2): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record(); mag, _, _ = plan.stft(wave, want_mag=True, want_spec=False); ev[1].record()
mel = plan.mel_from_waveform(wave); ev[2].record(); plan.image_encode(mel, False, thr); ev[3].record()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
stft_ms = ev[0].elapsed_time(ev[1]); mel_ms = ev[1].elapsed_time(ev[2]); enc_ms = ev[2].elapsed_time(ev[3])
gemm_ms = mel_ms - stft_ms
dense_flops = 2.0 * 512 * 8821 * B * T


def executed_k():
    """K actually multiplied: 32-position blocks of the slot-ordered filterbank with a non-zero row (the list
    rfx_plan_create builds); positions follow slot_pos_f of csrc/rfx_core.h."""
    fb = np.asarray(_hip.mel_filterbank(8821, 0.0, 10000.0, 512, 44100, None, "htk"))
    live_bin = np.abs(fb).sum(1) > 0
    live = np.zeros(9408, bool)
    for k1 in range(21):
        for kp in range(441):
            k = k1 + 40 * kp
            q, kb = k1 * 21 + kp % 21, kp // 21
            qp = q + q // 63
            p = ((kb >> 2) * 448 + qp) * 4 + (kb & 3) if kb < 20 else 20 * 448 + qp
            live[p] = live_bin[k if k <= 8820 else 17640 - k]
    return 32 * int(live.reshape(-1, 32).any(1).sum())


K_EXEC = executed_k()
print(json.dumps({"workload": f"batch={B} waveforms of {L} samples -> mel -> uint8 image", "images_per_s": round(B / dt, 1),
                  "ms_per_step": round(dt * 1e3, 3), "stft_ms": round(stft_ms, 3), "mel_total_ms": round(mel_ms, 3),
                  "mel_gemm_ms_est": round(gemm_ms, 3), "image_encode_ms": round(enc_ms, 3),
                  "mel_gemm_dense_equiv_tflops": round(dense_flops / (gemm_ms * 1e-3) / 1e12, 1),
                  "mel_gemm_executed_k": K_EXEC,
                  "mel_gemm_executed_tflops": round(2.0 * 512 * K_EXEC * B * T / (gemm_ms * 1e-3) / 1e12, 1),
                  "fp32_mfma_peak_tflops": 157.3}))
