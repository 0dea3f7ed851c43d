"""
Is the clock the chip runs the path's kernels at a POWER limit?  Runs each kernel back to back for a few seconds while a host thread
polls the device's current sclk level and its averaged socket power (bench.py::ClockSampler), and prints both next to the cap.

    python tools/probe_power.py        (on the GPU box)
"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
import bench
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

B, T = 64, 512
plan = _hip.get_plan(SpectrogramParams(), "cuda")
S = torch.rand(B * T, plan.frame_stride, device="cuda") * 1e6
mel = torch.rand(B, 512, T, device="cuda") * 3e7
wave = torch.randn(B, 441 * (T - 1), device="cuda") * 8000


def run(name, fn, seconds=4.0):
    fn(); torch.cuda.synchronize()
    cs = bench.ClockSampler(0)
    cs.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cs.stop()
    s = cs.summary()
    half = cs.power[len(cs.power) // 2:]  # second half of the run: the driver's average has caught up
    p = s.get("socket_power_w", {})
    print(f"{name:34s} {dt / n * 1e3:8.3f} ms per call   sclk median {s.get('mhz_median')} MHz (min {s.get('mhz_min')}, max {s.get('mhz_max')})   "
          f"socket power: second half of the run {sum(half) / max(len(half), 1):7.1f} W (max {p.get('max')}), cap {p.get('cap')} W")


print("hwmon files:", sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_*"))[:6])
run("idle (host sleeps)", lambda: time.sleep(0.05), 2.0)
run("Griffin-Lim 32, 64 tiles", lambda: plan.griffinlim(S, B, T, 32, 0.99, seed=1))
run("InverseMelScale 200, 64 tiles", lambda: plan.inverse_mel(mel, 1, seed=1))
run("audio -> mel, 64 waveforms", lambda: plan.mel_from_waveform(wave))
run("Griffin-Lim 32, 16 tiles", lambda: plan.griffinlim(S[: 16 * T], 16, T, 32, 0.99, seed=1))
