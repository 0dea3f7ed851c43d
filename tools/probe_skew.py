"""
Sweep of the Griffin-Lim run-length skew (rfx_api.hip::gl_partition; -DRFX_ABLATION build: RFX_GL_SKEW / RFX_GL_SKEW0 are read at
plan creation):   bash tools/build_variants.sh abl:"";  RFX_LIB_PATH=build_var/librfx_abl.so python tools/probe_skew.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams

B, T = int(os.environ.get("SKEW_B", 64)), 512
S = None

def time_gl(skew, skew0, n_iter=32, reps=6):
    global S
    os.environ["RFX_GL_SKEW"], os.environ["RFX_GL_SKEW0"] = str(skew), str(skew0)
    with _hip._plans_lock:
        _hip._plans.clear()
    plan = _hip.get_plan(SpectrogramParams(), "cuda", gl_form="runs")
    if S is None:
        S = torch.rand(B * T, plan.frame_stride, device="cuda") * 1e6
    for _ in range(2):
        plan.griffinlim(S, B, T, n_iter, 0.99, seed=1)
    torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.griffinlim(S, B, T, n_iter, 0.99, seed=1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best, tot = min(best, ms), tot + ms
    return best, tot / reps

if __name__ == "__main__":
    print(f"B = {B}: Griffin-Lim 32 (33 launches), best / mean of 6 calls, by skew of the iterations (init launch skew 0)")
    for rnd in range(2):  # twice, interleaved: the clock drifts
        for skew in (0, 40, 70, 90, 100, 110, 130, 160):
            b, m = time_gl(skew, 0)
            print(f"  skew {skew:4d}: {b:7.3f} / {m:7.3f} ms")
    print("init launch (n_iter = 0: the synthesis-only launch + combine), by its skew")
    for skew0 in (0, 100, 150, 200, 250, 300):
        b, m = time_gl(0, skew0, n_iter=0, reps=10)
        print(f"  skew0 {skew0:4d}: {b:7.4f} / {m:7.4f} ms")
