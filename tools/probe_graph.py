"""Round 6 probe: is the one-tile decode capturable in a HIP graph (torch.cuda.CUDAGraph), and what does replay cost against eager calls?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion import _hip
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import image_util

for stereo in (False, True):
    p = SpectrogramParams(stereo=stereo)
    plan = _hip.get_plan(p, "cuda")
    tile = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(1, 512, 512, 3), dtype=np.uint8)).cuda()
    lut = plan.device_constant(("lut",), lambda: image_util.decode_lut(0.25, 30e6))
    C = 2 if stereo else 1
    out = torch.empty((1, 441 * 511, C), dtype=torch.int16, device="cuda")
    ws = plan.audio_from_image_workspace(1, stereo, 512)

    def call():
        return plan.audio_from_image(tile, stereo, lut, 32, 0.99, seed=7, out=out, workspace=ws, magnitude_hint=30e6)

    call(); torch.cuda.synchronize()
    ref = out.clone()
    def timeit(fn, n=50):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    eager = timeit(call)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            call()
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        replay = timeit(g.replay)
        print(f"stereo={stereo}: eager {eager:.3f} ms per tile, graph replay {replay:.3f} ms, same bytes: {same}")
    except Exception as exc:
        print(f"stereo={stereo}: capture failed: {type(exc).__name__}: {str(exc)[:300]}")
