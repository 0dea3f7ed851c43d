"""
Kill-test for VERDICT round 4, item 1: can InverseMelScale of batch k+1 hide under Griffin-Lim of batch k on a second stream?

  A  the two stages alone (B mono tiles of 512 frames): InverseMelScale (wave kernel, and the 128-VGPR group kernels) and Griffin-Lim 32
  B  a stand-in for a 128-VGPR one-wave SGD kernel (tools/ubench/coresident.hip, same instruction mix and total work as the wave kernel's
     64-tile launch) alone, as a persistent grid of 512 / 1024 / 2048 waves
  C  Griffin-Lim 32 with the stand-in running beside it on a second stream (both launch orders, Griffin-Lim stream at high priority):
     how much longer Griffin-Lim takes and when the stand-in finishes
  D  the real thing with the SHIPPED kernels: n steps serial on one stream against the two-stream pipeline
     (decode + InverseMelScale of step k+1 on stream A while Griffin-Lim + PCM of step k run on stream B)

    python tools/probe_overlap.py            -> one line per measurement on stdout
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from riffusion import _hip  # noqa: E402
from riffusion.spectrogram_params import SpectrogramParams  # noqa: E402
from riffusion.util import image_util  # noqa: E402

B = int(os.environ.get("B", 64))
T, ITERS = 512, 32
STEPS = int(os.environ.get("STEPS", 8))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
params = SpectrogramParams()
plan = _hip.get_plan(params, dev)
plan_grp = _hip.get_plan(params, dev, imel_form="groups")
rng = np.random.default_rng(20240807)
tiles = torch.from_numpy(rng.integers(0, 256, size=(B, 512, T, 3), dtype=np.uint8)).to(dev)
lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)
gl_ws = [torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, B, T), dtype=torch.uint8, device=dev) for _ in range(2)]

cores = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libcoresident_ubench.so"))
cores.cores_launch.restype = ctypes.c_int
cores.cores_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cores_out = torch.empty(4096 * 64, dtype=torch.float32, device=dev)
TOTAL_WAVE_STEPS = B * T * 200  # what one InverseMelScale launch of the wave kernel executes: one wave-step per frame and SGD step


def standin(grid, stream, lds=1024):
    rc = cores.cores_launch(cores_out.data_ptr(), grid, TOTAL_WAVE_STEPS // grid, lds, stream.cuda_stream)
    assert rc == 0, rc


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


mel = plan.image_decode(tiles, False, lut)
lin = plan.inverse_mel(mel, 1, seed=1)
torch.cuda.synchronize()

# ---- A: stages alone
print(f"A imel wave kernel alone      {timed(lambda: plan.inverse_mel(mel, 1, seed=2)):8.3f} ms")
print(f"A imel group kernels alone    {timed(lambda: plan_grp.inverse_mel(mel, 1, seed=2)):8.3f} ms")
gl_alone = timed(lambda: plan.griffinlim(lin, B, T, ITERS, 0.99, seed=3, workspace=gl_ws[0]))
print(f"A griffinlim-32 alone         {gl_alone:8.3f} ms")

# ---- B: stand-in alone
main = torch.cuda.current_stream(dev)
for grid in (512, 1024, 2048, 4096):
    print(f"B stand-in alone grid={grid:5d}  {timed(lambda: standin(grid, main)):8.3f} ms")

# ---- C: Griffin-Lim with the stand-in beside it
for prio_name, s_gl, s_x in (("gl high prio", torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)),
                             ("equal prio  ", torch.cuda.Stream(dev), torch.cuda.Stream(dev))):
    for grid in (512, 1024):
        for order in ("standin first", "gl first"):
            for lds in (1024,):
                res = []
                for rep in range(3):
                    torch.cuda.synchronize()
                    e = [ev() for _ in range(4)]
                    t0 = time.perf_counter()

                    def run_gl():
                        with torch.cuda.stream(s_gl):
                            e[0].record(s_gl)
                            plan.griffinlim(lin, B, T, ITERS, 0.99, seed=4, workspace=gl_ws[0])
                            e[1].record(s_gl)

                    def run_x():
                        with torch.cuda.stream(s_x):
                            e[2].record(s_x)
                            standin(grid, s_x, lds)
                            e[3].record(s_x)

                    if order == "gl first":
                        run_gl(); run_x()
                    else:
                        run_x(); run_gl()
                    torch.cuda.synchronize()
                    wall = (time.perf_counter() - t0) * 1e3
                    res.append((wall, e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])))
                w, g, x = min(res)
                print(f"C {prio_name} grid={grid:5d} {order:13s}: wall {w:7.3f} ms  griffinlim {g:7.3f} ms (+{g - gl_alone:6.3f})  stand-in {x:7.3f} ms")


# ---- D: the shipped kernels, serial against the two-stream pipeline
def serial(n, pl=plan):
    for k in range(n):
        m = pl.image_decode(tiles, False, lut)
        l = pl.inverse_mel(m, 1, seed=10 + k)
        w = plan.griffinlim(l, B, T, ITERS, 0.99, seed=11 + k, workspace=gl_ws[0])
        plan.pcm16(w, channels=1, normalize=True)


def pipelined(n, s_a, s_b, pl=plan):
    """stream A: decode + InverseMelScale of step k (at most one step ahead of B); stream B: Griffin-Lim + PCM of step k."""
    keep = []
    gl_done = []
    cur = torch.cuda.current_stream(dev)
    s_a.wait_stream(cur); s_b.wait_stream(cur)
    for k in range(n):
        with torch.cuda.stream(s_a):
            if k >= 2:
                s_a.wait_event(gl_done[k - 2])  # its |S| buffer generation is free again
            m = pl.image_decode(tiles, False, lut)
            l = pl.inverse_mel(m, 1, seed=10 + k)
            ready = torch.cuda.Event(); ready.record(s_a)
        with torch.cuda.stream(s_b):
            s_b.wait_event(ready)
            w = plan.griffinlim(l, B, T, ITERS, 0.99, seed=11 + k, workspace=gl_ws[k & 1])
            p = plan.pcm16(w, channels=1, normalize=True)
            d = torch.cuda.Event(); d.record(s_b); gl_done.append(d)
        keep.append((m, l, w, p))
    cur.wait_stream(s_a); cur.wait_stream(s_b)
    return keep


serial(2)
t_ser = timed(lambda: serial(STEPS), reps=2) / STEPS
print(f"D serial, wave kernel         {t_ser:8.3f} ms per step  ({B / t_ser * 1e3:7.1f} tiles/s)")
t_ser_g = timed(lambda: serial(STEPS, plan_grp), reps=2) / STEPS
print(f"D serial, group kernels       {t_ser_g:8.3f} ms per step")
for name, mk in (("A low / B high", lambda: (torch.cuda.Stream(dev, priority=0), torch.cuda.Stream(dev, priority=-1))),
                 ("equal prio    ", lambda: (torch.cuda.Stream(dev), torch.cuda.Stream(dev))),
                 ("A high / B low", lambda: (torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)))):
    for pl_name, pl in (("wave ", plan), ("group", plan_grp)):
        s_a, s_b = mk()
        pipelined(2, s_a, s_b, pl)
        t = timed(lambda: pipelined(STEPS, s_a, s_b, pl), reps=2) / STEPS
        print(f"D two streams {name} imel={pl_name}: {t:8.3f} ms per step  ({B / t * 1e3:7.1f} tiles/s, {100 * (t_ser / t - 1):+5.1f} % vs serial)")
