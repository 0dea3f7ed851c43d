"""Round 6 probe: the product entry point under a thread pool for a while - results equal the serial ones, memory does not grow."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "riffusion-hobby_amd"))
from multiprocessing.pool import ThreadPool
import numpy as np, torch
from riffusion.spectrogram_image_converter import SpectrogramImageConverter
from riffusion.spectrogram_params import SpectrogramParams

convs = {st: SpectrogramImageConverter(SpectrogramParams(stereo=st, num_griffin_lim_iters=4, max_mel_iters=40), device="cuda") for st in (False, True)}
rng = random.Random(1)
jobs = []
for i in range(400):
    n, w, st = rng.randint(1, 12), rng.choice((64, 96, 128)), rng.random() < 0.3
    jobs.append((i, n, w, st, rng.choice((1, 3, 64))))
tiles = {(n, w): np.random.default_rng(n * 1000 + w).integers(0, 256, size=(n, 512, w, 3), dtype=np.uint8) for _, n, w, _, _ in jobs}

def run(job):
    i, n, w, st, per = job
    return convs[st].audio_from_spectrogram_images(tiles[(n, w)], seed=i, tiles_per_call=per)

serial = {j[0]: run(j) for j in jobs[:60]}
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
t = time.time()
with ThreadPool(8) as pool:
    out = pool.map(run, jobs)
torch.cuda.synchronize()
bad = [j[0] for j in jobs[:60] if not np.array_equal(out[j[0]], serial[j[0]])]
print(f"{len(jobs)} calls on 8 threads in {time.time() - t:.1f} s; mismatches against the serial results: {bad}; "
      f"allocated {m0 >> 20} -> {torch.cuda.memory_allocated() >> 20} MiB, reserved {r0 >> 20} -> {torch.cuda.memory_reserved() >> 20} MiB; "
      f"arena buffers ever allocated: {[c.converter._plan().arena.allocations for c in convs.values()]}")
assert not bad
