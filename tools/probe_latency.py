"""Single-request latency: one call of SpectrogramImageConverter.audio_from_spectrogram_images for 1 / 2 / 4 / 8 / 16 tiles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))
import numpy as np, torch
from riffusion.spectrogram_image_converter import SpectrogramImageConverter
from riffusion.spectrogram_params import SpectrogramParams

for stereo in (False, True):
    conv = SpectrogramImageConverter(SpectrogramParams(stereo=stereo), "cuda")
    for n in (1, 2, 4, 8, 16):
        tiles = torch.from_numpy(np.random.default_rng(n).integers(0, 256, size=(n, 512, 512, 3), dtype=np.uint8)).cuda()
        for _ in range(3):
            conv.audio_from_spectrogram_images(tiles, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for r in range(reps):
            pcm = conv.audio_from_spectrogram_images(tiles, seed=r)   # includes the D2H copy of the int16 PCM
        dt = (time.perf_counter() - t0) / reps
        print(f"{'stereo' if stereo else 'mono'} tiles per call {n:2d}: {dt*1e3:6.2f} ms per call ({n/dt:7.1f} tiles/s) [RFX_GL_LATENCY_MODE={os.environ.get('RFX_GL_LATENCY_MODE','1')}]", flush=True)
