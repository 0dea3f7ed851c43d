import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import numpy as np, torch
import riffusion_oracle as O
p = O.OracleParams()
rng = np.random.default_rng(0)
tile = rng.integers(0, 256, size=(512, 512, 3), dtype=np.uint8)
mel = torch.from_numpy(O.spectrogram_from_image_u8(tile, 0.25, False, 30e6))
g = torch.Generator().manual_seed(1)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    t0 = time.time(); lin = O.inverse_mel_scale_sgd(mel, p, generator=g); t1 = time.time()
    w = O.griffinlim(lin, p, generator=g); t2 = time.time()
    print(f"threads {th}: SGD-200 {t1-t0:.1f} s, GL-32 {t2-t1:.1f} s, total {t2-t0:.1f} s", flush=True)
    if t2 - t0 > 60: break
