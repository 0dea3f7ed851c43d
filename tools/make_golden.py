#!/usr/bin/env python3
"""
Generates tests/golden/image_codec_vectors.npz by IMPORTING the reference's own
riffusion/util/image_util.py and riffusion/spectrogram_params.py from /root/reference (both import
cleanly in the build container; torchaudio/pydub-dependent modules do not).  Run in the build
container only - /root/reference does not exist on the GPU box; the committed .npz travels instead.

Also documents where the other golden files come from:
  clip_2_*.wav / clip_2_*.png   copied verbatim from /root/reference/test/test_data/tired_traveler/
                                (the reference's own fixtures: genuine outputs of its forward path)
  og_beat_64.png                first 64 columns of /root/reference/seed_images/og_beat.png
"""
import importlib.util
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    # the reference's image_util imports `riffusion.spectrogram_params`: give it the reference's own
    import types

    pkg = types.ModuleType("riffusion")
    pkg.__path__ = [os.path.join(REF, "riffusion")]
    sys.modules["riffusion"] = pkg
    params_mod = load_ref("riffusion.spectrogram_params", "riffusion/spectrogram_params.py")
    util_pkg = types.ModuleType("riffusion.util")
    util_pkg.__path__ = [os.path.join(REF, "riffusion", "util")]
    sys.modules["riffusion.util"] = util_pkg
    iu = load_ref("riffusion.util.image_util", "riffusion/util/image_util.py")

    rng = np.random.default_rng(20240807)
    out = {}
    # encode vectors: heavy-tailed magnitudes, mono and stereo
    for tag, C in (("mono", 1), ("stereo", 2)):
        spec = (rng.random((C, 48, 37), dtype=np.float32) ** 5 * 4.6e7).astype(np.float32)
        img = np.array(iu.image_from_spectrogram(spec, power=0.25))
        out[f"enc_{tag}_in"] = spec
        out[f"enc_{tag}_out"] = img
        # decode vectors from that image and from a random one, both max_value choices the repo uses
        for mv_tag, mv in (("30e6", 30e6), ("max", float(spec.max()))):
            out[f"dec_{tag}_{mv_tag}"] = iu.spectrogram_from_image(
                Image.fromarray(img, mode="RGB"), power=0.25, stereo=(C == 2), max_value=mv
            )
    rnd = rng.integers(0, 256, size=(40, 23, 3), dtype=np.uint8)
    out["dec_rand_in"] = rnd
    out["dec_rand_mono"] = iu.spectrogram_from_image(Image.fromarray(rnd, mode="RGB"), 0.25, False, 30e6)
    out["dec_rand_stereo"] = iu.spectrogram_from_image(Image.fromarray(rnd, mode="RGB"), 0.25, True, 30e6)
    # all 256 decode values, and the encode result for ratios around every level boundary
    gray = np.arange(256, dtype=np.uint8).reshape(16, 16)
    out["dec_all256"] = iu.spectrogram_from_image(Image.fromarray(gray, mode="L"), 0.25, False, 30e6)
    ramp = np.linspace(0.0, 1.0, 65536, dtype=np.float32).reshape(1, 256, 256)
    out["enc_ramp_out"] = np.array(iu.image_from_spectrogram(ramp, power=0.25))[..., 0]
    # SpectrogramParams derived quantities + EXIF dict for a few parameter sets
    P = params_mod.SpectrogramParams
    rows = []
    for kw in ({}, {"stereo": True}, {"sample_rate": 48000}, {"sample_rate": 22050, "step_size_ms": 5},
               {"min_frequency": 20, "max_frequency": 20000, "num_frequencies": 256}):
        p = P(**kw)
        rows.append([p.n_fft, p.win_length, p.hop_length] + [float(v) for v in p.to_exif().values()])
    out["params_rows"] = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "image_codec_vectors.npz"), **out)
    print("wrote", os.path.join(OUT, "image_codec_vectors.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
