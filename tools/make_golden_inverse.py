#!/usr/bin/env python3
"""
Pins the INVERSE half of the oracle the day a real torchaudio is at hand.

The reference runs torchaudio 0.13.0 (cog.yaml:28-29); its InverseMelScale (SGD) and functional.griffinlim have no
golden output anywhere in the reference tree and torchaudio is not installed here, in the wheelhouse or on the GPU box, so
oracle/riffusion_oracle.py restates them (see its header and oracle/torchaudio_transcript.py).  With a torchaudio whose
InverseMelScale still takes `max_iter / tolerance_* / sgdargs` (<= 2.0) importable, this script runs the REAL modules,
constructed exactly as riffusion/spectrogram_converter.py:62-73 and :87-99 construct them, on seeded inputs and writes

    tests/golden/inverse_mel_sgd.npz   mel, seed, output of InverseMelScale.forward   (torch.manual_seed(seed) first)
    tests/golden/inverse_griffinlim.npz  magnitudes, seed, output of GriffinLim.forward

tests/test_oracle_golden.py::test_inverse_golden_vectors compares the oracle with them when they exist.  Without such a
torchaudio it prints why and exits 0 (nothing to generate; the inverse half stays "parity unpinned").
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main() -> int:
    try:
        import torchaudio
    except ImportError as e:
        print(f"torchaudio is not importable ({e}): no inverse golden vectors generated; the inverse half stays unpinned")
        return 0
    import inspect

    if "max_iter" not in inspect.signature(torchaudio.transforms.InverseMelScale.__init__).parameters:
        print(f"torchaudio {torchaudio.__version__} has the lstsq InverseMelScale (>= 2.1): the reference cannot run on it "
              "(spectrogram_converter.py:87-99 passes max_iter / tolerance_* / sgdargs); nothing generated")
        return 0
    n_fft, win, hop, sr, n_mels = 17640, 4410, 441, 44100, 512
    T = 24
    g = torch.Generator().manual_seed(20240807)
    mel = (torch.rand(1, n_mels, T, generator=g) ** 3 * 2e7).contiguous()
    inv = torchaudio.transforms.InverseMelScale(n_stft=n_fft // 2 + 1, n_mels=n_mels, sample_rate=sr, f_min=0, f_max=10000,
                                                max_iter=200, tolerance_loss=1e-5, tolerance_change=1e-8, sgdargs=None,
                                                norm=None, mel_scale="htk")
    torch.manual_seed(1234)
    lin = inv(mel)
    np.savez_compressed(os.path.join(OUT, "inverse_mel_sgd.npz"), mel=mel.numpy(), seed=1234, out=lin.numpy(),
                        torchaudio=torchaudio.__version__, torch=torch.__version__)
    mag = torch.rand(1, n_fft // 2 + 1, T, generator=g) * 1000
    gl = torchaudio.transforms.GriffinLim(n_fft=n_fft, n_iter=32, win_length=win, hop_length=hop, window_fn=torch.hann_window,
                                          power=1.0, wkwargs=None, momentum=0.99, length=None, rand_init=True)
    torch.manual_seed(4321)
    wave = gl(mag)
    np.savez_compressed(os.path.join(OUT, "inverse_griffinlim.npz"), mag=mag.numpy(), seed=4321, out=wave.numpy(),
                        torchaudio=torchaudio.__version__, torch=torch.__version__)
    print("wrote inverse_mel_sgd.npz and inverse_griffinlim.npz under", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
