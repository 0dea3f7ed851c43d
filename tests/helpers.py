"""Shared helpers of the test-suite (seeded synthetic inputs of SURVEY.md 8(d), SNR)."""
import numpy as np
import torch


def snr_db(ref: torch.Tensor, x: torch.Tensor) -> float:
    ref = ref.double().flatten()
    x = x.double().flatten()
    return float(10.0 * torch.log10(ref.pow(2).sum() / (ref - x).pow(2).sum().clamp_min(1e-300)))


def synthetic_tiles_u8(batch: int, height: int = 512, width: int = 512, seed: int = 20240807) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(batch, height, width, 3), dtype=np.uint8)


def synthetic_wave(batch: int, length: int, seed: int = 20240807) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal((batch, length)) * 8000).astype(np.float32))


def smooth_magnitudes(batch: int, n_stft: int, frames: int, seed: int = 7) -> torch.Tensor:
    """Magnitude spectrogram of a synthetic signal: gives Griffin-Lim something consistent to chew."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, n_stft, frames, generator=g) * 1000.0
