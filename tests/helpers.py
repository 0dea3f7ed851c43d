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


def mask_ill_conditioned_bins(O, mag: torch.Tensor, op, angles0: torch.Tensor, n_iter: int, rel: float = 1e-4, passes: int = 4):
    """
    Griffin-Lim's phase update normalises `a = rebuilt - m * tprev` bin by bin; where |a| is nearly zero the new phase is
    decided by rounding, on every implementation (profiles/r04_griffinlim_one_bin_events.txt: a 60 dB result after four
    iterations is ONE bin of one frame).  This finds those bins with the oracle in float64 - |a| < rel x the frame's largest
    |a| in any of the first n_iter iterations - and returns the magnitudes with them set to ZERO: a bin of magnitude zero
    contributes Z = 0 whatever its phase, so device and oracle can be compared on the same, now well-conditioned problem
    (zeroing changes the trajectory, so the scan is repeated until it comes back clean).
    Returns (masked magnitudes, number of masked (clip, bin, frame) entries).
    """
    mag = mag.clone()
    total = 0
    for _ in range(passes):
        bad = torch.zeros(mag.shape, dtype=torch.bool)

        def watch(k, a):
            small = a.abs() < rel * a.abs().amax(dim=-2, keepdim=True)
            bad.logical_or_(small & (mag > 0))

        O.griffinlim(mag, op, angles0=angles0, n_iter=n_iter, dtype=torch.float64, on_update=watch)
        n = int(bad.sum())
        if n == 0:
            break
        mag[bad] = 0.0
        total += n
    return mag, total
