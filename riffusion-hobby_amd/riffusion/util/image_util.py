"""
uint8 spectrogram image <-> float mel amplitudes.

Same functions and semantics as the reference's `riffusion/util/image_util.py:13-122`.  The power
curve has only finitely many cases on either side (256 pixel values when decoding, 256 output
levels when encoding), so it is tabulated with numpy's own float32 arithmetic:

* `decode_lut`       - the 256 float32 values numpy's chain `255-p, /255, **(1/power), *max_value`
                       produces (image_util.py:96-108);
* `encode_thresholds`- for each level v the smallest float32 ratio x/max that numpy's chain
                       `**power, *255, 255-, astype(uint8)` maps to a value <= v (image_util.py:32-41).

The HIP kernels (csrc/rfx_codec.hip) consume exactly these tables, which makes the device codec
bit-exact to the reference given the same float input; the numpy functions below use the same
tables so host and device agree byte for byte.
"""
import functools
import typing as T

import numpy as np
from PIL import Image

from riffusion.spectrogram_params import SpectrogramParams


@functools.lru_cache(maxsize=32)
def decode_lut(power: float = 0.25, max_value: float = 30e6) -> np.ndarray:
    p = np.arange(256, dtype=np.uint8).astype(np.float32)
    p = 255 - p
    p = p / 255
    p = np.power(p, 1 / power)
    p = p * max_value
    return np.ascontiguousarray(p, dtype=np.float32)


def _quantise_ratio(ratio: np.ndarray, power: float) -> np.ndarray:
    """numpy's own chain from image_util.py:32-41 applied to float32 ratios x/max."""
    d = np.power(ratio, power)
    d = d * 255
    d = 255 - d
    return d.astype(np.uint8)


@functools.lru_cache(maxsize=32)
def encode_thresholds(power: float = 0.25) -> np.ndarray:
    """thr[v], v = 0..254: smallest float32 r in [0, 1] with quantise(r) <= v (non-increasing in v)."""
    one = np.array([1.0], dtype=np.float32).view(np.uint32)[0]
    levels = np.arange(255, dtype=np.int64)
    lo = np.zeros(255, dtype=np.int64)  # invariant: quantise(lo-1) > v  (or lo == 0)
    hi = np.full(255, int(one), dtype=np.int64)  # invariant: quantise(hi) <= v  (quantise(1.0) == 0)
    while np.any(lo < hi):
        mid = (lo + hi) // 2
        q = _quantise_ratio(mid.astype(np.uint32).view(np.float32), power).astype(np.int64)
        ok = q <= levels
        hi = np.where(ok, mid, hi)
        lo = np.where(ok, lo, mid + 1)
    return np.ascontiguousarray(lo.astype(np.uint32).view(np.float32))


def quantise_spectrogram(spectrogram: np.ndarray, power: float = 0.25) -> np.ndarray:
    """(C, M, T) float32 -> (C, M, T) uint8: image_util.py:27-41 through the threshold table."""
    spectrogram = np.asarray(spectrogram, dtype=np.float32)
    ratio = spectrogram / np.max(spectrogram)
    thr = encode_thresholds(float(power))
    # q = number of thresholds strictly above the ratio; thr is non-increasing, search its reverse
    asc = thr[::-1]
    q = len(thr) - np.searchsorted(asc, ratio, side="right")
    return q.astype(np.uint8)


def image_from_spectrogram(spectrogram: np.ndarray, power: float = 0.25) -> Image.Image:
    """
    (channels, frequency, time) magnitudes -> RGB image (frequency, time), low frequencies at the
    bottom.  Mono is replicated into R=G=B, stereo goes to (0, left, right) - image_util.py:44-54.
    """
    data = quantise_spectrogram(spectrogram, power)
    if data.shape[0] == 1:
        rgb = np.repeat(data[0][:, :, None], 3, axis=2)
    elif data.shape[0] == 2:
        rgb = np.stack([np.zeros_like(data[0]), data[0], data[1]], axis=2)
    else:
        raise NotImplementedError(f"Unsupported number of channels: {data.shape[0]}")
    return Image.fromarray(np.ascontiguousarray(rgb[::-1]), mode="RGB")


def rgb_array_from_image(image: Image.Image) -> np.ndarray:
    """PIL image of any of the modes the reference accepts -> (H, W, 3) uint8 (image_util.py:81-82)."""
    if image.mode in ("P", "L"):
        image = image.convert("RGB")
    arr = np.array(image)
    if arr.ndim != 3 or arr.shape[2] < 3:
        raise ValueError(f"unsupported image mode {image.mode}")
    return np.ascontiguousarray(arr[:, :, :3])


def spectrogram_from_image(
    image: Image.Image,
    power: float = 0.25,
    stereo: bool = False,
    max_value: float = 30e6,
) -> np.ndarray:
    """RGB image -> (channels, frequency, time) float32 magnitudes (inverse of the above up to quantisation)."""
    rgb = rgb_array_from_image(image)[::-1]
    planes = rgb[:, :, [1, 2]] if stereo else rgb[:, :, 0:1]
    lut = decode_lut(float(power), float(max_value))
    return np.ascontiguousarray(lut[planes.transpose(2, 0, 1)])


def exif_from_image(pil_image: Image.Image) -> T.Dict[str, T.Any]:
    """EXIF of a spectrogram image as {tag name: value} (image_util.py:113-122)."""
    exif = pil_image.getexif()
    if exif is None or len(exif) == 0:
        return {}
    return {SpectrogramParams.ExifTags(key).name: val for key, val in exif.items()}
