"""
Conversion parameters shared by the audio <-> spectrogram <-> image codecs.

API-compatible with the reference's `riffusion/spectrogram_params.py:8-115` (same field names,
defaults, derived sample counts, EXIF tag numbers and (de)serialisation behaviour); the class stays
frozen and therefore hashable, which the HIP layer uses as its plan-cache key.
"""
from __future__ import annotations

import dataclasses
import enum
import typing as T

# (field, EXIF tag, cast applied when writing) - reference tags at spectrogram_params.py:44-60
_EXIF_LAYOUT: T.Tuple[T.Tuple[str, int, T.Optional[type]], ...] = (
    ("sample_rate", 11000, None),
    ("stereo", 11005, None),
    ("step_size_ms", 11010, None),
    ("window_duration_ms", 11020, None),
    ("padded_duration_ms", 11030, None),
    ("num_frequencies", 11040, None),
    ("min_frequency", 11050, None),
    ("max_frequency", 11060, None),
    ("power_for_image", 11070, float),
)


def _ms_to_samples(duration_ms: float, sample_rate: int) -> int:
    # Same float expression and truncation as the reference (spectrogram_params.py:62-81)
    return int(duration_ms / 1000.0 * sample_rate)


@dataclasses.dataclass(frozen=True)
class SpectrogramParams:
    # audio layout
    stereo: bool = False

    # STFT geometry, in milliseconds at `sample_rate`
    sample_rate: int = 44100
    step_size_ms: int = 10
    window_duration_ms: int = 100
    padded_duration_ms: int = 400

    # mel projection
    num_frequencies: int = 512
    min_frequency: int = 0
    max_frequency: int = 10000
    mel_scale_norm: T.Optional[str] = None
    mel_scale_type: str = "htk"
    max_mel_iters: int = 200

    # phase reconstruction
    num_griffin_lim_iters: int = 32

    # uint8 image curve
    power_for_image: float = 0.25

    class ExifTags(enum.Enum):
        SAMPLE_RATE = 11000
        STEREO = 11005
        STEP_SIZE_MS = 11010
        WINDOW_DURATION_MS = 11020
        PADDED_DURATION_MS = 11030

        NUM_FREQUENCIES = 11040
        MIN_FREQUENCY = 11050
        MAX_FREQUENCY = 11060

        POWER_FOR_IMAGE = 11070
        MAX_VALUE = 11080

    @property
    def n_fft(self) -> int:
        return _ms_to_samples(self.padded_duration_ms, self.sample_rate)

    @property
    def win_length(self) -> int:
        return _ms_to_samples(self.window_duration_ms, self.sample_rate)

    @property
    def hop_length(self) -> int:
        return _ms_to_samples(self.step_size_ms, self.sample_rate)

    def to_exif(self) -> T.Dict[int, T.Any]:
        out: T.Dict[int, T.Any] = {}
        for field, tag, cast in _EXIF_LAYOUT:
            value = getattr(self, field)
            out[tag] = cast(value) if cast is not None else value
        return out

    @classmethod
    def from_exif(cls, exif: T.Mapping[int, T.Any]) -> "SpectrogramParams":
        # A missing tag raises KeyError exactly like the reference (callers catch it, cli.py:77-87)
        kwargs = {field: exif[tag] for field, tag, _ in _EXIF_LAYOUT}
        kwargs["stereo"] = bool(kwargs["stereo"])
        return cls(**kwargs)
