"""
PCM container helpers.

`audio_from_waveform` keeps the reference's contract (riffusion/util/audio_util.py:13-36: joint
peak normalisation, truncation to int16, an `AudioSegment` back).  pydub is imported lazily: when it
is installed the functions return real `pydub.AudioSegment`s exactly like the reference; when it is
not (this image has no pydub) they return `PcmSegment`, a small read-only stand-in that offers the
part of the AudioSegment interface the spectrogram path touches (frame_rate, channels,
sample_width, split_to_mono, get_array_of_samples, set_channels, duration_seconds, export to wav).
"""
import io
import typing as T

import numpy as np


def _pydub():
    try:
        import pydub  # type: ignore

        return pydub
    except ImportError:
        return None


class PcmSegment:
    """int16 PCM, (samples, channels) interleaved - the subset of pydub.AudioSegment this path needs."""

    sample_width = 2

    def __init__(self, samples: np.ndarray, frame_rate: int):
        samples = np.asarray(samples)
        if samples.ndim == 1:
            samples = samples[:, None]
        if samples.dtype != np.int16:
            raise TypeError("PcmSegment holds int16 samples")
        self._data = np.ascontiguousarray(samples)
        self.frame_rate = int(frame_rate)

    @property
    def channels(self) -> int:
        return int(self._data.shape[1])

    @property
    def duration_seconds(self) -> float:
        return self._data.shape[0] / float(self.frame_rate)

    def __len__(self) -> int:  # milliseconds, like pydub
        return int(round(1000.0 * self.duration_seconds))

    def frame_count(self) -> float:
        return float(self._data.shape[0])

    def get_array_of_samples(self) -> np.ndarray:
        return self._data.reshape(-1)

    def split_to_mono(self) -> T.List["PcmSegment"]:
        return [PcmSegment(self._data[:, c].copy(), self.frame_rate) for c in range(self.channels)]

    def set_channels(self, channels: int) -> "PcmSegment":
        if channels == self.channels:
            return self
        if channels == 2 and self.channels == 1:
            return PcmSegment(np.repeat(self._data, 2, axis=1), self.frame_rate)
        if channels == 1 and self.channels == 2:
            # audioop.tomono(data, width, 0.5, 0.5): each side scaled by 0.5 (floor), then added
            left = np.floor(self._data[:, 0].astype(np.float64) * 0.5)
            right = np.floor(self._data[:, 1].astype(np.float64) * 0.5)
            return PcmSegment(np.clip(left + right, -32768, 32767).astype(np.int16), self.frame_rate)
        raise ValueError("PcmSegment.set_channels only converts between mono and stereo")

    # ---- the gain filters audio_util.apply_filters needs, restated from pydub 0.25 (AudioSegment.rms / dBFS /
    # max / apply_gain, effects.normalize) and CPython's audioop (rms = floor(sqrt(mean square)), mul =
    # floor of the clipped product).  pydub is not installed here: these are UNPINNED restatements.
    max_possible_amplitude = 32768.0

    @property
    def rms(self) -> int:
        x = self._data.astype(np.float64).reshape(-1)
        return int(np.sqrt(np.sum(x * x) / x.size)) if x.size else 0

    @property
    def dBFS(self) -> float:
        rms = self.rms
        return float("-inf") if rms == 0 else 20.0 * float(np.log10(rms / self.max_possible_amplitude))

    @property
    def max(self) -> int:
        return int(np.abs(self._data.astype(np.int32)).max()) if self._data.size else 0

    def apply_gain(self, volume_change: float) -> "PcmSegment":
        factor = 10.0 ** (float(volume_change) / 20.0)
        v = self._data.astype(np.float64) * factor
        v = np.where(v > 32767.0, 32767.0, np.where(v < -32767.0, -32768.0, v))  # audioop's fbound
        return PcmSegment(np.floor(v).astype(np.int16), self.frame_rate)

    def normalize(self, headroom: float = 0.1) -> "PcmSegment":
        peak = self.max
        if peak == 0:
            return self
        target_peak = self.max_possible_amplitude * 10.0 ** (-headroom / 20.0)
        return self.apply_gain(20.0 * float(np.log10(target_peak / peak)))

    def export(self, out_f: T.Any, format: str = "wav") -> T.Any:
        if format != "wav":
            raise NotImplementedError("PcmSegment exports wav only; install pydub + ffmpeg for other formats")
        from scipy.io import wavfile

        wavfile.write(out_f, self.frame_rate, self._data if self.channels > 1 else self._data[:, 0])
        return out_f

    @classmethod
    def from_wav(cls, path_or_file: T.Any) -> "PcmSegment":
        from scipy.io import wavfile

        rate, data = wavfile.read(path_or_file)
        if data.dtype != np.int16:
            raise NotImplementedError("only 16-bit PCM wav files are supported without pydub")
        return cls(data, rate)


def pcm16_from_waveform(samples: np.ndarray, normalize: bool = False) -> np.ndarray:
    """(channels, samples) float -> (samples, channels) int16 with the reference's arithmetic
    (audio_util.py:22-28): in-place scale by 32767 / max|x| over all channels, then truncation."""
    samples = np.array(samples, dtype=np.float32, copy=True)
    if normalize:
        # numpy 1.x evaluates python-int / float32-scalar in float64; the in-place multiply then
        # happens in float32.  Written out so that numpy 2's weak scalars give the same result.
        scale = np.float32(np.float64(np.iinfo(np.int16).max) / np.float64(np.max(np.abs(samples))))
        samples *= scale
    return np.ascontiguousarray(samples.transpose(1, 0).astype(np.int16))


def segment_from_pcm16(pcm: np.ndarray, sample_rate: int) -> T.Any:
    """(samples, channels) int16 -> pydub.AudioSegment when pydub exists, else PcmSegment."""
    pydub = _pydub()
    if pydub is None:
        return PcmSegment(pcm, sample_rate)
    from scipy.io import wavfile

    wav_bytes = io.BytesIO()
    wavfile.write(wav_bytes, sample_rate, pcm)
    wav_bytes.seek(0)
    return pydub.AudioSegment.from_wav(wav_bytes)


def audio_from_waveform(samples: np.ndarray, sample_rate: int, normalize: bool = False) -> T.Any:
    """(channels, samples) float array -> audio segment (reference audio_util.py:13-36)."""
    return segment_from_pcm16(pcm16_from_waveform(samples, normalize=normalize), sample_rate)


def apply_filters(segment: T.Any, compression: bool = False) -> T.Any:
    """Gain to -12 dBFS and peak normalisation with 0.1 dB headroom (reference audio_util.py:39-72): pydub /
    audioop integer filters on the host.  pydub segments go through pydub itself; PcmSegment carries a
    restatement of the two filters."""
    if isinstance(segment, PcmSegment):
        if compression:
            raise NotImplementedError("dynamic range compression needs pydub (the hot path calls compression=False)")
        return segment.apply_gain(-12 - segment.dBFS).normalize(headroom=0.1)
    pydub = _pydub()
    if pydub is None:
        raise NotImplementedError("apply_filters on a foreign segment type needs pydub")
    if compression:
        segment = pydub.effects.normalize(segment, headroom=0.1)
        segment = segment.apply_gain(-10 - segment.dBFS)
        segment = pydub.effects.compress_dynamic_range(segment, threshold=-20.0, ratio=4.0, attack=5.0, release=50.0)
    segment = segment.apply_gain(-12 - segment.dBFS)
    return pydub.effects.normalize(segment, headroom=0.1)


def stitch_segments(segments: T.Sequence[T.Any], crossfade_s: float) -> T.Any:
    """Concatenate with a crossfade (reference audio_util.py:75-85); pydub segments only."""
    crossfade_ms = int(crossfade_s * 1000)
    out = segments[0]
    for seg in segments[1:]:
        out = out.append(seg, crossfade=crossfade_ms)
    return out
