"""
PCM container helpers.

`audio_from_waveform` keeps the reference's contract (riffusion/util/audio_util.py:13-36: joint
peak normalisation, truncation to int16, an `AudioSegment` back).  pydub is imported lazily: when it
is installed the functions return real `pydub.AudioSegment`s exactly like the reference; when it is
not (this image has no pydub) they return `PcmSegment`, a small stand-in that offers the part of the
AudioSegment interface the spectrogram path and its callers touch (frame_rate, channels,
sample_width, split_to_mono, get_array_of_samples, set_channels, set_frame_rate, duration_seconds,
rms / dBFS / max, apply_gain, append with crossfade, overlay, export to wav).

pydub.AudioSegment is a thin layer over CPython's `audioop` C module (mul, rms, max, tomono,
tostereo, ratecv, add).  `PcmSegment` calls the same `audioop` functions when the interpreter still
ships the module (Python <= 3.12; the reference pins 3.9), so its integer arithmetic IS the
reference's; the numpy restatements below it are used only where audioop is gone, and
tests/test_host_logic.py pins them against audioop bit for bit.  What remains from memory is
pydub 0.25.1's glue around those calls (which audioop function, which factor), cited per method.
"""
import io
import typing as T

import numpy as np

try:  # the C module pydub itself calls; removed from CPython 3.13
    import audioop as _audioop  # type: ignore
except ImportError:  # pragma: no cover
    _audioop = None


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

    # ---- byte-level helpers ---------------------------------------------------------------------------------
    def _bytes(self) -> bytes:
        return self._data.tobytes()

    def _spawn(self, raw: bytes, channels: T.Optional[int] = None, frame_rate: T.Optional[int] = None) -> "PcmSegment":
        ch = self.channels if channels is None else channels
        return PcmSegment(np.frombuffer(raw, dtype=np.int16).reshape(-1, ch).copy(), self.frame_rate if frame_rate is None else frame_rate)

    @staticmethod
    def _mul_np(x: np.ndarray, factor: float) -> np.ndarray:
        """audioop.mul on int16: floor(clip(sample * factor)) with audioop's fbound (max 32767, min -32768)."""
        v = x.astype(np.float64) * float(factor)
        v = np.where(v > 32767.0, 32767.0, np.where(v < -32768.0, -32768.0, v))
        return np.floor(v).astype(np.int16)

    def set_channels(self, channels: int) -> "PcmSegment":
        """pydub AudioSegment.set_channels: mono -> stereo = audioop.tostereo(data, 2, 1, 1); stereo -> mono =
        audioop.tomono(data, 2, 0.5, 0.5)."""
        if channels == self.channels:
            return self
        if channels == 2 and self.channels == 1:
            if _audioop is not None:
                return self._spawn(_audioop.tostereo(self._bytes(), 2, 1, 1), channels=2)
            return PcmSegment(np.repeat(self._data, 2, axis=1), self.frame_rate)
        if channels == 1 and self.channels == 2:
            if _audioop is not None:
                return self._spawn(_audioop.tomono(self._bytes(), 2, 0.5, 0.5), channels=1)
            return PcmSegment(self._tomono_np(self._data), self.frame_rate)
        raise ValueError("PcmSegment.set_channels only converts between mono and stereo")

    @staticmethod
    def _tomono_np(x: np.ndarray) -> np.ndarray:
        """audioop.tomono(data, 2, 0.5, 0.5): floor(clip(l * 0.5 + r * 0.5)) in double."""
        v = x[:, 0].astype(np.float64) * 0.5 + x[:, 1].astype(np.float64) * 0.5
        return np.floor(np.clip(v, -32768.0, 32767.0)).astype(np.int16)

    def set_frame_rate(self, frame_rate: int) -> "PcmSegment":
        """pydub AudioSegment.set_frame_rate: audioop.ratecv(data, 2, channels, old, new, None) (linear
        interpolation resampler).  The reference's batch CLI calls it for files whose rate differs from the
        params' (cli.py:186-187)."""
        if int(frame_rate) == self.frame_rate:
            return self
        if _audioop is None:
            raise NotImplementedError("resampling needs CPython's audioop module (or pydub)")
        raw, _ = _audioop.ratecv(self._bytes(), 2, self.channels, self.frame_rate, int(frame_rate), None)
        return self._spawn(raw, frame_rate=int(frame_rate))

    # ---- the gain filters audio_util.apply_filters needs (pydub 0.25.1 AudioSegment.rms / dBFS / max /
    # apply_gain, effects.normalize) ---------------------------------------------------------------------
    max_possible_amplitude = 32768.0  # (2 ** 16) / 2

    @property
    def rms(self) -> int:
        """audioop.rms: floor(sqrt(sum(x^2) / n)) in double."""
        if _audioop is not None:
            return int(_audioop.rms(self._bytes(), 2))
        x = self._data.astype(np.float64).reshape(-1)
        return int(np.sqrt(np.sum(x * x) / x.size)) if x.size else 0

    @property
    def dBFS(self) -> float:
        """ratio_to_db(rms / max_possible_amplitude) = 20 * log(ratio, 10); -inf for silence."""
        import math

        rms = self.rms
        return float("-inf") if rms == 0 else 20.0 * math.log(rms / self.max_possible_amplitude, 10)

    @property
    def max(self) -> int:
        """audioop.max: largest absolute sample value."""
        if _audioop is not None:
            return int(_audioop.max(self._bytes(), 2))
        return int(np.abs(self._data.astype(np.int32)).max()) if self._data.size else 0

    def apply_gain(self, volume_change: float) -> "PcmSegment":
        """audioop.mul(data, 2, db_to_float(volume_change)), db_to_float(db) = 10 ** (db / 20)."""
        factor = 10 ** (float(volume_change) / 20)
        if _audioop is not None:
            return self._spawn(_audioop.mul(self._bytes(), 2, factor))
        return PcmSegment(self._mul_np(self._data, factor), self.frame_rate)

    def normalize(self, headroom: float = 0.1) -> "PcmSegment":
        """pydub.effects.normalize: boost so that the peak sits `headroom` dB below full scale."""
        import math

        peak = self.max
        if peak == 0:
            return self
        target_peak = self.max_possible_amplitude * (10 ** (-float(headroom) / 20))
        return self.apply_gain(20 * math.log(target_peak / peak, 10))

    # ---- joining clips (pydub AudioSegment.append / fade / overlay), used by stitch_segments / overlay_segments
    def _frames_of_ms(self, ms: float) -> int:
        return int(ms * (self.frame_rate / 1000.0))

    def _parse_position(self, val: float) -> int:
        """pydub AudioSegment._parse_position: milliseconds (negative = from the end, measured on the ROUNDED length in ms) -> frame index."""
        if val < 0:
            val = len(self) - abs(val)
        return int(self._frames_of_ms_f(len(self) if val == float("inf") else val))

    def _frames_of_ms_f(self, ms: float) -> float:
        return ms * (self.frame_rate / 1000.0)

    def _slice_ms(self, start_ms: T.Optional[float], end_ms: T.Optional[float]) -> "PcmSegment":
        """pydub AudioSegment.__getitem__(slice(start_ms, end_ms)): bounds clipped to len(self) - the length ROUNDED to whole
        milliseconds - so `seg[a:]` drops the frames past the last whole millisecond of a clip that is not a whole number of ms
        long, and a slice whose end lies past the data (the length rounded UP) is padded with silence (at most 2 ms, like pydub,
        which raises beyond that)."""
        length_ms = len(self)
        start = 0 if start_ms is None else min(start_ms, length_ms)
        end = length_ms if end_ms is None else min(end_ms, length_ms)
        a, b = self._parse_position(start), self._parse_position(end)
        data = self._data[a:b] if b > a else self._data[0:0]
        missing = max(0, b - a) - data.shape[0]
        if missing > 0:
            if missing > self._frames_of_ms_f(2):
                raise ValueError(f"slice is missing {missing} frames (pydub: TooManyMissingFrames)")
            data = np.concatenate([data, np.zeros((missing, self.channels), dtype=np.int16)])
        return PcmSegment(data, self.frame_rate)

    def _fade(self, to_gain: float = 0.0, from_gain: float = 0.0) -> "PcmSegment":
        """pydub AudioSegment.fade(to_gain, from_gain, start=0, end=inf), as append() calls it: the result is REBUILT from
        pieces - fades of more than 100 ms from the one-millisecond slices self[i] (gain stepped once per millisecond), shorter
        ones frame by frame - followed by self[len(self):], which is empty: frames past the last whole millisecond are dropped,
        exactly as pydub drops them.  Each step is an audioop.mul."""
        if to_gain == 0 and from_gain == 0:
            return self
        duration = len(self)
        from_power = 10 ** (float(from_gain) / 20)
        gain_delta = 10 ** (float(to_gain) / 20) - from_power
        mul = (lambda x, f: np.frombuffer(_audioop.mul(np.ascontiguousarray(x).tobytes(), 2, f), dtype=np.int16).reshape(x.shape)) \
            if _audioop is not None else self._mul_np
        pieces = []
        if duration > 100:
            scale_step = gain_delta / duration
            for i in range(duration):  # chunk i = self[i] = self[i : i + 1]
                pieces.append(mul(self._slice_ms(i, i + 1)._data, from_power + scale_step * i))
        else:
            fade_frames = self._frames_of_ms_f(duration)
            scale_step = gain_delta / fade_frames if fade_frames else 0.0
            for i in range(int(fade_frames)):
                pieces.append(mul(self._data[i : i + 1], from_power + scale_step * i))
        after = self._slice_ms(duration, None)._data  # self[end:] with end = len(self): empty
        pieces.append(mul(after, 10 ** (float(to_gain) / 20)) if to_gain != 0 else after)
        return PcmSegment(np.concatenate(pieces) if pieces else self._data[0:0], self.frame_rate)

    def overlay(self, other: "PcmSegment") -> "PcmSegment":
        """pydub AudioSegment.overlay(seg) at position 0 without looping: audioop.add over the overlap
        (saturating int16 sum); the result keeps this segment's length."""
        other = other.set_channels(self.channels).set_frame_rate(self.frame_rate)
        n = min(self._data.shape[0], other._data.shape[0])
        out = self._data.copy()
        if _audioop is not None:
            raw = _audioop.add(np.ascontiguousarray(self._data[:n]).tobytes(), np.ascontiguousarray(other._data[:n]).tobytes(), 2)
            out[:n] = np.frombuffer(raw, dtype=np.int16).reshape(n, self.channels)
        else:
            out[:n] = np.clip(self._data[:n].astype(np.int32) + other._data[:n].astype(np.int32), -32768, 32767).astype(np.int16)
        return PcmSegment(out, self.frame_rate)

    def append(self, seg: "PcmSegment", crossfade: int = 100) -> "PcmSegment":
        """pydub AudioSegment.append: the last `crossfade` ms of this segment faded to -120 dB are overlaid with the first
        `crossfade` ms of `seg` faded in from -120 dB."""
        seg = seg.set_channels(self.channels).set_frame_rate(self.frame_rate)
        if not crossfade:
            return PcmSegment(np.concatenate([self._data, seg._data]), self.frame_rate)
        if crossfade > len(self):
            raise ValueError(f"Crossfade is longer than the original AudioSegment ({crossfade}ms > {len(self)}ms)")
        if crossfade > len(seg):
            raise ValueError(f"Crossfade is longer than the appended AudioSegment ({crossfade}ms > {len(seg)}ms)")
        xf = self._slice_ms(-crossfade, None)._fade(to_gain=-120).overlay(seg._slice_ms(None, crossfade)._fade(from_gain=-120))
        return PcmSegment(
            np.concatenate([self._slice_ms(None, -crossfade)._data, xf._data, seg._slice_ms(crossfade, None)._data]), self.frame_rate
        )

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
    copy of the two filters on the same audioop calls."""
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
    """Concatenate with a crossfade (reference audio_util.py:75-85); pydub segments or PcmSegments."""
    crossfade_ms = int(crossfade_s * 1000)
    out = segments[0]
    for seg in segments[1:]:
        out = out.append(seg, crossfade=crossfade_ms)
    return out


def overlay_segments(segments: T.Sequence[T.Any]) -> T.Any:
    """Overlay segments on top of each other (reference audio_util.py:88-100)."""
    assert len(segments) > 0
    output: T.Any = None
    for segment in segments:
        output = segment if output is None else output.overlay(segment)
    return output
