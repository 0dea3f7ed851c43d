"""
ctypes binding of librfx.so (include/rfx.h) and the per-`SpectrogramParams` plan cache.

This is the only place where Python touches the native library.  Tensors are handed over as raw
device pointers (`tensor.data_ptr()`) together with torch's current HIP stream; the library never
sees a torch type.  If the shared library is missing the import of the HIP path fails loudly -
there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import collections
import ctypes
import math
import os
import threading
import typing as T

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RFX_LIB_PATH") or os.path.join(os.path.dirname(_PKG_DIR), "librfx.so")

_lib: T.Optional[ctypes.CDLL] = None
_lib_lock = threading.Lock()

c_void_p, c_int, c_size_t, c_float, c_uint64 = (
    ctypes.c_void_p,
    ctypes.c_int,
    ctypes.c_size_t,
    ctypes.c_float,
    ctypes.c_uint64,
)


class RfxParams(ctypes.Structure):
    """rfx_params of include/rfx.h."""

    _fields_ = [
        ("sample_rate", ctypes.c_int32),
        ("n_fft", ctypes.c_int32),
        ("win_length", ctypes.c_int32),
        ("hop_length", ctypes.c_int32),
        ("n_mels", ctypes.c_int32),
        ("max_mel_iters", ctypes.c_int32),
    ]


class RfxPlanOptions(ctypes.Structure):
    """rfx_plan_options of include/rfx.h."""

    _fields_ = [("struct_size", ctypes.c_uint32), ("gl_form", ctypes.c_int32), ("gl_frames_per_slot", ctypes.c_int32),
                ("frame_engine", ctypes.c_int32), ("plan_layout", ctypes.c_int32), ("imel_form", ctypes.c_int32)]


class RfxCallOptions(ctypes.Structure):
    """rfx_call_options of include/rfx.h (round 6): per-call options of the inverse entry points."""

    _fields_ = [("struct_size", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("row_base", ctypes.c_uint64),
                ("magnitude_hint", ctypes.c_float), ("reserved", ctypes.c_float)]


def call_options(row_base: int = 0, magnitude_hint: float = 0.0) -> RfxCallOptions:
    if row_base < 0:
        raise ValueError("row_base must be >= 0")
    return RfxCallOptions(ctypes.sizeof(RfxCallOptions), 0, int(row_base), float(magnitude_hint), 0.0)


GL_FORMS = {"auto": 0, "runs": 1, "frames": 2}  # rfx_gl_form
FRAME_ENGINES = {"auto": 0, "generic": 1}       # rfx_frame_engine
PLAN_LAYOUTS = {"auto": 0, "generic": 1}        # rfx_plan_layout
IMEL_FORMS = {"auto": 0, "groups": 1}           # rfx_imel_form
GL_ENGINE_NAMES = {0: "specialised", 1: "generic", 2: "row-family"}  # rfx_plan_griffinlim_engine


class RfxError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol declared in include/rfx.h
SIGNATURES: T.Dict[str, T.Tuple[T.Any, T.List[T.Any]]] = {
    "rfx_last_error": (ctypes.c_char_p, []),
    "rfx_version": (c_int, []),
    "rfx_frame_stride": (c_int, []),
    "rfx_num_bins": (c_int, []),
    "rfx_plan_frame_stride": (c_int, [c_void_p]),
    "rfx_plan_is_generic": (c_int, [c_void_p]),
    "rfx_plan_griffinlim_engine": (c_int, [c_void_p]),
    "rfx_mel_scale_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "rfx_plan_create": (c_int, [ctypes.POINTER(RfxParams), c_void_p, c_void_p, c_int, ctypes.POINTER(c_void_p)]),
    "rfx_plan_create_ex": (c_int, [ctypes.POINTER(RfxParams), c_void_p, c_void_p, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "rfx_griffinlim_form": (c_int, [c_void_p, c_int, c_int]),
    "rfx_griffinlim_runs": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int]),
    "rfx_debug_run_start": (ctypes.c_int64, [ctypes.c_int64] * 6),
    "rfx_debug_gl_partition": (c_int, [c_int, c_int, c_int, c_void_p, c_int]),
    "rfx_debug_range_exponents": (c_int, [c_float, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "rfx_stft_frames": (c_int, [c_void_p, c_int]),
    "rfx_plan_imel_kernel": (c_int, [c_void_p]),
    "rfx_plan_imel_unit_form": (c_int, [c_void_p]),
    "rfx_plan_destroy": (c_int, [c_void_p]),
    "rfx_pack_magnitudes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rfx_pack_complex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rfx_unpack_complex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rfx_stft": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rfx_griffinlim_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "rfx_griffinlim_output_samples": (c_int, [c_void_p, c_int]),
    "rfx_griffinlim": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "rfx_griffinlim_timed": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    "rfx_griffinlim_ex": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p],
    ),
    "rfx_unpack_magnitudes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rfx_mel_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "rfx_mel_from_waveform": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rfx_mel_scale": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rfx_inverse_mel_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "rfx_inverse_mel": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_uint64, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "rfx_inverse_mel_ex": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_uint64, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    "rfx_image_decode_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rfx_image_encode_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rfx_audio_from_image_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "rfx_audio_from_image_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_uint64, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rfx_audio_from_image_u8_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_uint64, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "rfx_waveform_from_mel_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "rfx_waveform_from_mel_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_uint64, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "rfx_waveform_from_mel": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_uint64, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rfx_image_from_waveform_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "rfx_image_from_waveform": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rfx_pcm16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}


def library_path() -> str:
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    """Load librfx.so (built by `__graft_entry__.build()` / csrc/build.sh).  Raises if absent."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RfxError(
                    f"{_LIB_PATH} not found: build the HIP library first "
                    "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                )
            lib = ctypes.CDLL(_LIB_PATH)
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so does not export it
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = load_library().rfx_last_error()
        raise RfxError(f"librfx error {status}: {msg.decode() if msg else '?'}")


def current_stream(device: T.Optional[torch.device] = None) -> int:
    """torch's current stream ON `device` (not on the calling thread's current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def resolve_device(device: T.Union[str, torch.device]) -> torch.device:
    """'cuda' -> the indexed device it means for the calling thread right now."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RfxError("the HIP path runs on the GPU only (device 'cuda'); there is no CPU implementation")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


# ------------------------------------------------------------------------------------------------
# Host-side constants, built with the same torch ops torchaudio uses so that they are bit-identical
# to the buffers of the reference's modules (spectrogram_converter.py:47-99)
# ------------------------------------------------------------------------------------------------


def hann_window(win_length: int) -> torch.Tensor:
    return torch.hann_window(win_length, periodic=True, dtype=torch.float32)


def _hz_to_mel(freq: float, mel_scale: str) -> float:
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    # slaney
    f_sp = 200.0 / 3
    if freq >= 1000.0:
        return 1000.0 / f_sp + math.log(freq / 1000.0) / (math.log(6.4) / 27.0)
    return freq / f_sp


def _mel_to_hz(mels: torch.Tensor, mel_scale: str) -> torch.Tensor:
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    is_log = mels >= min_log_mel
    freqs[is_log] = 1000.0 * torch.exp(logstep * (mels[is_log] - min_log_mel))
    return freqs


def mel_filterbank(
    n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int, norm: T.Optional[str], mel_scale: str
) -> torch.Tensor:
    """Triangular mel filterbank (n_freqs, n_mels), the buffer of torchaudio's MelScale/InverseMelScale."""
    if norm is not None and norm != "slaney":
        raise ValueError('norm must be one of None or "slaney"')
    if mel_scale not in ("htk", "slaney"):
        raise ValueError('mel_scale should be one of "htk" or "slaney".')
    grid = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_pts = torch.linspace(_hz_to_mel(f_min, mel_scale), _hz_to_mel(f_max, mel_scale), n_mels + 2)
    hz_pts = _mel_to_hz(mel_pts, mel_scale)
    widths = hz_pts[1:] - hz_pts[:-1]
    dist = hz_pts.unsqueeze(0) - grid.unsqueeze(1)
    falling = (-1.0 * dist[:, :-2]) / widths[:-1]
    rising = dist[:, 2:] / widths[1:]
    fb = torch.max(torch.zeros(1), torch.min(falling, rising))
    if norm == "slaney":
        fb = fb * (2.0 / (hz_pts[2 : n_mels + 2] - hz_pts[:n_mels])).unsqueeze(0)
    return fb.to(torch.float32).contiguous()


class WorkspaceArena:
    """
    Reusable device workspaces of ONE plan (round 6).  Every C entry point takes a caller-provided scratch buffer (1.7 GB for
    64 mono tiles); until round 5 each Python call asked torch's caching allocator for a fresh one, and the allocator - which
    splits a freed 1.7 GB block as soon as a smaller request comes by - answered one call in twenty with a 20 ms hipMalloc.
    The arena keeps the buffers instead:

    * `take(nbytes, stream)` CHECKS OUT an idle buffer of at least `nbytes` that was last used on the same HIP stream (kernels
      of consecutive calls on one stream are ordered, so the hand-over needs no event), or allocates one - grow-only: a buffer
      that is too small is dropped in favour of the bigger one, sizes are rounded up to 32 MiB;
    * `give(buf, stream)` returns it when the call has QUEUED its kernels.
    A buffer that is checked out belongs to one host call: two threads of a pool that share a converter - and torch's default
    stream - (reference cli.py:172-204) get two buffers.  The lock guards the free lists only (microseconds); allocation happens
    outside it.  At most `max_idle` idle buffers are kept per plan (least recently used dropped first); `clear()` / the plan's
    `close()` release them.  The buffers are torch tensors, so `torch.cuda.memory_stats()` sees them.
    """

    GRANULE = 32 << 20

    def __init__(self, device: torch.device, max_idle: int = 4):
        self.device, self.max_idle = device, max_idle
        self._lock = threading.Lock()
        self._idle: "collections.OrderedDict[int, T.Tuple[int, torch.Tensor]]" = collections.OrderedDict()  # id -> (stream, buffer), LRU first
        self.allocations = 0  # buffers ever allocated (tests: steady state allocates nothing)

    def take(self, nbytes: int, stream: int) -> torch.Tensor:
        dropped = None
        with self._lock:
            best = None
            for key, (st, buf) in self._idle.items():
                if st == stream and buf.numel() >= nbytes and (best is None or buf.numel() < self._idle[best][1].numel()):
                    best = key
            if best is not None:
                return self._idle.pop(best)[1]
            for key, (st, buf) in self._idle.items():  # grow: the too-small buffer of this stream makes room for its successor
                if st == stream:
                    dropped = self._idle.pop(key)[1]
                    break
            self.allocations += 1
        del dropped  # back to torch's allocator (it was allocated and used on `stream` only: stream-ordered reuse is safe)
        size = max(self.GRANULE, -(-int(nbytes) // self.GRANULE) * self.GRANULE)
        return torch.empty(size, dtype=torch.uint8, device=self.device)

    def give(self, buf: torch.Tensor, stream: int) -> None:
        with self._lock:
            self._idle[id(buf)] = (stream, buf)
            while len(self._idle) > self.max_idle:
                self._idle.popitem(last=False)

    def clear(self) -> None:
        with self._lock:
            self._idle.clear()

    def idle_bytes(self) -> int:
        with self._lock:
            return sum(buf.numel() for _, buf in self._idle.values())


class _Borrowed:
    """`with plan._workspace(n) as ws:` - a checked-out arena buffer, returned when the call's kernels have been queued."""

    __slots__ = ("arena", "stream", "buf")

    def __init__(self, arena: WorkspaceArena, nbytes: int, stream: int):
        self.arena, self.stream = arena, stream
        self.buf = arena.take(nbytes, stream)

    def __enter__(self) -> torch.Tensor:
        return self.buf

    def __exit__(self, *exc: T.Any) -> bool:
        self.arena.give(self.buf, self.stream)
        return False


class Plan:
    """Owns one rfx_plan (device constants for one parameter set on one device)."""

    def __init__(self, params: T.Any, device: torch.device, gl_form: str = "auto", frame_engine: str = "auto",
                 plan_layout: str = "auto", imel_form: str = "auto"):
        self.lib = load_library()
        if gl_form not in GL_FORMS:
            raise ValueError(f"gl_form must be one of {sorted(GL_FORMS)}, got {gl_form!r}")
        if frame_engine not in FRAME_ENGINES:
            raise ValueError(f"frame_engine must be one of {sorted(FRAME_ENGINES)}, got {frame_engine!r}")
        if plan_layout not in PLAN_LAYOUTS:
            raise ValueError(f"plan_layout must be one of {sorted(PLAN_LAYOUTS)}, got {plan_layout!r}")
        if imel_form not in IMEL_FORMS:
            raise ValueError(f"imel_form must be one of {sorted(IMEL_FORMS)}, got {imel_form!r}")
        self.gl_form = gl_form
        self.device = device
        self.n_fft, self.win_length, self.hop_length = params.n_fft, params.win_length, params.hop_length
        self.n_stft = self.n_fft // 2 + 1
        self.n_mels = params.num_frequencies
        self.window = hann_window(self.win_length)
        self.melfb = mel_filterbank(
            self.n_stft,
            float(params.min_frequency),
            float(params.max_frequency),
            self.n_mels,
            params.sample_rate,
            params.mel_scale_norm,
            params.mel_scale_type,
        )
        cp = RfxParams(params.sample_rate, self.n_fft, self.win_length, self.hop_length, self.n_mels, params.max_mel_iters)
        handle = c_void_p()
        self.device = device = resolve_device(device)
        opt = RfxPlanOptions(ctypes.sizeof(RfxPlanOptions), GL_FORMS[gl_form], 0, FRAME_ENGINES[frame_engine], PLAN_LAYOUTS[plan_layout],
                             IMEL_FORMS[imel_form])
        check(
            self.lib.rfx_plan_create_ex(
                ctypes.byref(cp), self.window.data_ptr(), self.melfb.data_ptr(), device.index, ctypes.byref(opt), ctypes.byref(handle)
            )
        )
        self.handle = handle
        self.frame_stride = self.lib.rfx_plan_frame_stride(self.handle)
        self.generic = bool(self.lib.rfx_plan_is_generic(self.handle))
        self.griffinlim_engine = GL_ENGINE_NAMES[self.lib.rfx_plan_griffinlim_engine(self.handle)]
        self.arena = WorkspaceArena(device, max_idle=max(1, int(os.environ.get("RFX_ARENA_IDLE", "4"))))
        self._consts: "collections.OrderedDict[T.Any, torch.Tensor]" = collections.OrderedDict()
        self._consts_lock = threading.Lock()

    def _workspace(self, nbytes: int) -> _Borrowed:
        """A scratch buffer of at least `nbytes` from the plan's arena for the duration of one call on the current stream."""
        return _Borrowed(self.arena, nbytes, self._stream())

    def release_workspaces(self) -> None:
        """Hands the idle scratch buffers back to torch's allocator (they are re-made on demand)."""
        self.arena.clear()

    def device_constant(self, key: T.Any, build: T.Callable[[], T.Any]) -> torch.Tensor:
        """Small host-built tables (decode LUT, encoder thresholds) uploaded ONCE per plan and key: an upload from pageable
        memory per call is a synchronous copy - the host would wait for the kernels queued before it, call after call."""
        with self._consts_lock:
            t = self._consts.get(key)
            if t is not None:
                self._consts.move_to_end(key)
                return t
        t = torch.as_tensor(build()).to(self.device)
        with self._consts_lock:
            self._consts[key] = t
            while len(self._consts) > 64:
                self._consts.popitem(last=False)
        return t

    def close(self) -> None:
        """Releases the plan's device memory now (it is released anyway when the last reference goes)."""
        arena = getattr(self, "arena", None)
        if arena is not None:
            arena.clear()
        handle, self.handle = getattr(self, "handle", None), None
        if handle:
            self.lib.rfx_plan_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thin typed wrappers -----------------------------------------------------------------
    def _chk(self, t: torch.Tensor, dtype: T.Optional[torch.dtype] = None) -> torch.Tensor:
        """Tensors must live on THIS plan's GPU: its constant tables do, and kernels are queued on that device's stream."""
        if t.device != self.device:
            raise RfxError(f"tensor on {t.device} handed to a plan that lives on {self.device}")
        return (t if dtype is None else t.to(dtype)).contiguous()

    def _stream(self) -> int:
        return current_stream(self.device)

    def pack_magnitudes(self, lin_bft: torch.Tensor) -> torch.Tensor:
        lin_bft = self._chk(lin_bft, torch.float32)
        B, F, Tn = lin_bft.shape
        if F != self.n_stft:
            raise ValueError(f"expected {self.n_stft} linear bins, got {F}")
        out = torch.zeros((B * Tn, self.frame_stride), dtype=torch.float32, device=lin_bft.device)
        check(self.lib.rfx_pack_magnitudes(self.handle, lin_bft.data_ptr(), B, Tn, out.data_ptr(), self._stream()))
        return out

    def pack_complex(self, x_bft: torch.Tensor) -> torch.Tensor:
        x_bft = self._chk(x_bft, torch.complex64)
        B, F, Tn = x_bft.shape
        out = torch.zeros((B * Tn, self.frame_stride), dtype=torch.complex64, device=x_bft.device)
        check(self.lib.rfx_pack_complex(self.handle, x_bft.data_ptr(), B, Tn, out.data_ptr(), self._stream()))
        return out

    def unpack_complex(self, slots: torch.Tensor, B: int, Tn: int) -> torch.Tensor:
        slots = self._chk(slots, torch.complex64)
        out = torch.empty((B, self.n_stft, Tn), dtype=torch.complex64, device=slots.device)
        check(self.lib.rfx_unpack_complex(self.handle, slots.data_ptr(), B, Tn, out.data_ptr(), self._stream()))
        return out

    def stft(self, wave: torch.Tensor, want_mag: bool, want_spec: bool):
        wave = self._chk(wave, torch.float32)
        B, Lw = wave.shape
        if Lw <= self.n_fft // 2:
            # same condition under which torch.stft(pad_mode="reflect") raises in the reference
            raise RuntimeError(
                f"Argument #4: Padding size should be less than the corresponding input dimension, "
                f"but got: padding ({self.n_fft // 2}, {self.n_fft // 2}) at dimension 2 of input {list(wave.shape)}"
            )
        Tn = self.lib.rfx_stft_frames(self.handle, Lw)  # torch.stft's count: 1 + (Lw + 2*(n_fft//2) - n_fft) // hop
        mag = torch.empty((B * Tn, self.frame_stride), dtype=torch.float32, device=wave.device) if want_mag else None
        spec = torch.empty((B * Tn, self.frame_stride), dtype=torch.complex64, device=wave.device) if want_spec else None
        check(
            self.lib.rfx_stft(
                self.handle,
                wave.data_ptr(),
                B,
                Lw,
                mag.data_ptr() if mag is not None else None,
                spec.data_ptr() if spec is not None else None,
                self._stream(),
            )
        )
        return mag, spec, Tn

    def griffinlim(
        self,
        mag_slots: torch.Tensor,
        B: int,
        Tn: int,
        n_iter: int,
        momentum: float = 0.99,
        angles0_slots: T.Optional[torch.Tensor] = None,
        seed: int = 0,
        workspace: T.Optional[torch.Tensor] = None,
        launch_ms: T.Optional[T.Any] = None,
        row_base: int = 0,
        magnitude_hint: float = 0.0,
    ) -> torch.Tensor:
        """GriffinLim on magnitudes in slot layout -> (B, samples).  `row_base`: index of the call's first row in the caller's
        whole batch (the random phases of row r are drawn from (seed, row_base + r): chunked and sharded batches get the starts
        of the single call); `magnitude_hint`: an upper bound of the magnitudes if the caller knows one (rfx_call_options)."""
        mag_slots = self._chk(mag_slots, torch.float32)
        if angles0_slots is not None:
            angles0_slots = self._chk(angles0_slots, torch.complex64)
        if mag_slots.numel() < B * Tn * self.frame_stride:
            raise ValueError(f"magnitude slots hold {mag_slots.numel()} values, {B} x {Tn} frames need {B * Tn * self.frame_stride}")
        need = self.lib.rfx_griffinlim_workspace_bytes(self.handle, B, Tn)
        if workspace is not None:
            workspace = self._chk(workspace)
        if workspace is None or workspace.numel() < need:  # no (or too small a) caller-owned workspace: the plan's arena
            with self._workspace(need) as ws:
                return self.griffinlim(mag_slots, B, Tn, n_iter, momentum, angles0_slots, seed, ws, launch_ms, row_base, magnitude_hint)
        out = torch.empty((B, self.lib.rfx_griffinlim_output_samples(self.handle, Tn)), dtype=torch.float32, device=mag_slots.device)
        opt = call_options(row_base, magnitude_hint)
        check(
            self.lib.rfx_griffinlim_ex(
                self.handle,
                mag_slots.data_ptr(),
                angles0_slots.data_ptr() if angles0_slots is not None else None,
                seed & 0xFFFFFFFFFFFFFFFF,
                B,
                Tn,
                n_iter,
                momentum,
                out.data_ptr(),
                workspace.data_ptr(),
                workspace.numel(),
                self._stream(),
                ctypes.byref(opt),
                ctypes.cast(launch_ms, c_void_p) if launch_ms is not None else None,  # ctypes float array of n_iter + 1 entries, filled after a stream sync
            )
        )
        return out

    def unpack_magnitudes(self, slots: torch.Tensor, B: int, Tn: int) -> torch.Tensor:
        slots = self._chk(slots, torch.float32)
        out = torch.empty((B, self.n_stft, Tn), dtype=torch.float32, device=slots.device)
        check(self.lib.rfx_unpack_magnitudes(self.handle, slots.data_ptr(), B, Tn, out.data_ptr(), self._stream()))
        return out

    def mel_from_waveform(self, wave: torch.Tensor) -> torch.Tensor:
        """spectrogram_converter.py:165-185 on the device: (B, Lw) -> (B, n_mels, T)."""
        wave = self._chk(wave, torch.float32)
        B, Lw = wave.shape
        if Lw <= self.n_fft // 2:
            raise RuntimeError(
                f"Argument #4: Padding size should be less than the corresponding input dimension, "
                f"but got: padding ({self.n_fft // 2}, {self.n_fft // 2}) at dimension 2 of input {list(wave.shape)}"
            )
        Tn = self.lib.rfx_stft_frames(self.handle, Lw)
        need = self.lib.rfx_mel_workspace_bytes(self.handle, B, Lw)
        out = torch.empty((B, self.n_mels, Tn), dtype=torch.float32, device=wave.device)
        with self._workspace(need) as ws:
            check(
                self.lib.rfx_mel_from_waveform(
                    self.handle, wave.data_ptr(), B, Lw, out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()
                )
            )
        return out

    def mel_scale(self, lin_bft: torch.Tensor) -> torch.Tensor:
        """MelScale.forward on (B, n_stft, T) magnitudes through the MFMA projection."""
        lin_bft = self._chk(lin_bft, torch.float32)
        B, F, Tn = lin_bft.shape
        if F != self.n_stft:
            raise ValueError(f"expected {self.n_stft} linear bins, got {F}")
        out = torch.empty((B, self.n_mels, Tn), dtype=torch.float32, device=lin_bft.device)
        with self._workspace(self.lib.rfx_mel_scale_workspace_bytes(self.handle, B, Tn)) as ws:
            check(self.lib.rfx_mel_scale(self.handle, lin_bft.data_ptr(), B, Tn, out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return out

    def inverse_mel(
        self,
        mel: torch.Tensor,
        channels_per_clip: int,
        spec0: T.Optional[torch.Tensor] = None,
        seed: int = 0,
        row_base: int = 0,
        magnitude_hint: float = 0.0,
    ) -> torch.Tensor:
        """InverseMelScale (SGD): (B, n_mels, T) -> linear magnitudes in slot layout (B*T, stride).  `row_base`,
        `magnitude_hint`: as in `griffinlim` (row_base must be a multiple of channels_per_clip: clips are not split)."""
        mel = self._chk(mel, torch.float32)
        B, M, Tn = mel.shape
        if M != self.n_mels:
            raise ValueError(f"Expected an input with {self.n_mels} mel bins. Found: {M}")  # torchaudio's message
        if spec0 is not None:
            spec0 = self._chk(spec0, torch.float32)
            if tuple(spec0.shape) != (B, Tn, self.n_stft):
                raise ValueError(f"spec0 must be (B, T, n_stft) = {(B, Tn, self.n_stft)}, got {tuple(spec0.shape)}")
        need = self.lib.rfx_inverse_mel_workspace_bytes(self.handle, B, Tn)
        out = torch.empty((B * Tn, self.frame_stride), dtype=torch.float32, device=mel.device)
        opt = call_options(row_base, magnitude_hint)
        with self._workspace(need) as ws:
            check(
                self.lib.rfx_inverse_mel_ex(
                    self.handle,
                    mel.data_ptr(),
                    B,
                    Tn,
                    channels_per_clip,
                    spec0.data_ptr() if spec0 is not None else None,
                    seed & 0xFFFFFFFFFFFFFFFF,
                    out.data_ptr(),
                    ws.data_ptr(),
                    ws.numel(),
                    self._stream(),
                    ctypes.byref(opt),
                )
            )
        return out

    # ---- codecs (no plan state needed, kept here for one binding site) -------------------------
    def image_decode(self, img_u8: torch.Tensor, stereo: bool, lut: torch.Tensor) -> torch.Tensor:
        """(N, H, W, 3) uint8 -> (N*C, H, W) float32."""
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
            raise ValueError("expected (N, H, W, 3) uint8 images")
        img_u8 = self._chk(img_u8)
        lut = self._chk(lut, torch.float32)
        N, H, W, _ = img_u8.shape
        C = 2 if stereo else 1
        out = torch.empty((N * C, H, W), dtype=torch.float32, device=img_u8.device)
        check(self.lib.rfx_image_decode_u8(img_u8.data_ptr(), N, H, W, int(stereo), lut.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def image_encode(self, mel: torch.Tensor, stereo: bool, thresholds: torch.Tensor):
        """(N*C, M, T) float32 -> ((N, M, T, 3) uint8, per-clip max (N,))."""
        mel = self._chk(mel, torch.float32)
        thresholds = self._chk(thresholds, torch.float32)
        C = 2 if stereo else 1
        NC, M, Tn = mel.shape
        if NC % C:
            raise ValueError("batch must be a multiple of the channel count")
        N = NC // C
        img = torch.empty((N, M, Tn, 3), dtype=torch.uint8, device=mel.device)
        mx = torch.empty((N,), dtype=torch.float32, device=mel.device)
        check(self.lib.rfx_image_encode_u8(mel.data_ptr(), N, M, Tn, int(stereo), thresholds.data_ptr(), mx.data_ptr(), img.data_ptr(), self._stream()))
        return img, mx

    def waveform_from_mel(self, mel: torch.Tensor, channels_per_clip: int, n_iter: int, momentum: float = 0.99, seed: int = 0,
                          row_base: int = 0, magnitude_hint: float = 0.0) -> torch.Tensor:
        """spectrogram_converter.py:187-204 in one call: (B, n_mels, T) -> (B, hop * (T - 1)); `inverse_mel` (seed) + `griffinlim`
        (seed + 1), same bits, the linear magnitudes stay in the workspace."""
        mel = self._chk(mel, torch.float32)
        B, M, Tn = mel.shape
        if M != self.n_mels:
            raise ValueError(f"Expected an input with {self.n_mels} mel bins. Found: {M}")  # torchaudio's message
        out = torch.empty((B, self.lib.rfx_griffinlim_output_samples(self.handle, Tn)), dtype=torch.float32, device=mel.device)
        opt = call_options(row_base, magnitude_hint)
        with self._workspace(self.lib.rfx_waveform_from_mel_workspace_bytes(self.handle, B, Tn)) as ws:
            check(self.lib.rfx_waveform_from_mel_ex(self.handle, mel.data_ptr(), B, Tn, channels_per_clip, seed & 0xFFFFFFFFFFFFFFFF, n_iter, momentum,
                                                    out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream(), ctypes.byref(opt)))
        return out

    def audio_from_image_workspace(self, N: int, stereo: bool, Tn: int) -> torch.Tensor:
        """A caller-owned workspace for `audio_from_image` calls of up to N images of Tn frames (optional since round 6: without
        one the call borrows a buffer from the plan's arena)."""
        return torch.empty(self.lib.rfx_audio_from_image_workspace_bytes(self.handle, N, int(stereo), Tn), dtype=torch.uint8, device=self.device)

    def audio_from_image(self, img: torch.Tensor, stereo: bool, lut: torch.Tensor, n_iter: int, momentum: float = 0.99, seed: int = 0,
                         normalize: bool = True, out: T.Optional[torch.Tensor] = None, workspace: T.Optional[torch.Tensor] = None,
                         clip_base: int = 0, magnitude_hint: float = 0.0):
        """spectrogram_image_converter.py:54-91 on the device in one call: (N, n_mels, T, 3) uint8 -> ((N, L, C) int16, per-clip peak (N,));
        `image_decode` + `waveform_from_mel` (clips of C rows) + `pcm16`, same bytes.  `out` as in `pcm16`.  `clip_base`: index of
        the call's first image in the caller's whole batch (row_base = clip_base * C); `magnitude_hint`: the image path's
        max_value (the largest entry of `lut`)."""
        if img.dtype != torch.uint8 or img.dim() != 4:
            raise ValueError("expected (N, H, W, 3) uint8 images")
        img = self._chk(img)
        lut = self._chk(lut, torch.float32)
        N, H, W, ch = img.shape
        if ch != 3 or H != self.n_mels:
            raise ValueError(f"expected (N, {self.n_mels}, T, 3) uint8 images, got {tuple(img.shape)}")
        C = 2 if stereo else 1
        L = self.lib.rfx_griffinlim_output_samples(self.handle, W)
        if out is not None:
            if out.device != self.device or out.dtype != torch.int16 or tuple(out.shape) != (N, L, C) or not out.is_contiguous():
                raise ValueError(f"out must be a contiguous int16 tensor of shape {(N, L, C)} on {self.device}")
            pcm = out
        else:
            pcm = torch.empty((N, L, C), dtype=torch.int16, device=img.device)
        need = self.lib.rfx_audio_from_image_workspace_bytes(self.handle, N, int(stereo), W)
        ws = self._chk(workspace) if workspace is not None else None
        if ws is None or ws.numel() < need:
            with self._workspace(need) as borrowed:
                return self.audio_from_image(img, stereo, lut, n_iter, momentum, seed, normalize, out=pcm, workspace=borrowed,
                                             clip_base=clip_base, magnitude_hint=magnitude_hint)
        peak = torch.zeros((N,), dtype=torch.float32, device=img.device)
        opt = call_options(clip_base * C, magnitude_hint)
        check(self.lib.rfx_audio_from_image_u8_ex(self.handle, img.data_ptr(), N, W, int(stereo), lut.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, n_iter, momentum,
                                                  int(normalize), peak.data_ptr(), pcm.data_ptr(), ws.data_ptr(), ws.numel(), self._stream(), ctypes.byref(opt)))
        return pcm, peak

    def image_from_waveform(self, wave: torch.Tensor, stereo: bool, thresholds: torch.Tensor):
        """spectrogram_image_converter.py:30-51 on the device: (N*C, Lw) float32 -> ((N, n_mels, T, 3) uint8, per-clip max (N,));
        `mel_from_waveform` + `image_encode` in one call, byte for byte, without the (N*C, n_mels, T) tensor in between."""
        wave = self._chk(wave, torch.float32)
        thresholds = self._chk(thresholds, torch.float32)
        C = 2 if stereo else 1
        NC, Lw = wave.shape
        if NC % C:
            raise ValueError("batch must be a multiple of the channel count")
        if Lw <= self.n_fft // 2:
            raise RuntimeError(
                f"Argument #4: Padding size should be less than the corresponding input dimension, "
                f"but got: padding ({self.n_fft // 2}, {self.n_fft // 2}) at dimension 2 of input {list(wave.shape)}"
            )
        N = NC // C
        Tn = self.lib.rfx_stft_frames(self.handle, Lw)
        img = torch.empty((N, self.n_mels, Tn, 3), dtype=torch.uint8, device=wave.device)
        mx = torch.empty((N,), dtype=torch.float32, device=wave.device)
        with self._workspace(self.lib.rfx_image_from_waveform_workspace_bytes(self.handle, N, int(stereo), Lw)) as ws:
            check(self.lib.rfx_image_from_waveform(self.handle, wave.data_ptr(), N, int(stereo), Lw, thresholds.data_ptr(), mx.data_ptr(),
                                                   img.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return img, mx

    def pcm16(self, wave: torch.Tensor, channels: int, normalize: bool = True, out: T.Optional[torch.Tensor] = None):
        """(N*C, L) float32 -> ((N, L, C) int16, per-clip peak (N,)).  `out`: preallocated contiguous (N, L, C) int16
        destination on this device (e.g. the rows of a batch-wide result), otherwise a fresh tensor."""
        wave = self._chk(wave, torch.float32)
        NC, L = wave.shape
        if NC % channels:
            raise ValueError("batch must be a multiple of the channel count")
        N = NC // channels
        if out is not None:
            if out.device != self.device or out.dtype != torch.int16 or tuple(out.shape) != (N, L, channels) or not out.is_contiguous():
                raise ValueError(f"out must be a contiguous int16 tensor of shape {(N, L, channels)} on {self.device}")
            pcm = out
        else:
            pcm = torch.empty((N, L, channels), dtype=torch.int16, device=wave.device)
        peak = torch.zeros((N,), dtype=torch.float32, device=wave.device)
        check(self.lib.rfx_pcm16(wave.data_ptr(), N, channels, L, int(normalize), peak.data_ptr(), pcm.data_ptr(), self._stream()))
        return pcm, peak


# Plans are cached, least recently used first out: a plan pins its tables on the device (the dense filterbank alone is
# 18 MB at the default parameters), and a server that builds its parameters from image EXIF (cli.py:77-87) would otherwise
# grow the cache with every distinct parameter set for ever.  An evicted plan is destroyed when its last user lets go of it.
# The bound is PER DEVICE (round 5): one process driving eight GPUs with two parameter sets holds sixteen plans, and a miss on
# one GPU never evicts another GPU's plan (a rebuild is a filterbank construction, hipMallocs and an 18 MB upload; an eviction's
# hipFree synchronises its device).
PLAN_CACHE_SIZE = max(1, int(os.environ.get("RFX_PLAN_CACHE", "8")))
_plans: "collections.OrderedDict[T.Tuple[T.Any, int, str, str, str, str], Plan]" = collections.OrderedDict()
_plans_lock = threading.Lock()


def _evict_over_bound(plans: "collections.OrderedDict", dev_index: int, bound: int) -> None:
    """Drop the least recently used plans OF ONE DEVICE until at most `bound` of them are cached (key[1] is the device index)."""
    mine = [k for k in plans if k[1] == dev_index]  # OrderedDict iterates least recently used first
    for k in mine[: max(0, len(mine) - bound)]:
        del plans[k]  # dropped from the cache; freed when the last converter holding it goes


def get_plan(params: T.Any, device: T.Union[str, torch.device], gl_form: str = "auto", frame_engine: str = "auto",
             plan_layout: str = "auto", imel_form: str = "auto") -> Plan:
    """Plans are immutable and cached per (frozen params, device, options), at most PLAN_CACHE_SIZE of them per device (least
    recently used evicted): constructing a converter per request, as the reference's server does (server.py:159), costs a
    dictionary lookup.

    `gl_form` picks the Griffin-Lim device form (rfx_plan_options.gl_form): "auto" (per call, from the batch
    shape), "runs" (always the run-based fused kernel) or "frames" (always the per-frame kernel + fold);
    `frame_engine` = "generic" keeps Griffin-Lim of the 40 h / 10 h geometries (48 kHz ...) on the generic FFT engine
    instead of the row-family kernels (rfx_plan_options.frame_engine; cross-checks); `plan_layout` = "generic" builds the
    generic plan also for the default geometry (rfx_plan_options.plan_layout; cross-checks of the specialised engine);
    `imel_form` = "groups" keeps InverseMelScale on the group kernels where "auto" takes the wave kernel
    (rfx_plan_options.imel_form; cross-checks)."""
    dev = resolve_device(device)  # 'cuda' is keyed by the GPU it means now, not bound for good to the first one used
    key = (params, dev.index, gl_form, frame_engine, plan_layout, imel_form)
    with _plans_lock:
        plan = _plans.get(key)
        if plan is None:
            plan = Plan(params, dev, gl_form, frame_engine, plan_layout, imel_form)
            _plans[key] = plan
            _evict_over_bound(_plans, dev.index, PLAN_CACHE_SIZE)
        else:
            _plans.move_to_end(key)
    return plan


def cached_plans() -> int:
    with _plans_lock:
        return len(_plans)
