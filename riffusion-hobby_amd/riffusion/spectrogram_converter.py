"""
Audio <-> mel-amplitude spectrograms on the MI355X.

Drop-in for the reference's `riffusion/spectrogram_converter.py:12-204`: same constructor, same
attributes (`p`, `device`, `spectrogram_func`, `inverse_spectrogram_func`, `mel_scaler`,
`inverse_mel_scaler`), same methods and tensor layouts.  The four torchaudio modules the reference
builds (:47-99) are replaced by callables that run hand-written HIP kernels through librfx.so:

    spectrogram_func          torchaudio.transforms.Spectrogram(power=None)  -> framed 17640-pt transform
    mel_scaler                torchaudio.transforms.MelScale                 -> fp32-MFMA GEMM
    inverse_mel_scaler        torchaudio.transforms.InverseMelScale (SGD)    -> banded on-chip SGD
    inverse_spectrogram_func  torchaudio.transforms.GriffinLim               -> fused per-frame iteration

The two torch-level methods fuse these pairs so the (B, n_stft, T) intermediates of the reference are
never materialised; the standalone callables exist for code that pokes at the members directly.
Every call is re-entrant: the converter holds immutable constants only (plans are cached per
(params, device)), workspaces are checked out of the plan's arena per call (`_hip.WorkspaceArena`: two
threads never share one) - the reference shares one converter across a thread pool (cli.py:172-204).  There is no CPU implementation: on a machine without a GPU the constructor still
mirrors the reference's fallback warning, and any compute call raises.
"""
import typing as T
import warnings

import numpy as np
import torch

from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import audio_util, torch_util


class _HipOp:
    """A callable member standing where the reference keeps a torchaudio nn.Module."""

    def __init__(self, owner: "SpectrogramConverter", fn: T.Callable[..., torch.Tensor], name: str):
        self._owner, self._fn, self._name = owner, fn, name

    def __call__(self, *args: T.Any, **kwargs: T.Any) -> torch.Tensor:
        return self._fn(*args, **kwargs)

    def to(self, device: T.Any) -> "_HipOp":  # nn.Module-style chaining used by the reference ctor
        return self

    def __repr__(self) -> str:
        return f"<rfx HIP op {self._name} on {self._owner.device}>"


class SpectrogramConverter:
    def __init__(self, params: SpectrogramParams, device: str = "cuda"):
        self.p = params
        self.device = torch_util.check_device(device)
        if device.lower().startswith("mps"):
            warnings.warn(
                "WARNING: MPS does not support audio operations, falling back to CPU for them",
                stacklevel=2,
            )
            self.device = "cpu"

        self.spectrogram_func = _HipOp(self, self._spectrogram, "Spectrogram(power=None)")
        self.inverse_spectrogram_func = _HipOp(self, self._griffinlim, "GriffinLim")
        self.mel_scaler = _HipOp(self, self._mel_scale, "MelScale")
        self.inverse_mel_scaler = _HipOp(self, self._inverse_mel_scale, "InverseMelScale")

    # ---- plan -------------------------------------------------------------------------------
    def _plan(self):
        from riffusion import _hip  # deferred: importing the package must work without the .so

        if not str(self.device).startswith("cuda"):
            raise RuntimeError(
                f"SpectrogramConverter(device={self.device!r}): this build runs on the MI355X only "
                "(HIP kernels through librfx.so); there is no CPU implementation"
            )
        return _hip.get_plan(self.p, self.device)

    @property
    def _channels_per_clip(self) -> int:
        return 2 if self.p.stereo else 1

    # ---- the four members, with the reference's tensor layouts ----------------------------------
    def _spectrogram(self, waveform: torch.Tensor) -> torch.Tensor:
        """(B, samples) -> (B, n_stft, T) complex64."""
        plan = self._plan()
        lead = waveform.shape[:-1]
        w = waveform.reshape(-1, waveform.shape[-1]).to(self.device)
        _, spec, Tn = plan.stft(w, want_mag=False, want_spec=True)
        return plan.unpack_complex(spec, w.shape[0], Tn).reshape(*lead, plan.n_stft, Tn)

    def _mel_scale(self, amplitudes: torch.Tensor) -> torch.Tensor:
        """(B, n_stft, T) -> (B, n_mels, T); standalone use only (the fused path never builds the input)."""
        plan = self._plan()
        lead = amplitudes.shape[:-2]
        x = amplitudes.reshape(-1, amplitudes.shape[-2], amplitudes.shape[-1]).to(self.device)
        return plan.mel_scale(x).reshape(*lead, plan.n_mels, x.shape[-1])

    def _inverse_mel_scale(
        self, melspec: torch.Tensor, *, spec0: T.Optional[torch.Tensor] = None, seed: T.Optional[int] = None
    ) -> torch.Tensor:
        """(B, n_mels, T) -> (B, n_stft, T)."""
        plan = self._plan()
        B, _, Tn = melspec.shape
        spec0 = spec0.to(self.device) if spec0 is not None else None
        slots = plan.inverse_mel(melspec.to(self.device), B, spec0=spec0, seed=self._seed(seed))
        return plan.unpack_magnitudes(slots, B, Tn)

    def _griffinlim(
        self, specgram: torch.Tensor, *, angles0: T.Optional[torch.Tensor] = None, seed: T.Optional[int] = None
    ) -> torch.Tensor:
        """(B, n_stft, T) magnitudes -> (B, hop*(T-1))."""
        plan = self._plan()
        B, _, Tn = specgram.shape
        slots = plan.pack_magnitudes(specgram.to(self.device))
        a0 = plan.pack_complex(angles0.to(self.device)) if angles0 is not None else None
        return plan.griffinlim(slots, B, Tn, self.p.num_griffin_lim_iters, 0.99, angles0_slots=a0, seed=self._seed(seed))

    @staticmethod
    def _seed(seed: T.Optional[int]) -> int:
        # the reference draws from torch's global generator (no seed parameter exists in its API):
        # derive ours from the same generator so torch.manual_seed() controls reproducibility
        if seed is not None:
            return int(seed)
        return int(torch.randint(0, 2**62, (1,)).item())

    # ---- numpy / pydub level (reference :101-163) -------------------------------------------------
    def spectrogram_from_audio(self, audio: T.Any) -> np.ndarray:
        """Audio segment -> (channels, n_mels, T) float32 mel amplitudes."""
        assert int(audio.frame_rate) == self.p.sample_rate, "Audio sample rate must match params"
        waveform = np.array([c.get_array_of_samples() for c in audio.split_to_mono()])
        if waveform.dtype != np.float32:
            waveform = waveform.astype(np.float32)
        waveform_tensor = torch.from_numpy(waveform).to(self.device)
        amplitudes_mel = self.mel_amplitudes_from_waveform(waveform_tensor)
        return amplitudes_mel.cpu().numpy()

    def audio_from_spectrogram(self, spectrogram: np.ndarray, apply_filters: bool = True) -> T.Any:
        """(channels, n_mels, T) mel amplitudes -> audio segment with that many channels."""
        amplitudes_mel = torch.from_numpy(np.ascontiguousarray(spectrogram)).to(self.device)
        plan = self._plan()
        waveform = self._waveform_from_mel(plan, amplitudes_mel)
        # peak-normalise + int16 truncation on the device (audio_util.py:22-28), one D2H of int16
        pcm, _ = plan.pcm16(waveform, channels=waveform.shape[0], normalize=True)
        segment = audio_util.segment_from_pcm16(pcm[0].cpu().numpy(), self.p.sample_rate)
        if apply_filters:
            segment = audio_util.apply_filters(segment, compression=False)
        return segment

    # ---- torch level seam (reference :165-204) ----------------------------------------------------
    def mel_amplitudes_from_waveform(self, waveform: torch.Tensor) -> torch.Tensor:
        """(B, samples) -> (B, n_mels, T): framed transform, magnitude and mel GEMM without leaving the GPU."""
        return self._plan().mel_from_waveform(waveform.to(self.device))

    def waveform_from_mel_amplitudes(
        self,
        amplitudes_mel: torch.Tensor,
        *,
        spec0: T.Optional[torch.Tensor] = None,
        angles0: T.Optional[torch.Tensor] = None,
        seed: T.Optional[int] = None,
        channels_per_clip: T.Optional[int] = None,
    ) -> torch.Tensor:
        """
        (B, n_mels, T) -> (B, hop*(T-1)).  The reference treats the whole batch as ONE clip (the SGD
        loss mean couples its rows); `channels_per_clip` lets batched callers say how many consecutive
        rows form a clip (default: all of them, like the reference).  `spec0` (B, T, n_stft) and
        `angles0` (B, n_stft, T) inject the two random initialisations (tests); otherwise they are drawn
        on the device from `seed` / torch's global generator.
        """
        return self._waveform_from_mel(self._plan(), amplitudes_mel, spec0=spec0, angles0=angles0, seed=seed,
                                       channels_per_clip=channels_per_clip)

    def _waveform_from_mel(self, plan: T.Any, amplitudes_mel: torch.Tensor, *, spec0: T.Optional[torch.Tensor] = None,
                           angles0: T.Optional[torch.Tensor] = None, seed: T.Optional[int] = None,
                           channels_per_clip: T.Optional[int] = None, row_base: int = 0, magnitude_hint: float = 0.0) -> torch.Tensor:
        """`waveform_from_mel_amplitudes` on a plan the caller already holds (the batch entry points fetch it once per call,
        not once per chunk and stage: a fetch is a lock and a dictionary lookup, and after an eviction a rebuild)."""
        mel = amplitudes_mel.to(self.device)
        B, _, Tn = mel.shape
        cpc = B if channels_per_clip is None else channels_per_clip
        s = self._seed(seed)
        if spec0 is None and angles0 is None:  # the production path: one call (rfx_waveform_from_mel), same bits as the two below
            return plan.waveform_from_mel(mel, cpc, self.p.num_griffin_lim_iters, 0.99, seed=s, row_base=row_base, magnitude_hint=magnitude_hint)
        spec0 = spec0.to(self.device) if spec0 is not None else None
        lin_slots = plan.inverse_mel(mel, cpc, spec0=spec0, seed=s, row_base=row_base, magnitude_hint=magnitude_hint)
        a0 = plan.pack_complex(angles0.to(self.device)) if angles0 is not None else None
        return plan.griffinlim(lin_slots, B, Tn, self.p.num_griffin_lim_iters, 0.99, angles0_slots=a0, seed=s + 1, row_base=row_base,
                               magnitude_hint=magnitude_hint)
