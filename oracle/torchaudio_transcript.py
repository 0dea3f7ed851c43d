"""
SECOND RESTATEMENT of the inverse half - TEST INFRASTRUCTURE ONLY (same rules as riffusion_oracle.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything here).

`riffusion_oracle.py` restates torchaudio 0.13.0's InverseMelScale with a hand-derived gradient and a
hand-written SGD-with-momentum update, and its Griffin-Lim with an out-of-place momentum term.  This
file instead TRANSCRIBES the two torchaudio 0.13.0 functions statement by statement on the machinery
the reference really executes - `torch.optim.SGD`, autograd (`requires_grad=True`, `.backward()`,
`optim.step()`), `specgram.data.clamp`, `tprev.mul_()` - so that the only thing left "from memory" is
the sequence of statements itself, not the calculus.  tests/test_oracle_transcript.py asserts that
the two restatements agree to fp32 round-off, i.e. the hand-derived oracle (which the GPU parity
tests use because it is 3x faster and accepts fp64) computes what autograd + torch.optim.SGD compute.

What remains from memory after this (torchaudio is absent here and on the GPU box, pinned version
`torchaudio==0.13.0` in /root/reference/cog.yaml:28-29):
  * the statement order and constants of transforms.InverseMelScale.forward  (loss = sum over mel
    then mean; clamp AFTER the step; stop test on the pre-step loss; sgdargs default lr 0.1,
    momentum 0.9; init torch.rand(B, T, F));
  * the statement order and constants of functional.griffinlim  (momentum/(1+momentum); complex
    torch.rand init; `angles - tprev.mul_(momentum)`; `.div(abs + 1e-16)`; final istft);
  * functional.melscale_fbanks, which IS pinned (forward golden PNGs).
The constructor arguments come from the reference itself: spectrogram_converter.py:62-73 and :87-99.
"""
from __future__ import annotations

import typing as T

import torch

import riffusion_oracle as O


class InverseMelScaleTranscript(torch.nn.Module):
    """torchaudio 0.13.0 transforms.InverseMelScale, constructed as spectrogram_converter.py:87-99 does."""

    def __init__(
        self,
        n_stft: int,
        n_mels: int = 128,
        sample_rate: int = 16000,
        f_min: float = 0.0,
        f_max: T.Optional[float] = None,
        max_iter: int = 100000,
        tolerance_loss: float = 1e-5,
        tolerance_change: float = 1e-8,
        sgdargs: T.Optional[dict] = None,
        norm: T.Optional[str] = None,
        mel_scale: str = "htk",
    ) -> None:
        super().__init__()
        self.n_mels = n_mels
        self.sample_rate = sample_rate
        self.f_max = f_max or float(sample_rate // 2)
        self.f_min = f_min
        self.max_iter = max_iter
        self.tolerance_loss = tolerance_loss
        self.tolerance_change = tolerance_change
        self.sgdargs = sgdargs or {"lr": 0.1, "momentum": 0.9}
        if f_min > self.f_max:
            raise ValueError("Require f_min: {} < f_max: {}".format(f_min, self.f_max))
        # functional.melscale_fbanks - the pinned restatement (golden PNGs)
        p = O.OracleParams(
            sample_rate=sample_rate,
            num_frequencies=n_mels,
            min_frequency=f_min,
            max_frequency=self.f_max,
            mel_scale_norm=norm,
            mel_scale_type=mel_scale,
        )
        fb = O.mel_filterbank(p)
        assert fb.shape == (n_stft, n_mels), "transcript supports the reference's n_stft = n_fft//2+1 at its sample rate"
        self.register_buffer("fb", fb)
        self.steps_run = 0

    def forward(self, melspec: torch.Tensor, spec0: T.Optional[torch.Tensor] = None) -> torch.Tensor:
        # pack batch
        shape = melspec.size()
        melspec = melspec.view(-1, shape[-2], shape[-1])

        n_mels, time = shape[-2], shape[-1]
        freq, _ = self.fb.size()  # (freq, n_mels)
        melspec = melspec.transpose(-1, -2)
        if self.n_mels != n_mels:
            raise ValueError("Expected an input with {} mel bins. Found: {}".format(self.n_mels, n_mels))

        if spec0 is None:
            specgram = torch.rand(
                melspec.size()[0], time, freq, requires_grad=True, dtype=melspec.dtype, device=melspec.device
            )
        else:  # test hook: the injected initial guess stands where torch.rand's draw would
            specgram = spec0.detach().clone().to(melspec.dtype).requires_grad_(True)

        optim = torch.optim.SGD([specgram], **self.sgdargs)

        loss = float("inf")
        self.steps_run = 0
        for _ in range(self.max_iter):
            optim.zero_grad()
            diff = melspec - specgram.matmul(self.fb)
            new_loss = diff.pow(2).sum(axis=-1).mean()
            # take sum over mel-frequency then average over other dimensions
            # so that loss threshold is applied par unit timeframe
            new_loss.backward()
            optim.step()
            specgram.data = specgram.data.clamp(min=0)
            self.steps_run += 1

            new_loss = new_loss.item()
            if new_loss < self.tolerance_loss or abs(loss - new_loss) < self.tolerance_change:
                break
            loss = new_loss

        specgram.requires_grad_(False)
        specgram = specgram.clamp(min=0).transpose(-1, -2)

        # unpack batch
        specgram = specgram.view(shape[:-2] + (freq, time))
        return specgram


def inverse_mel_scale(p: O.OracleParams) -> InverseMelScaleTranscript:
    """The module exactly as the reference constructs it, spectrogram_converter.py:87-99."""
    return InverseMelScaleTranscript(
        n_stft=p.n_stft,
        n_mels=p.num_frequencies,
        sample_rate=p.sample_rate,
        f_min=p.min_frequency,
        f_max=p.max_frequency,
        max_iter=p.max_mel_iters,
        tolerance_loss=1e-5,
        tolerance_change=1e-8,
        sgdargs=None,
        norm=p.mel_scale_norm,
        mel_scale=p.mel_scale_type,
    )


def griffinlim_transcript(
    specgram: torch.Tensor,
    window: torch.Tensor,
    n_fft: int,
    hop_length: int,
    win_length: int,
    power: float,
    n_iter: int,
    momentum: float,
    length: T.Optional[int],
    rand_init: bool,
    angles0: T.Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """torchaudio 0.13.0 functional.griffinlim, statement by statement (`angles0`: test hook for the init)."""
    if not 0 <= momentum < 1:
        raise ValueError("momentum must be in range [0, 1). Found: {}".format(momentum))
    momentum = momentum / (1 + momentum)

    # pack batch
    shape = specgram.size()
    specgram = specgram.reshape([-1] + list(shape[-2:]))

    specgram = specgram.pow(1 / power)

    # initialize the phase
    cdtype = torch.complex64 if specgram.dtype == torch.float32 else torch.complex128
    if angles0 is not None:
        angles = angles0.reshape(specgram.size()).to(cdtype).clone()
    elif rand_init:
        angles = torch.rand(specgram.size(), dtype=cdtype, device=specgram.device)
    else:
        angles = torch.full(specgram.size(), 1, dtype=cdtype, device=specgram.device)

    # And initialize the previous iterate to 0
    tprev = torch.tensor(0.0, dtype=specgram.dtype, device=specgram.device)
    for _ in range(n_iter):
        # Invert with our current estimate of the phases
        inverse = torch.istft(
            specgram * angles, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, length=length
        )

        # Rebuild the spectrogram
        rebuilt = torch.stft(
            input=inverse,
            n_fft=n_fft,
            hop_length=hop_length,
            win_length=win_length,
            window=window,
            center=True,
            pad_mode="reflect",
            normalized=False,
            onesided=True,
            return_complex=True,
        )

        # Update our phase estimates
        angles = rebuilt
        if momentum:
            angles = angles - tprev.mul_(momentum)
        angles = angles.div(angles.abs().add(1e-16))

        # Store the previous iterate
        tprev = rebuilt

    # Return the final phase estimates
    waveform = torch.istft(
        specgram * angles, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, length=length
    )

    # unpack batch
    waveform = waveform.reshape(shape[:-2] + waveform.shape[-1:])
    return waveform


def griffinlim(specgram: torch.Tensor, p: O.OracleParams, angles0: T.Optional[torch.Tensor] = None,
               n_iter: T.Optional[int] = None) -> torch.Tensor:
    """transforms.GriffinLim.forward as the reference constructs it, spectrogram_converter.py:62-73."""
    return griffinlim_transcript(
        specgram,
        O.hann_window(p),
        p.n_fft,
        p.hop_length,
        p.win_length,
        1.0,
        p.num_griffin_lim_iters if n_iter is None else n_iter,
        0.99,
        None,
        True,
        angles0=angles0,
    )
