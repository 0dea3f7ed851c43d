"""
Spectrogram images <-> audio segments.

Drop-in for the reference's `riffusion/spectrogram_image_converter.py:10-91` (same constructor,
attributes and the two per-clip methods) plus batch entry points that keep many tiles in flight on
the GPU: `audio_from_spectrogram_images` takes uint8 tiles in and hands int16 PCM out with one H2D
and one D2H copy per batch, takes the diffusion pipeline's output tensors directly
(`riffusion_pipeline.py:427-434`), and shards a batch of clips over the ranks of a
`torch.distributed` process group (one process per GPU; clips are independent, so the only
collective is the final all_gather of the int16 PCM).
"""
import typing as T

import numpy as np
import torch
from PIL import Image

from riffusion.spectrogram_converter import SpectrogramConverter
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import audio_util, image_util


class SpectrogramImageConverter:
    def __init__(self, params: SpectrogramParams, device: str = "cuda"):
        self.p = params
        self.device = device
        self.converter = SpectrogramConverter(params=params, device=device)

    # ---- reference API: one clip per call -------------------------------------------------------------
    def spectrogram_image_from_audio(self, segment: T.Any) -> Image.Image:
        """Audio segment -> spectrogram image carrying the params (and MAX_VALUE) as EXIF."""
        assert int(segment.frame_rate) == self.p.sample_rate, "Sample rate mismatch"

        if self.p.stereo:
            if segment.channels == 1:
                print("WARNING: Mono audio but stereo=True, cloning channel")
                segment = segment.set_channels(2)
            elif segment.channels > 2:
                print("WARNING: Multi channel audio, reducing to stereo")
                segment = segment.set_channels(2)
        else:
            if segment.channels > 1:
                print("WARNING: Stereo audio but stereo=False, setting to mono")
                segment = segment.set_channels(1)

        waveform = np.array([c.get_array_of_samples() for c in segment.split_to_mono()]).astype(np.float32)
        images, max_values = self.spectrogram_images_from_waveforms(torch.from_numpy(waveform)[None])
        image = images[0]
        exif_data = self.p.to_exif()
        exif_data[SpectrogramParams.ExifTags.MAX_VALUE.value] = float(max_values[0])
        image.getexif().update(exif_data.items())
        return image

    def audio_from_spectrogram_image(
        self,
        image: Image.Image,
        apply_filters: bool = True,
        max_value: float = 30e6,
    ) -> T.Any:
        """Spectrogram image -> audio segment (the EXIF MAX_VALUE is not read back, like the reference)."""
        pcm = self.audio_from_spectrogram_images(
            np.asarray(image_util.rgb_array_from_image(image))[None], max_value=max_value
        )
        segment = audio_util.segment_from_pcm16(pcm[0], self.p.sample_rate)
        if apply_filters:
            segment = audio_util.apply_filters(segment, compression=False)
        return segment

    # ---- batch entry points ------------------------------------------------------------------------------
    def spectrogram_images_from_waveforms(self, waveforms: torch.Tensor) -> T.Tuple[T.List[Image.Image], np.ndarray]:
        """(N, C, samples) float waveforms at int16 scale -> N RGB images and their float32 MAX_VALUEs."""
        conv = self.converter
        plan = conv._plan()
        N, C, L = waveforms.shape
        if C != (2 if self.p.stereo else 1):
            raise ValueError(f"expected {2 if self.p.stereo else 1} channel(s), got {C}")
        power = float(self.p.power_for_image)
        thr = plan.device_constant(("encode_thresholds", power), lambda: image_util.encode_thresholds(power))
        # one call (rfx_image_from_waveform): the mel amplitudes go from the forward kernel to the encoder without the (N*C, M, T) tensor
        img, mx = plan.image_from_waveform(waveforms.reshape(N * C, L).to(conv.device, torch.float32), self.p.stereo, thr)
        img_np, mx_np = img.cpu().numpy(), mx.cpu().numpy()
        return [Image.fromarray(a, mode="RGB") for a in img_np], mx_np

    @staticmethod
    def quantize_pipeline_images(images: torch.Tensor) -> torch.Tensor:
        """
        The diffusion pipeline's hand-off (`riffusion_pipeline.py:427-434`): the VAE output is moved to
        [0, 1] and NHWC float32 (`(image / 2 + 0.5).clamp(0, 1)`, `.permute(0, 2, 3, 1)`), then
        `numpy_to_pil` quantises it with `(images * 255).round().astype("uint8")`.  This does that
        quantisation on whatever device the tensor lives on (round-half-to-even like numpy, float32
        like the reference), so the decoder can take the tensor without the `.cpu()` -> PIL -> numpy
        -> `.to(device)` round trip.  (N, H, W, 3) float in [0, 1] -> (N, H, W, 3) uint8.
        """
        if images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError("expected (N, H, W, 3) images, channels last, as riffusion_pipeline.py:431 produces them")
        return (images.to(torch.float32) * 255).round().to(torch.uint8)

    def audio_from_spectrogram_images(
        self,
        images_u8: T.Union[np.ndarray, torch.Tensor],
        max_value: float = 30e6,
        seed: T.Optional[int] = None,
        return_waveform: bool = False,
        group: T.Any = None,
        tiles_per_call: int = 64,
        gather: T.Optional[str] = None,
        return_device: bool = False,
        validate: T.Optional[bool] = None,
        return_range_flag: bool = False,
    ) -> T.Union[np.ndarray, torch.Tensor, T.Tuple[torch.Tensor, torch.Tensor]]:
        """
        (N, H, W, 3) RGB tiles -> (n, samples, C) int16 PCM (or, with `return_waveform`, the (n, C, samples)
        float waveforms); a numpy array on the host, or with `return_device=True` a tensor that never left the GPU.

        `images_u8` is a uint8 array / tensor on any device, or the diffusion pipeline's float [0, 1]
        NHWC tensor (quantised here like `numpy_to_pil`, see `quantize_pipeline_images`; a float input
        with values outside [0, 1] - or NaN - is refused: pass pixel values as uint8).  `validate` says when
        that range check runs: None (default) at once for a host tensor, and for a device tensor as a flag
        computed on the device and read where the call synchronises anyway (the copy of the result to the
        host) - no host sync on the one-tile-per-request path; True: at once, with a host sync; False: never
        (`return_device=True` with a device input never synchronises and so cannot raise: there None turns a failed check
        into an all-zero result - silence, not garbage audio - and attaches the device flag as `result.range_ok`; with
        `return_range_flag=True` the call returns `(result, range_ok)` instead, the flag as a value of its own: an attribute
        does not survive slicing, `.to()` or a gather).
        Host tiles are uploaded chunk by chunk through pinned memory on a side stream (`batch_shard.ChunkSource`).

        `group`: a `torch.distributed` process group (or True for the default group).  Every rank
        passes the SAME full batch; rank r converts clips `shard_range(N, world, r)` on its own GPU
        (a clip's channels never leave their rank: they share the SGD loss mean and the peak
        normalisation), `tiles_per_call` clips at a time.  `gather` says which clips a rank RETURNS
        (`batch_shard.result_rows`):
            "none"  (default, also `None`) the own shard only, no collective at all: each rank writes / serves its own
                    clips, as the reference's consumers do (server.py:159-183, cli.py:172-204) - the mode that scales;
            "rank0" the whole batch on the group's rank 0 (one RCCL gather), the own shard elsewhere;
            "all"   the whole batch on every rank - one RCCL all_gather_into_tensor of the int16 PCM (round 2-3 default:
                    every rank receives and copies all N clips, so it does not scale; ask for it explicitly).
        Host results are staged through pinned memory; without a collective each chunk's device-to-host
        copy runs on a side stream while the next chunk computes (`batch_shard.ChunkSink`).
        With a `seed`, a clip's audio is a function of (the clip, the seed, the clip's index in `images_u8`) alone - the same
        bytes whatever `tiles_per_call` is and however many ranks share the batch (round 6: the random starts are keyed by
        the clip's global row, and nothing in the kernels' arithmetic depends on the batch a clip travels in).  Without one the
        seed is drawn from torch's global generator, like the reference's random starts (in a group: pass a seed, or seed
        torch identically on every rank).
        """
        from riffusion import batch_shard

        if tiles_per_call < 1:
            raise ValueError(f"tiles_per_call must be >= 1, got {tiles_per_call}")
        if return_range_flag and not return_device:
            raise ValueError("return_range_flag goes with return_device=True (a host result raises on a failed range check instead)")
        if gather is None:
            gather = batch_shard.default_gather(group)
        if gather not in batch_shard.GATHER_MODES:
            raise ValueError(f"gather must be one of {batch_shard.GATHER_MODES}, got {gather!r}")
        conv = self.converter
        plan = conv._plan()
        imgs = torch.as_tensor(np.ascontiguousarray(images_u8) if isinstance(images_u8, np.ndarray) else images_u8)
        range_msg = ("float images must be the pipeline's [0, 1] output (riffusion_pipeline.py:427-431); "
                     "pass 0..255 pixel values as uint8")
        range_ok = None  # device flag of a deferred range check
        if imgs.is_floating_point():
            if imgs.numel() and validate is not False:
                ok = ((imgs >= 0) & (imgs <= 1.0 + 1e-6)).all()  # NaN compares false
                if validate or not imgs.is_cuda:
                    if not bool(ok):
                        raise ValueError(range_msg)
                else:
                    range_ok = ok
            imgs = self.quantize_pipeline_images(imgs)
        n_total = imgs.shape[0]
        C = 2 if self.p.stereo else 1
        L = plan.lib.rfx_griffinlim_output_samples(plan.handle, int(imgs.shape[2]))
        base_seed = conv._seed(seed)
        power, max_value = float(self.p.power_for_image), float(max_value)
        if not (max_value > 0.0 and max_value < float("inf")):
            raise ValueError(f"max_value must be a positive finite number, got {max_value}")
        lut = plan.device_constant(("decode_lut", power, max_value), lambda: image_util.decode_lut(power, max_value))
        row_shape, dtype = ((C, L), torch.float32) if return_waveform else ((L, C), torch.int16)

        pg = None if group is None else batch_shard._resolve_group(group)
        world = 1 if group is None else torch.distributed.get_world_size(pg)
        # a shard that takes part in a collective stays on the device until the collective has run
        collective = world > 1 and gather != "none"

        def convert(lo: int, hi: int) -> torch.Tensor:
            sink = batch_shard.ChunkSink(hi - lo, row_shape, dtype, plan.device, to_host=not (collective or return_device))
            bounds = [(a, min(hi, a + tiles_per_call)) for a in range(lo, hi, tiles_per_call)]  # bounded working set: |S| alone is 19 MB per tile-channel
            source = batch_shard.ChunkSource(imgs, bounds, plan.device)
            # (the scratch space - 1.7 GB for 64 mono tiles - comes from the plan's arena: the same buffer chunk after chunk and call after call)
            for i, (a, b) in enumerate(bounds):
                if return_waveform:
                    mel = plan.image_decode(source.get(i), self.p.stereo, lut)
                    wave = conv._waveform_from_mel(plan, mel, seed=base_seed, channels_per_clip=C, row_base=a * C, magnitude_hint=max_value)
                    out = wave.reshape(b - a, C, -1)
                else:  # uint8 tiles -> int16 PCM in one call (rfx_audio_from_image_u8_ex), same bytes as the three calls above + pcm16
                    dst = sink.rows(a - lo, b - lo)  # device sink: the PCM kernel writes the batch rows in place
                    out = plan.audio_from_image(source.get(i), self.p.stereo, lut, self.p.num_griffin_lim_iters, 0.99, seed=base_seed,
                                                normalize=True, out=dst, clip_base=a, magnitude_hint=max_value)[0]
                # this chunk's kernels are queued: the host stages and uploads the next chunk underneath them
                source.prefetch(i + 1)
                sink.put(a - lo, b - lo, out)
            return sink.finish()  # a rank with an empty shard still joins the collective with 0 rows

        result = batch_shard.sharded_map(convert, n_total, group, gather)
        if return_device:
            if range_ok is not None:
                # nothing on this path ever synchronises, so the deferred check cannot raise here: out-of-range (or NaN) input
                # yields SILENCE instead of garbage audio (one in-place multiply on the device, no host sync, no copy of the batch)
                result.mul_(range_ok.to(result.dtype))
            # the flag travels with the result for a caller that wants to look: `result.range_ok`, a 0-dim bool tensor on the device
            # (True when nothing was checked: uint8 input, validate=False, or a check that already ran on the host).  It is a plain
            # attribute: slicing / .to() / a gather make a new tensor without it - read it from the tensor this call returned.
            flag = range_ok if range_ok is not None else torch.ones((), dtype=torch.bool, device=result.device)
            result.range_ok = flag
            return (result, flag) if return_range_flag else result
        if result.is_cuda:
            host = torch.empty(result.shape, dtype=result.dtype, pin_memory=True)  # gathered batch: one pinned copy
            host.copy_(result, non_blocking=True)
            torch.cuda.current_stream(plan.device).synchronize()
            result = host
        if range_ok is not None and not bool(range_ok):  # the stream has been synchronised above / by the sink: no extra wait
            raise ValueError(range_msg)
        return result.numpy()
