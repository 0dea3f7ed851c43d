"""
Command line front-end for the spectrogram <-> audio codec.

Same sub-commands and keyword flags as the part of the reference CLI that drives this path
(`riffusion/cli.py:23-95` audio-to-image / image-to-audio / print-exif, `:134-204`
audio-to-images-batch), but the batch commands feed whole batches to the GPU
(`SpectrogramImageConverter.spectrogram_images_from_waveforms` /
`audio_from_spectrogram_images`) instead of one clip per thread-pool task, and split the file list
over the ranks when launched one process per GPU (`python -m torch.distributed.run
--nproc-per-node 8 -m riffusion.cli images-to-audio-batch ...`).  argparse replaces argh (not
installed here); without pydub only 16-bit PCM wav files are read and only wav is written.

    python -m riffusion.cli image-to-audio --image tile.png --audio out.wav
    python -m riffusion.cli audio-to-image --audio clip.wav --image tile.png
    python -m riffusion.cli print-exif --image tile.png
    python -m riffusion.cli images-to-audio-batch --image-dir tiles/ --output-dir wavs/
    python -m riffusion.cli audio-to-images-batch --audio-dir wavs/ --output-dir tiles/
"""
import argparse
import glob
import os
import sys
import typing as T

import numpy as np
from PIL import Image

from riffusion.spectrogram_image_converter import SpectrogramImageConverter
from riffusion.spectrogram_params import SpectrogramParams
from riffusion.util import audio_util, image_util


def _load_segment(path: str) -> T.Any:
    pydub = audio_util._pydub()
    if pydub is not None:
        return pydub.AudioSegment.from_file(path)
    return audio_util.PcmSegment.from_wav(path)


def _params_from_image(image: Image.Image) -> SpectrogramParams:
    """EXIF -> params, defaults when the image carries none (reference cli.py:77-87)."""
    try:
        return SpectrogramParams.from_exif(exif=image.getexif())
    except (KeyError, AttributeError):
        print("WARNING: Could not find spectrogram parameters in exif data. Using defaults.")
        return SpectrogramParams()


def audio_to_image(*, audio: str, image: str, step_size_ms: int = 10, num_frequencies: int = 512, min_frequency: int = 0,
                   max_frequency: int = 10000, window_duration_ms: int = 100, padded_duration_ms: int = 400,
                   power_for_image: float = 0.25, stereo: bool = False, device: str = "cuda") -> None:
    segment = _load_segment(audio)
    params = SpectrogramParams(
        sample_rate=segment.frame_rate, stereo=stereo, window_duration_ms=window_duration_ms,
        padded_duration_ms=padded_duration_ms, step_size_ms=step_size_ms, min_frequency=min_frequency,
        max_frequency=max_frequency, num_frequencies=num_frequencies, power_for_image=power_for_image,
    )
    converter = SpectrogramImageConverter(params=params, device=device)
    pil_image = converter.spectrogram_image_from_audio(segment)
    pil_image.save(image, exif=pil_image.getexif(), format="PNG")
    print(f"Wrote {image}")


def image_to_audio(*, image: str, audio: str, device: str = "cuda") -> None:
    pil_image = Image.open(image)
    params = _params_from_image(pil_image)
    converter = SpectrogramImageConverter(params=params, device=device)
    segment = converter.audio_from_spectrogram_image(pil_image, apply_filters=True)
    segment.export(audio, format=os.path.splitext(audio)[1][1:] or "wav")
    print(f"Wrote {audio} ({segment.duration_seconds:.2f} seconds)")


def print_exif(*, image: str) -> None:
    pil_image = Image.open(image)
    exif = image_util.exif_from_image(pil_image)
    for name, value in exif.items():
        print(f"{name:<20} = {value:>15}")


def _rank_slice(items: T.Sequence[T.Any]) -> T.Sequence[T.Any]:
    """Under `torchrun` / `torch.distributed.run` (one process per GPU) every rank converts its contiguous share of
    the files and writes its own outputs: the N-GPU form of the reference's per-file thread pool (cli.py:172-204).
    No process group is needed - files are independent and nothing is gathered."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world <= 1:
        return items
    from riffusion.batch_shard import shard_range

    lo, hi = shard_range(len(items), world, rank)
    return items[lo:hi]


def _rank_device(device: str) -> str:
    """'cuda' -> this rank's GPU when launched one process per GPU."""
    if device == "cuda" and "LOCAL_RANK" in os.environ:
        return f"cuda:{int(os.environ['LOCAL_RANK'])}"
    return device


def images_to_audio_batch(*, image_dir: str, output_dir: str, batch_size: int = 64, no_filters: bool = False,
                          device: str = "cuda") -> None:
    """Decode every *.png of a directory, `batch_size` same-width tiles per GPU call.  Each clip then gets the same
    post-processing as `image-to-audio` (audio_util.apply_filters, reference spectrogram_image_converter.py:65-91)
    unless --no-filters is given."""
    os.makedirs(output_dir, exist_ok=True)
    device = _rank_device(device)
    paths = _rank_slice(sorted(glob.glob(os.path.join(image_dir, "*.png"))))
    groups: T.Dict[T.Tuple[SpectrogramParams, T.Tuple[int, int]], T.List[str]] = {}
    for path in paths:
        with Image.open(path) as im:
            groups.setdefault((_params_from_image(im), im.size), []).append(path)
    for (params, _size), members in groups.items():
        converter = SpectrogramImageConverter(params=params, device=device)
        for i in range(0, len(members), batch_size):
            chunk = members[i : i + batch_size]
            tiles = []
            for p in chunk:
                with Image.open(p) as im:
                    tiles.append(image_util.rgb_array_from_image(im))
            pcm = converter.audio_from_spectrogram_images(np.stack(tiles))
            for path, samples in zip(chunk, pcm):
                segment = audio_util.PcmSegment(samples, params.sample_rate)
                if not no_filters:
                    segment = audio_util.apply_filters(segment, compression=False)
                out = os.path.join(output_dir, os.path.splitext(os.path.basename(path))[0] + ".wav")
                segment.export(out, format="wav")
            print(f"Wrote {len(chunk)} clips to {output_dir}")


def audio_to_images_batch(*, audio_dir: str, output_dir: str, image_extension: str = "jpg", step_size_ms: int = 10,
                          num_frequencies: int = 512, min_frequency: int = 0, max_frequency: int = 10000,
                          power_for_image: float = 0.25, mono: bool = False, sample_rate: int = 44100, device: str = "cuda",
                          num_threads: int = 0, limit: int = -1, batch_size: int = 64) -> None:
    """Process audio clips into spectrogram images in batch (reference cli.py:134-204, same flags and defaults: stereo
    tiles unless --mono, files resampled to --sample-rate, unreadable files skipped, jpg output).  Instead of one clip
    per thread-pool task (`num_threads` is accepted and ignored) same-length clips go to the GPU `batch_size` at a time."""
    import torch

    os.makedirs(output_dir, exist_ok=True)
    device = _rank_device(device)
    image_format = {"jpg": "JPEG", "jpeg": "JPEG", "png": "PNG"}[image_extension]
    paths = sorted(p for p in glob.glob(os.path.join(audio_dir, "*")) if os.path.isfile(p))
    if limit > 0:
        paths = paths[:limit]
    paths = _rank_slice(paths)
    params = SpectrogramParams(step_size_ms=step_size_ms, num_frequencies=num_frequencies, min_frequency=min_frequency,
                               max_frequency=max_frequency, power_for_image=power_for_image, stereo=not mono,
                               sample_rate=sample_rate)
    converter = SpectrogramImageConverter(params=params, device=device)
    channels = 1 if mono else 2
    # Streaming: files are decoded one at a time and grouped by sample count (a GPU call needs equal lengths); a group is
    # converted and released as soon as it holds `batch_size` clips, and when the waveforms held in host memory pass
    # `max_pending_bytes` the LARGEST group is flushed early - a directory of long clips of many different lengths never
    # sits in RAM as a whole (the reference streams one file per thread-pool task, cli.py:172-204).
    max_pending_bytes = 2 << 30
    pending: T.Dict[int, T.List[T.Tuple[str, np.ndarray]]] = {}
    pending_bytes = 0

    def flush(n_samples: int) -> None:
        nonlocal pending_bytes
        chunk = pending.pop(n_samples)
        pending_bytes -= sum(w.nbytes for _, w in chunk)
        images, max_values = converter.spectrogram_images_from_waveforms(torch.from_numpy(np.stack([w for _, w in chunk])))
        for (path, _), image, mx in zip(chunk, images, max_values):
            exif_data = params.to_exif()
            exif_data[SpectrogramParams.ExifTags.MAX_VALUE.value] = float(mx)
            image.getexif().update(exif_data.items())
            out = os.path.join(output_dir, os.path.splitext(os.path.basename(path))[0] + "." + image_extension)
            image.save(out, exif=image.getexif(), format=image_format)
        print(f"Wrote {len(chunk)} images to {output_dir}")

    for path in paths:
        try:
            seg = _load_segment(path)
        except Exception:  # the reference skips files it cannot read (cli.py:176-179)
            continue
        if seg.channels != channels:
            seg = seg.set_channels(channels)
        if seg.frame_rate != params.sample_rate:
            seg = seg.set_frame_rate(params.sample_rate)
        wave = np.array([c.get_array_of_samples() for c in seg.split_to_mono()]).astype(np.float32)
        del seg
        group = pending.setdefault(wave.shape[1], [])
        group.append((path, wave))
        pending_bytes += wave.nbytes
        if len(group) >= batch_size:
            flush(wave.shape[1])
        while pending_bytes > max_pending_bytes and pending:
            flush(max(pending, key=lambda n: sum(w.nbytes for _, w in pending[n])))
    for n_samples in sorted(pending):
        flush(n_samples)


_COMMANDS: T.Dict[str, T.Callable[..., None]] = {
    "audio-to-image": audio_to_image,
    "image-to-audio": image_to_audio,
    "print-exif": print_exif,
    "images-to-audio-batch": images_to_audio_batch,
    "audio-to-images-batch": audio_to_images_batch,
}


def build_parser() -> argparse.ArgumentParser:
    import inspect

    parser = argparse.ArgumentParser(prog="riffusion.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = parser.add_subparsers(dest="command", required=True)
    for name, fn in _COMMANDS.items():
        sp = sub.add_parser(name, help=(fn.__doc__ or "").strip().split("\n")[0])
        for arg, spec in inspect.signature(fn).parameters.items():
            flag = "--" + arg.replace("_", "-")
            if spec.default is inspect.Parameter.empty:
                sp.add_argument(flag, required=True)
            elif isinstance(spec.default, bool):
                sp.add_argument(flag, action="store_true", default=spec.default)
            else:
                sp.add_argument(flag, type=type(spec.default), default=spec.default)
    return parser


def main(argv: T.Optional[T.Sequence[str]] = None) -> None:
    args = vars(build_parser().parse_args(argv))
    command = args.pop("command")
    _COMMANDS[command](**args)


if __name__ == "__main__":
    main(sys.argv[1:])
