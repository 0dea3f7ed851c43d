"""
Round 3, the scalable batch entry point: which clips a rank returns (`gather=`), results that never
leave the GPU (`return_device=`), pinned / overlapped device-to-host copies, argument validation.
Everything goes through the C ABI (librfx.so).
"""
import socket

import numpy as np
import pytest
import torch

from helpers import synthetic_tiles_u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def conv():
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    return SpectrogramImageConverter(SpectrogramParams(stereo=True, num_griffin_lim_iters=8), device="cuda")


def test_device_result_equals_host_result_and_chunks_overlap_safely(conv):
    """return_device=True hands back the tensor the PCM kernel wrote (no D2H); the host route streams chunk k to pinned
    memory on a side stream while chunk k+1 computes - same bits, for every chunk size incl. a ragged last chunk."""
    tiles = synthetic_tiles_u8(7, 512, 40, seed=3)
    dev = conv.audio_from_spectrogram_images(tiles, seed=5, tiles_per_call=3, return_device=True)
    assert isinstance(dev, torch.Tensor) and dev.is_cuda and dev.dtype == torch.int16 and dev.shape == (7, 441 * 39, 2)
    for tpc in (1, 3, 7, 64):
        host = conv.audio_from_spectrogram_images(tiles, seed=5, tiles_per_call=tpc)
        assert isinstance(host, np.ndarray) and host.dtype == np.int16
        # a chunk's seed is base_seed + 2*first_clip: equal chunking -> equal bits
        if tpc == 3:
            assert np.array_equal(host, dev.cpu().numpy())
        assert host.shape == (7, 441 * 39, 2) and np.abs(host).max() == 32767
    # the float waveforms take the same two routes
    wd = conv.audio_from_spectrogram_images(tiles[:2], seed=5, return_waveform=True, return_device=True)
    wh = conv.audio_from_spectrogram_images(tiles[:2], seed=5, return_waveform=True)
    assert wd.shape == (2, 2, 441 * 39) and np.array_equal(wd.cpu().numpy(), wh)
    # results of consecutive calls do not alias (each owns its pinned block)
    a = conv.audio_from_spectrogram_images(tiles[:2], seed=1)
    a_copy = a.copy()
    b = conv.audio_from_spectrogram_images(tiles[:2], seed=2)
    assert np.array_equal(a, a_copy) and not np.array_equal(a, b)


def test_gather_modes_over_rccl_world_size_1(conv):
    """One RCCL rank: "all", "rank0" and "none" return the same clips as the ungrouped call, on the host and on the device."""
    import torch.distributed as dist

    tiles = synthetic_tiles_u8(5, 512, 40, seed=8)
    plain = conv.audio_from_spectrogram_images(tiles, seed=77, tiles_per_call=2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        for gather in ("all", "rank0", "none"):
            got = conv.audio_from_spectrogram_images(tiles, seed=77, group=dist.group.WORLD, tiles_per_call=2, gather=gather)
            assert np.array_equal(plain, got), gather
            got_dev = conv.audio_from_spectrogram_images(tiles, seed=77, group=True, tiles_per_call=2, gather=gather, return_device=True)
            assert got_dev.is_cuda and np.array_equal(plain, got_dev.cpu().numpy()), gather
        # the collectives themselves on device tensors (int16 carried as bytes), one preallocated destination
        from riffusion.batch_shard import gather_clips

        x = torch.arange(24, dtype=torch.int16, device="cuda").reshape(4, 3, 2)
        assert torch.equal(gather_clips(x, 4, dist.group.WORLD), x) and torch.equal(gather_clips(x, 4, dist.group.WORLD, dst=0), x)
    finally:
        dist.destroy_process_group()


def test_argument_validation(conv):
    tiles = synthetic_tiles_u8(1, 512, 40, seed=1)
    with pytest.raises(ValueError, match="tiles_per_call"):
        conv.audio_from_spectrogram_images(tiles, tiles_per_call=0)
    with pytest.raises(ValueError, match="tiles_per_call"):
        conv.audio_from_spectrogram_images(tiles, tiles_per_call=-4)
    with pytest.raises(ValueError, match="gather"):
        conv.audio_from_spectrogram_images(tiles, gather="some")
    with pytest.raises(ValueError, match=r"\[0, 1\]"):
        conv.audio_from_spectrogram_images(tiles.astype(np.float32))  # 0..255 pixel values as float: refused, not quantised to garbage
    ok = conv.audio_from_spectrogram_images(tiles.astype(np.float32) / 255.0, seed=3)  # the pipeline's [0, 1] range
    assert ok.shape == (1, 441 * 39, 2)
