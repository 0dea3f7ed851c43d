"""
Round 6: a clip's audio is a function of the clip, the seed and the clip's index - not of the batch it is converted in.

Two things made the result depend on chunking and sharding until round 5: the random starts were keyed by the row index INSIDE
the call (rfx_call_options.row_base now carries the global one), and Griffin-Lim's overlap-add summed a hop block as two partial
chains wherever a run boundary of the launch happened to fall (csrc/rfx_kernels.h: canonical groups of kGlGroup frames now split
every chain at the same places, in both device forms).  Also here: the plan's workspace arena (no allocator traffic in steady
state, two threads never share a buffer) and the 2-rank RCCL tests that run by themselves on a box with two GPUs.
Everything goes through the C ABI.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import synthetic_tiles_u8

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _conv(stereo=False, iters=32, **kw):
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    return SpectrogramImageConverter(SpectrogramParams(stereo=stereo, num_griffin_lim_iters=iters, **kw), device="cuda")


@pytest.mark.parametrize("stereo,width", [(False, 512), (True, 512), (False, 101)])
def test_pcm_is_independent_of_chunking_and_sharding(stereo, width):
    """Same tiles, same seed: one call, chunks of 7, one tile per call (the per-frame Griffin-Lim form) and two manual shards
    give byte-identical PCM.  width 101: rows that are not whole groups of 16 frames."""
    conv = _conv(stereo, iters=8 if width != 512 else 32)
    n = 20
    tiles = synthetic_tiles_u8(n, 512, width, seed=66)
    whole = conv.audio_from_spectrogram_images(tiles, seed=4242, tiles_per_call=64)
    assert whole.dtype == np.int16 and whole.shape[0] == n and np.abs(whole.astype(np.int32)).max() > 30000
    for per_call in (7, 1):
        got = conv.audio_from_spectrogram_images(tiles, seed=4242, tiles_per_call=per_call)
        assert np.array_equal(got, whole), f"tiles_per_call={per_call}: {int((got != whole).sum())} samples differ"
    # the same clips as two separately converted shards: what two ranks compute (batch_shard.shard_range), here on one GPU
    from riffusion import _hip

    plan = conv.converter._plan()
    C = 2 if stereo else 1
    from riffusion.util import image_util

    lut = plan.device_constant(("decode_lut", 0.25, 30e6), lambda: image_util.decode_lut(0.25, 30e6))
    dev_tiles = torch.from_numpy(tiles).cuda()
    for lo, hi in ((0, 3), (3, 5), (17, 20)):
        pcm, _ = plan.audio_from_image(dev_tiles[lo:hi], stereo, lut, conv.p.num_griffin_lim_iters, 0.99, seed=4242, clip_base=lo, magnitude_hint=30e6)
        assert np.array_equal(pcm.cpu().numpy(), whole[lo:hi]), f"shard [{lo}:{hi}) differs"
    # ... and a different row_base IS a different draw (the key really is the global row)
    other, _ = plan.audio_from_image(dev_tiles[3:5], stereo, lut, conv.p.num_griffin_lim_iters, 0.99, seed=4242, clip_base=0, magnitude_hint=30e6)
    assert not np.array_equal(other.cpu().numpy(), whole[3:5])
    assert _hip.call_options(3 * C).row_base == 3 * C


@pytest.mark.parametrize("rate", [48000, 22050, 11025])
def test_other_engines_are_independent_of_chunking_too(rate):
    """The row-family (48 / 22.05 kHz) and generic (11.025 kHz) engines: per-frame kernels + a fold in fixed order were always
    partition-free; with the global row in the key of the random starts their PCM is chunking-free as well."""
    conv = _conv(False, iters=4, sample_rate=rate)
    tiles = synthetic_tiles_u8(5, 512, 64, seed=rate)
    whole = conv.audio_from_spectrogram_images(tiles, seed=11, tiles_per_call=5)
    assert np.abs(whole.astype(np.int32)).max() > 30000
    for per_call in (2, 1):
        assert np.array_equal(conv.audio_from_spectrogram_images(tiles, seed=11, tiles_per_call=per_call), whole), (rate, per_call)
    assert not np.array_equal(conv.audio_from_spectrogram_images(tiles, seed=12, tiles_per_call=5), whole)


@pytest.mark.parametrize("B", [64, 32, 16])
def test_forward_run_skew_changes_no_bit(B):
    """The forward kernel's runs by dispatch order (StftMelArgs::run_skew: the first half of the grid walks longer runs) apply where
    the launch is exactly two workgroups per CU - B x ceil(T / fpb) = 512: B = 64, 32, 16 at T = 512.  Frames are independent, so
    the mel amplitudes, the image and its maximum must be what the same clips give in a batch one clip short (another launch shape:
    equal runs), bit for bit."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    plan = _hip.get_plan(SpectrogramParams(), "cuda")
    wave = torch.from_numpy((np.random.default_rng(B).standard_normal((B, 441 * 511)) * 8000).astype(np.float32)).cuda()
    thr = torch.from_numpy(image_util.encode_thresholds(0.25)).cuda()
    mel = plan.mel_from_waveform(wave)
    ref = torch.cat([plan.mel_from_waveform(wave[: B - 1]), plan.mel_from_waveform(wave[B - 1:])])
    assert torch.equal(mel.view(torch.int32), ref.view(torch.int32))
    img, mx = plan.image_from_waveform(wave, False, thr)
    img_a, mx_a = plan.image_from_waveform(wave[: B - 1], False, thr)
    img_b, mx_b = plan.image_from_waveform(wave[B - 1:], False, thr)
    assert torch.equal(img, torch.cat([img_a, img_b])) and torch.equal(mx, torch.cat([mx_a, mx_b]))
    assert torch.equal(mx, mel.amax(dim=(1, 2)))  # the per-workgroup keys add up to the image's maximum
    # stereo images: an image's maximum spans the keys of its two rows
    if B == 64:
        from riffusion.spectrogram_params import SpectrogramParams as P

        plan2 = _hip.get_plan(P(stereo=True), "cuda")
        img2, mx2 = plan2.image_from_waveform(wave[:32].contiguous(), True, thr)
        assert torch.equal(mx2, mel[:32].reshape(16, 2, 512, 512).amax(dim=(1, 2, 3))) and img2.shape == (16, 512, 512, 3)


def test_a_step_count_that_outgrows_64_kb_of_lds_falls_back_to_the_general_kernel():
    """the line-form / group / wave SGD kernels take their loss history from dynamic LDS without the opt-in attribute;
    a full-band bank with max_mel_iters = 2000 needs 72 KB there.  One function (imel_kernel_choice) now decides for the launcher,
    rfx_plan_imel_kernel and the fused path alike: such a plan reports the general kernel (0) and runs on it; the same bank at
    200 steps keeps its line-form kernel (5)."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    small = _hip.get_plan(SpectrogramParams(min_frequency=20, max_frequency=20000), "cuda")
    big = _hip.get_plan(SpectrogramParams(min_frequency=20, max_frequency=20000, max_mel_iters=2000), "cuda")
    assert small.lib.rfx_plan_imel_kernel(small.handle) == 5 and big.lib.rfx_plan_imel_kernel(big.handle) == 0
    mel = torch.rand(1, 512, 6, generator=torch.Generator().manual_seed(1)).cuda() * 1e6
    out = big.unpack_magnitudes(big.inverse_mel(mel, 1, seed=3), 1, 6)
    assert bool(torch.isfinite(out).all()) and float(out.max()) > 0
    wave = big.waveform_from_mel(torch.rand(1, 512, 30, generator=torch.Generator().manual_seed(2)).cuda() * 1e6, 1, 2, 0.99, seed=4)
    assert wave.shape == (1, 441 * 29) and bool(torch.isfinite(wave).all())


def test_one_tile_decode_can_be_captured_in_a_hip_graph():
    """No entry point allocates, synchronises or touches host state once the plan exists, so the whole one-tile decode (75 launches)
    records into a HIP graph and replays to the same bytes.  (It buys nothing - 1.054 ms replayed against 1.045 ms eager,
    tools/probe_graph.py: the path is bound by its kernels, not by their launches - but a caller that captures its own pipeline
    can include this call.)"""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    plan = _hip.get_plan(SpectrogramParams(num_griffin_lim_iters=8), "cuda")
    tile = torch.from_numpy(synthetic_tiles_u8(1, 512, 128, seed=8)).cuda()
    lut = plan.device_constant(("decode_lut", 0.25, 30e6), lambda: image_util.decode_lut(0.25, 30e6))
    out = torch.empty((1, 441 * 127, 1), dtype=torch.int16, device="cuda")
    ws = plan.audio_from_image_workspace(1, False, 128)

    def call():
        plan.audio_from_image(tile, False, lut, 8, 0.99, seed=7, out=out, workspace=ws, magnitude_hint=30e6)

    call()
    torch.cuda.synchronize()
    ref = out.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_float_waveforms_are_independent_of_chunking():
    """return_waveform=True (decode, rfx_waveform_from_mel_ex per chunk): bit-identical float waveforms for 9 tiles in chunks of 9 / 4 / 1."""
    conv = _conv(False, iters=6)
    tiles = synthetic_tiles_u8(9, 512, 200, seed=5)
    ref = conv.audio_from_spectrogram_images(tiles, seed=77, return_waveform=True, tiles_per_call=9)
    for per_call in (4, 1):
        got = conv.audio_from_spectrogram_images(tiles, seed=77, return_waveform=True, tiles_per_call=per_call)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"tiles_per_call={per_call}"


@pytest.mark.parametrize("B,Tn", [(3, 512), (5, 100), (70, 48)])
def test_run_form_and_frame_form_give_the_same_bits(B, Tn):
    """The run-based kernel (one launch per iteration, register sliding window) and the per-frame kernel + fold give every sample
    the same bits: the fold reproduces the run kernel's fma chains and their split at the group boundaries."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams()
    runs, frames = _hip.get_plan(p, "cuda", gl_form="runs"), _hip.get_plan(p, "cuda", gl_form="frames")
    g = torch.Generator(device="cuda").manual_seed(9 * B + Tn)
    mag = torch.rand(B, runs.n_stft, Tn, device="cuda", generator=g) * 1000.0
    S = runs.pack_magnitudes(mag)
    for n_iter in (0, 1, 5):
        a = runs.griffinlim(S, B, Tn, n_iter, 0.99, seed=31, row_base=11)
        b = frames.griffinlim(S, B, Tn, n_iter, 0.99, seed=31, row_base=11)
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"n_iter={n_iter}: {int((a != b).sum())} of {a.numel()} samples differ"


def test_a_clip_inside_any_batch_equals_the_clip_alone():
    """Griffin-Lim of one row, alone and as row 37 of a batch of 65 (run boundaries elsewhere, another partition): same bits when
    the call says which global row it is."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    plan = _hip.get_plan(SpectrogramParams(), "cuda")
    B, Tn = 65, 512
    g = torch.Generator(device="cuda").manual_seed(1)
    S = torch.rand(B * Tn, plan.frame_stride, device="cuda", generator=g) * 1000.0
    S = plan.pack_magnitudes(plan.unpack_magnitudes(S, B, Tn))
    whole = plan.griffinlim(S, B, Tn, 6, 0.99, seed=5)
    for row in (0, 37, 64):
        alone = plan.griffinlim(S[row * Tn:(row + 1) * Tn].contiguous(), 1, Tn, 6, 0.99, seed=5, row_base=row)
        assert torch.equal(alone[0].view(torch.int32), whole[row].view(torch.int32)), f"row {row}"
    part = plan.griffinlim(S[30 * Tn:].contiguous(), 35, Tn, 6, 0.99, seed=5, row_base=30)
    assert torch.equal(part.view(torch.int32), whole[30:].view(torch.int32))


def test_run_partition_is_made_of_whole_groups():
    """rfx_griffinlim_runs: every run starts at a multiple of 16 frames of its row; at the headline shape every one of the 512
    resident workgroups gets 64 frames."""
    import ctypes

    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    plan = _hip.get_plan(SpectrogramParams(), "cuda", gl_form="runs")
    for B, Tn in ((64, 512), (65, 512), (100, 60), (9, 57), (1, 512)):
        cap = 4096
        starts = (ctypes.c_int64 * cap)()
        runs = plan.lib.rfx_griffinlim_runs(plan.handle, B, Tn, 1, ctypes.cast(starts, ctypes.c_void_p), cap)
        st = list(starts[: runs + 1])
        assert st[0] == 0 and st[-1] == B * Tn and all(a < b for a, b in zip(st, st[1:]))
        assert all((s % Tn) % 16 == 0 for s in st[:-1])
        if (B, Tn) == (64, 512):
            assert runs == 512 and all(b - a == 64 for a, b in zip(st, st[1:]))


def test_workspace_arena_no_allocator_traffic_in_steady_state():
    """thirty consecutive product calls at the headline shape ask the device allocator for nothing."""
    conv = _conv(False, iters=2)
    tiles = torch.from_numpy(synthetic_tiles_u8(64, 512, 512, seed=3)).cuda()
    plan = conv.converter._plan()
    for k in range(3):
        conv.audio_from_spectrogram_images(tiles, seed=k, return_device=True)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["num_device_alloc"]
    made = plan.arena.allocations
    for k in range(30):
        out = conv.audio_from_spectrogram_images(tiles, seed=10 + k, return_device=True)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["num_device_alloc"] == before, "the steady state reached the device allocator"
    assert plan.arena.allocations == made and plan.arena.idle_bytes() > 0
    assert out.shape == (64, 441 * 511, 1) and bool(out.range_ok)
    # a smaller call reuses the buffer, a bigger one replaces it (grow-only, one idle buffer per stream)
    conv.audio_from_spectrogram_images(tiles[:5], seed=1, return_device=True)
    assert plan.arena.allocations == made
    plan.release_workspaces()
    assert plan.arena.idle_bytes() == 0
    conv.audio_from_spectrogram_images(tiles[:5], seed=1, return_device=True)
    assert plan.arena.allocations == made + 1


def test_threads_sharing_a_converter_get_their_own_workspaces():
    """cli.py:172-204: one converter, a thread pool.  Same stream, concurrent host calls: the arena hands every call in flight
    its own buffer, and the results equal the serial ones."""
    from multiprocessing.pool import ThreadPool

    conv = _conv(False, iters=3)
    tiles = [synthetic_tiles_u8(2, 512, 64, seed=s) for s in range(8)]
    serial = [conv.audio_from_spectrogram_images(t, seed=9) for t in tiles]
    with ThreadPool(4) as pool:
        threaded = pool.map(lambda t: conv.audio_from_spectrogram_images(t, seed=9), tiles * 3)
    for i, got in enumerate(threaded):
        assert np.array_equal(got, serial[i % 8])


# ---- 2-rank RCCL: light up by themselves on a box with two GPUs -------------------------------------------------------------

_WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "riffusion-hobby_amd")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from helpers import synthetic_tiles_u8
from riffusion.spectrogram_image_converter import SpectrogramImageConverter
from riffusion.spectrogram_params import SpectrogramParams
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"])); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
conv = SpectrogramImageConverter(SpectrogramParams(stereo=True, num_griffin_lim_iters=8), device=str(dev))
tiles = synthetic_tiles_u8(7, 512, 128, seed=12)
want = np.load(sys.argv[2])
from riffusion.batch_shard import result_rows
for gather in ("none", "rank0", "all"):
    got = conv.audio_from_spectrogram_images(tiles, seed=99, group=dist.group.WORLD, gather=gather, tiles_per_call=3)
    lo, hi = result_rows(7, dist.group.WORLD, gather)
    assert got.shape[0] == hi - lo and np.array_equal(got, want[lo:hi]), (rank, gather)
dist.barrier(); dist.destroy_process_group()
print(json.dumps({"rank": rank, "ok": True}))
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_the_rank_worker_itself_with_one_rank(tmp_path):
    """The worker script of the 2-rank test below under `torch.distributed.run --nproc-per-node=1` (RCCL communicator with one
    rank): runs on every box, so the script cannot rot while it waits for a second GPU."""
    _run_ranks(tmp_path, 1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL, one process per GPU)")
def test_two_ranks_equal_one_rank_byte_for_byte(tmp_path):
    """clips sharded over two ranks, gather none / rank0 / all, equal the single-process result byte for byte."""
    _run_ranks(tmp_path, 2)


def _run_ranks(tmp_path, nproc):
    conv = _conv(True, iters=8)
    want = conv.audio_from_spectrogram_images(synthetic_tiles_u8(7, 512, 128, seed=12), seed=99)
    ref, script = tmp_path / "want.npy", tmp_path / "worker.py"
    np.save(ref, want)
    script.write_text(_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), ROOT, str(ref)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count('"ok": true') == nproc


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL, one process per GPU)")
def test_bench_two_gpus_prints_one_line_with_n_gpus_2():
    import json

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-forward", "--no-other-rates", "--no-other-configs"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and "per_rank_ms" in out
