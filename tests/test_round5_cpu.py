"""
Round 5, CPU side (no GPU needed).

* the line the InverseMelScale wave kernel uses instead of a weight table (rfx_api.hip, wave-kernel admission;
  rfx_imel.hip::imel_wave_kernel) stays within one ulp of a group's LARGEST weight on the reference's banks - the plan's gate is
  relative to that maximum (4e-7), not an absolute 1e-6 that an area-normalised bank (weights ~1e-2) would pass at 1e-4 relative;
* host logic added in round 5: per-device plan cache bound, the one-time warning for `group` without `gather`, ChunkSource.prefetch;
* the run partition of the Griffin-Lim launches (rfx_kernels.h::gl_run_start through the C ABI): a partition of the frames for
  every batch shape, equal runs by default, and with a dispatch-order skew (ablation builds) never a run under 11 frames.
"""
import os
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worst_line_deviation(fb: np.ndarray):
    """(worst |line - weight|, worst of the same over the group's largest weight) for the two weight lines of every mel group,
    with the grouping and the double-precision least-squares fit of rfx_api.hip."""
    F, M = fb.shape
    nz = fb != 0
    act = nz.any(1)
    rows = np.where(act)[0]
    f_lo, f_hi = rows[0], rows[-1] + 1
    m0 = np.where(act, nz.argmax(1), -1)
    w0 = np.where(act, fb[np.arange(F), np.maximum(m0, 0)], 0).astype(np.float32)
    w1 = np.where(act & (m0 + 1 < M), fb[np.arange(F), np.minimum(m0 + 1, M - 1)], 0).astype(np.float32)
    worst_abs = worst_rel = 0.0
    for g in range(M):
        idx = np.where(m0[f_lo:f_hi] == g)[0] + f_lo
        n = len(idx)
        if n == 0:
            continue
        for w in (w0, w1):
            y, x = w[idx].astype(np.float64), np.arange(n, dtype=np.float64)
            slope = (n * (x * y).sum() - x.sum() * y.sum()) / (n * (x * x).sum() - x.sum() ** 2) if n > 1 else 0.0
            icpt = (y.sum() - slope * x.sum()) / n
            dev = np.abs(float(np.float32(icpt)) + float(np.float32(slope)) * x - y).max()
            worst_abs, worst_rel = max(worst_abs, dev), max(worst_rel, dev / max(np.abs(y).max(), 1e-30))
    return worst_abs, worst_rel


@pytest.mark.parametrize("norm", [None, "slaney"])
def test_weight_lines_of_the_wave_kernel_are_within_an_ulp_of_the_groups_largest_weight(norm):
    import riffusion_oracle as O

    fb = O.mel_filterbank(O.OracleParams(mel_scale_norm=norm)).numpy()
    a, r = worst_line_deviation(fb)
    print(f"norm={norm}: largest weight {fb.max():.4f}; worst |line - weight| {a:.3e} absolute, {r:.3e} of the group's largest weight")
    assert r <= 2e-7  # the plan's gate is 4e-7 (rfx_api.hip); measured 0.72e-7 (no norm) / 1.16e-7 (slaney)
    src = open(os.path.join(ROOT, "riffusion-hobby_amd", "csrc", "rfx_api.hip")).read()
    assert "> 4e-7 * wmax" in src and "> 1e-6) wave_ok" not in src  # the gate this test's margin refers to


def test_plan_cache_bound_is_per_device(monkeypatch):
    """one process driving several GPUs must not thrash a global eight-entry cache - the bound counts the plans
    of ONE device, and a miss on a device evicts that device's least recently used plan only."""
    import collections
    import gc

    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    made, closed = [], []

    class FakePlan:
        def __init__(self, params, dev, *opts):
            self.key = (params.max_mel_iters, dev.index)
            made.append(self.key)

        def __del__(self):
            closed.append(self.key)

    monkeypatch.setattr(_hip, "Plan", FakePlan)
    monkeypatch.setattr(_hip, "resolve_device", lambda d: torch.device(d))
    monkeypatch.setattr(_hip, "_plans", collections.OrderedDict())
    monkeypatch.setattr(_hip, "PLAN_CACHE_SIZE", 2)
    ps = [SpectrogramParams(max_mel_iters=100 + i) for i in range(3)]
    for dev in range(4):
        for p in ps[:2]:
            _hip.get_plan(p, f"cuda:{dev}")
    assert _hip.cached_plans() == 8 and not closed  # 4 devices x 2 plans: nothing evicted (a global bound of 2 would keep 2)
    _hip.get_plan(ps[2], "cuda:1")  # a third parameter set on device 1: evicts device 1's least recently used plan only
    gc.collect()
    assert closed == [(100, 1)] and _hip.cached_plans() == 8
    n = len(made)
    for dev in (0, 2, 3):
        for p in ps[:2]:
            _hip.get_plan(p, f"cuda:{dev}")
    assert len(made) == n  # the other devices' plans are all still hits


def test_chunk_source_prefetch_is_optional_and_idempotent():
    """`prefetch` exists so that the caller can queue a chunk's kernels BEFORE the host stages the next chunk;
    without a GPU (and for device input) it is a no-op and `get` keeps returning plain slices, in any order."""
    from riffusion import batch_shard

    items = torch.arange(12 * 2, dtype=torch.uint8).reshape(12, 2)
    bounds = [(0, 5), (5, 10), (10, 12)]
    src = batch_shard.ChunkSource(items, bounds, torch.device("cpu"))
    for i in (-1, 0, 1, 1, 2, 3, 99):
        src.prefetch(i)
    assert not src.ready
    assert [src.get(i).tolist() for i in (2, 0, 1)] == [items[a:b].tolist() for a, b in (bounds[2], bounds[0], bounds[1])]
    # the batch entry point calls it after a chunk's kernels and hands the plan down instead of fetching it per stage
    import inspect

    from riffusion.spectrogram_converter import SpectrogramConverter
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter

    body = inspect.getsource(SpectrogramImageConverter.audio_from_spectrogram_images)
    assert body.index("conv._waveform_from_mel(plan") < body.index("source.prefetch(i + 1)") < body.index("sink.put(")
    assert "self._plan()" not in inspect.getsource(SpectrogramConverter._waveform_from_mel)


def test_default_gather_without_a_process_group_is_silent():
    from riffusion import batch_shard

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert batch_shard.default_gather(None) == "none"
        assert batch_shard.default_gather(True) == "none"  # no process group initialised: nothing to warn about


def _lib():
    import ctypes

    lib = ctypes.CDLL(os.path.join(ROOT, "riffusion-hobby_amd", "librfx.so"))
    lib.rfx_debug_run_start.restype = ctypes.c_int64
    lib.rfx_debug_run_start.argtypes = [ctypes.c_int64] * 6
    return lib


@pytest.mark.parametrize("B,T", [(64, 512), (65, 512), (100, 512), (48, 512), (128, 512), (11, 512), (512, 40), (10, 1024), (3, 2000)])
@pytest.mark.parametrize("skew", [0, 60, 100, 250])
def test_run_partition_by_dispatch_order_is_a_partition(B, T, skew):
    """rfx_kernels.h::gl_run_start (through the C ABI, no GPU): the runs of a launch tile the frames exactly, in order, the first
    h of them longer by 2 * skew per mille of the mean than the others, none shorter than 11 frames when the host's admission rule
    (rfx_api.hip::gl_partition) lets the skew through - so a hop block (10 frames) is shared by at most two runs."""
    lib = _lib()
    runs, h, N = 512, 256, B * T
    admitted = N // 10 >= runs and (N * (1000 - skew)) // (1000 * runs) >= 11
    w1, w2 = (1000 + skew, 1000 - skew) if admitted else (1000, 1000)
    if N // 10 < runs:
        runs = max(1, N // 10)
    st = [lib.rfx_debug_run_start(b, runs, N, h, w1, w2) for b in range(runs + 1)]
    assert st[0] == 0 and st[-1] == N and all(a < b for a, b in zip(st, st[1:]))
    lens = np.diff(st)
    assert lens.min() >= 10
    if w1 == w2:
        assert lens.max() - lens.min() <= 1 and st == [N * b // runs for b in range(runs + 1)]  # the equal partition, exactly
    else:
        first, second = lens[:h], lens[h:]
        assert first.max() - first.min() <= 1 and second.max() - second.min() <= 1 and lens.min() >= 11
        mean = N / runs
        assert abs(first.mean() / mean - w1 / 1000) < 2e-3 and abs(second.mean() / mean - w2 / 1000) < 2e-3
