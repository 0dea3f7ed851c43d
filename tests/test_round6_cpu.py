"""
Round 6, the parts that need no GPU: the workspace arena's bookkeeping, the per-call options struct, the Griffin-Lim partition in
whole groups, the exponents of the numeric-range contract, the comment-insensitive source fingerprint of bench.py, and a numpy
model of the canonical overlap-add that shows why a clip's bits cannot depend on the run partition.
"""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from riffusion import _hip

    return _hip.load_library()


# ---- workspace arena -----------------------------------------------------------------------------------------------------------

def _arena(max_idle=4):
    from riffusion import _hip

    return _hip.WorkspaceArena(torch.device("cpu"), max_idle=max_idle)


def test_arena_reuses_a_buffer_on_the_same_stream_and_grows_only():
    a = _arena()
    b1 = a.take(100, stream=7)
    assert b1.numel() == a.GRANULE and b1.dtype == torch.uint8 and a.allocations == 1
    a.give(b1, 7)
    b2 = a.take(a.GRANULE - 1, stream=7)          # fits: the same buffer comes back
    assert b2.data_ptr() == b1.data_ptr() and a.allocations == 1 and a.idle_bytes() == 0
    a.give(b2, 7)
    b3 = a.take(a.GRANULE + 1, stream=7)          # too small: replaced by a bigger one, not kept next to it
    assert b3.numel() == 2 * a.GRANULE and a.allocations == 2 and a.idle_bytes() == 0
    a.give(b3, 7)
    assert a.idle_bytes() == 2 * a.GRANULE
    a.clear()
    assert a.idle_bytes() == 0


def test_arena_never_hands_one_buffer_to_two_calls_and_keeps_streams_apart():
    a = _arena()
    x, y = a.take(10, 1), a.take(10, 1)           # two calls in flight on one stream (two threads of a pool): two buffers
    assert x.data_ptr() != y.data_ptr() and a.allocations == 2
    a.give(x, 1)
    z = a.take(10, 2)                             # another stream never gets a buffer whose kernels may still run on stream 1
    assert z.data_ptr() != x.data_ptr() and a.allocations == 3
    a.give(y, 1)
    a.give(z, 2)
    got = {a.take(10, 1).data_ptr(), a.take(10, 1).data_ptr()}
    assert got == {x.data_ptr(), y.data_ptr()} and a.allocations == 3


def test_arena_bounds_its_idle_buffers_least_recently_used_first():
    a = _arena(max_idle=2)
    bufs = [a.take(10, s) for s in (1, 2, 3)]
    for s, b in zip((1, 2, 3), bufs):
        a.give(b, s)                              # the third give evicts stream 1's buffer
    assert a.idle_bytes() == 2 * a.GRANULE
    assert a.take(10, 3).data_ptr() == bufs[2].data_ptr() and a.take(10, 2).data_ptr() == bufs[1].data_ptr()
    before = a.allocations
    a.take(10, 1)
    assert a.allocations == before + 1            # stream 1's was dropped


def test_arena_under_threads():
    a = _arena(max_idle=8)
    seen, lock, errors = set(), threading.Lock(), []

    def worker():
        for _ in range(200):
            b = a.take(1000, 5)
            with lock:
                if b.data_ptr() in seen:
                    errors.append("one buffer checked out twice")
                seen.add(b.data_ptr())
            b[:8] = 1
            with lock:
                seen.discard(b.data_ptr())
            a.give(b, 5)

    threads = [threading.Thread(target=worker) for _ in range(6)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors and a.allocations <= 6


def test_borrowed_workspace_returns_on_exceptions_too():
    from riffusion import _hip

    a = _arena()
    with pytest.raises(RuntimeError):
        with _hip._Borrowed(a, 10, 3) as ws:
            assert ws.numel() >= 10 and a.idle_bytes() == 0
            raise RuntimeError("the C call failed")
    assert a.idle_bytes() == a.GRANULE


# ---- rfx_call_options ----------------------------------------------------------------------------------------------------------

def test_call_options_struct_matches_the_header(repo_root):
    from riffusion import _hip

    o = _hip.call_options(row_base=5, magnitude_hint=30e6)
    assert ctypes.sizeof(_hip.RfxCallOptions) == 24 and o.struct_size == 24 and o.flags == 0 and o.row_base == 5
    assert _hip.RfxCallOptions.row_base.offset == 8 and _hip.RfxCallOptions.magnitude_hint.offset == 16
    with pytest.raises(ValueError):
        _hip.call_options(row_base=-1)
    header = open(os.path.join(repo_root, "include", "rfx.h")).read()
    body = header[header.index("typedef struct {\n  uint32_t struct_size;\n  uint32_t flags;"):header.index("} rfx_call_options;")]
    assert [ln.split()[1].rstrip(";") for ln in body.splitlines()[1:] if ln.strip()] == ["struct_size", "flags", "row_base", "magnitude_hint", "reserved"]
    for name in ("rfx_inverse_mel_ex", "rfx_griffinlim_ex", "rfx_waveform_from_mel_ex", "rfx_audio_from_image_u8_ex"):
        assert name in header and name in _hip.SIGNATURES


def test_options_are_validated_before_any_device_work(lib):
    from riffusion import _hip

    bad = _hip.RfxCallOptions(0, 0, 0, 0.0, 0.0)  # struct_size not set
    rc = lib.rfx_inverse_mel_ex(None, None, 1, 1, 1, None, 0, None, None, 0, None, ctypes.byref(bad))
    assert rc == -1 and b"struct_size" in lib.rfx_last_error()
    flagged = _hip.RfxCallOptions(24, 1, 0, 0.0, 0.0)
    assert lib.rfx_griffinlim_ex(None, None, None, 0, 1, 2, 0, 0.5, None, None, 0, None, ctypes.byref(flagged), None) == -1
    assert b"flags" in lib.rfx_last_error()
    for hint in (-1.0, float("nan"), float("inf")):
        o = _hip.RfxCallOptions(24, 0, 0, hint, 0.0)
        assert lib.rfx_waveform_from_mel_ex(None, None, 1, 1, 1, 0, 1, 0.5, None, None, 0, None, ctypes.byref(o)) == -1
        assert b"magnitude_hint" in lib.rfx_last_error()


# ---- Griffin-Lim partition: whole groups, the extra group first ---------------------------------------------------------------------

@pytest.mark.parametrize("B,T", [(64, 512), (65, 512), (100, 512), (1, 512), (7, 101), (523, 22), (9, 57), (3, 16), (1, 23)])
def test_partition_is_made_of_whole_groups_and_the_first_runs_are_the_long_ones(lib, B, T):
    slots, cap = 512, 1024
    st = (ctypes.c_int64 * cap)()
    runs = lib.rfx_debug_gl_partition(slots, B, T, ctypes.cast(st, ctypes.c_void_p), cap)
    s = list(st[: runs + 1])
    groups_per_row = -(-T // 16)
    assert runs == min(slots, B * groups_per_row)
    assert s[0] == 0 and s[-1] == B * T and all(a < b for a, b in zip(s, s[1:]))
    assert all((x % T) % 16 == 0 for x in s[:-1])          # every run starts at a group boundary of its row
    # in units of groups: the first r runs hold q + 1, the others q
    def group_index(frame):
        row, t = divmod(frame, T)
        return row * groups_per_row + t // 16 if frame < B * T else B * groups_per_row
    sizes = [group_index(b) - group_index(a) for a, b in zip(s, s[1:])]
    q, r = divmod(B * groups_per_row, runs)
    assert sizes == [q + 1] * r + [q] * (runs - r)
    if (B, T) == (64, 512):
        assert set(b - a for a, b in zip(s, s[1:])) == {64}


# ---- numeric range: the two exponents -----------------------------------------------------------------------------------------------

def test_range_exponents(lib):
    def ex(mx, mel):
        e, j = ctypes.c_int(), ctypes.c_int()
        assert lib.rfx_debug_range_exponents(mx, mel, ctypes.byref(e), ctypes.byref(j)) == 0
        return e.value, j.value

    assert ex(30e6, 1) == (60, 0)          # the reference's default max_value: 2^-60 as in rounds 2-5, Griffin-Lim unscaled
    assert ex(30e6 * 2.0 ** 13, 1) == (73, 13) and ex(30e6 * 2.0 ** -13, 1) == (47, -13)   # a power of two moves both by itself
    assert ex(1e20, 1) == (102, 42)
    assert ex(1e-6, 1) == (30, -25)        # tiny mel amplitudes: the U[0,1) start of the untouched bins sets both scales
    assert ex(0.0, 1) == (35, -25) and ex(float("nan"), 1) == (35, -25)
    assert ex(float("inf"), 1) == (126, 100) and ex(3e38, 0)[1] == 100
    assert ex(1000.0, 0) == (45, -16)      # standalone Griffin-Lim: the row's own magnitudes decide
    assert ex(1e-30, 0)[1] == -100         # (clamped: eps^2 must stay a normal float)


# ---- bench.py: the fingerprint follows the code, not the comments -----------------------------------------------------------------

def test_source_fingerprint_ignores_comments_and_space(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    assert bench.strip_comments_and_space('int a = 1; // c\n/* x\n y */ char* s = "a // b";\n') == 'inta=1;char*s="a // b";'
    d = tmp_path / "riffusion-hobby_amd" / "csrc"
    d.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (d / "k.hip").write_text("__global__ void k(float* x) { x[0] = 1.f; }  // one\n")
    f1 = bench.kernel_source_fingerprint(("k.hip",))
    (d / "k.hip").write_text("// a new header comment\n__global__ void k(float* x) {\n    x[0] = 1.f;  /* moved */\n}\n")
    assert bench.kernel_source_fingerprint(("k.hip",)) == f1
    (d / "k.hip").write_text("__global__ void k(float* x) { x[0] = 2.f; }\n")
    assert bench.kernel_source_fingerprint(("k.hip",)) != f1


# ---- why the bits cannot depend on the partition: a numpy model of the two device forms ----------------------------------------------

def _ola_model(y, w, scale, T, runs):
    """gl_iter_kernel's overlap-add for one sample position n' of every hop block: y[t][j] = frame t's synthesis sample for block
    t - 5 + j, w[j] the window, `runs` = [(t0, t1)] whole groups.  Returns the block values a0 + a1 the next launch reads."""
    f32 = np.float32
    nblk = T - 1
    buf = np.zeros((2, nblk), f32)
    written = np.zeros((2, nblk), bool)

    def emit(par, tg0, tg1, blk, val):
        if blk < 0 or blk >= nblk:
            return
        buf[par, blk] = f32(val) * scale[blk]
        written[par, blk] = True
        if max(blk - 4, 0) >= tg0 and min(blk + 5, T - 1) <= tg1:
            buf[par ^ 1, blk] = 0
            written[par ^ 1, blk] = True

    for t0, t1 in runs:
        acc = np.zeros(10, f32)
        tg0, tg1, par = t0, min(T - 1, t0 + 15), (t0 // 16) & 1
        for fr in range(t0, t1 + 1):
            if fr % 16 == 0 and fr != t0:
                for j in range(9):
                    emit(par, tg0, tg1, fr - 5 + j, acc[j])
                    acc[j] = 0
                tg0, tg1, par = fr, min(T - 1, fr + 15), par ^ 1
            for j in range(10):
                acc[j] = f32(np.float64(y[fr, j]) * np.float64(w[j]) + np.float64(acc[j]))  # fma: one rounding
            emit(par, tg0, tg1, fr - 5, acc[0])
            acc[:9] = acc[1:]
            acc[9] = 0
        for j in range(9):
            emit(par, tg0, tg1, t1 - 4 + j, acc[j])
    assert written.all()
    return buf[0] + buf[1]


def _fold_model(y, w, scale, T):
    """gl_fold_kernel: two fma chains split at the group boundary inside the block's frames, each scaled, then added."""
    f32 = np.float32
    out = np.zeros(T - 1, f32)
    for blk in range(T - 1):
        tlo, thi = max(blk - 4, 0), min(blk + 5, T - 1)
        cut = thi & ~15
        lo = hi = f32(0)
        for t in range(tlo, thi + 1):
            j = blk - t + 5
            v = np.float64(y[t, j]) * np.float64(w[j])
            if t < cut:
                lo = f32(v + np.float64(lo))
            else:
                hi = f32(v + np.float64(hi))
        out[blk] = f32(lo * scale[blk]) + f32(hi * scale[blk])
    return out


@pytest.mark.parametrize("T", [48, 57, 112])
def test_canonical_groups_make_the_overlap_add_independent_of_the_partition(T):
    rng = np.random.default_rng(T)
    y = (rng.standard_normal((T, 10)) * 1000).astype(np.float32)
    w = rng.random(10).astype(np.float32)
    scale = (rng.random(T - 1) + 0.5).astype(np.float32)
    bounds = list(range(0, T, 16)) + [T]
    one_run = _ola_model(y, w, scale, T, [(0, T - 1)])
    every_group = _ola_model(y, w, scale, T, [(a, b - 1) for a, b in zip(bounds, bounds[1:])])
    two_runs = _ola_model(y, w, scale, T, [(0, 31), (32, T - 1)])
    fold = _fold_model(y, w, scale, T)
    for other in (every_group, two_runs, fold):
        assert np.array_equal(one_run.view(np.uint32), other.view(np.uint32))
    # the round-5 arithmetic (one chain per block inside a run) is NOT partition-free: that is what round 6 removed
    plain = np.zeros(T - 1, np.float32)
    for blk in range(T - 1):
        acc = np.float32(0)
        for t in range(max(blk - 4, 0), min(blk + 5, T - 1) + 1):
            acc = np.float32(np.float64(y[t, blk - t + 5]) * np.float64(w[blk - t + 5]) + np.float64(acc))
        plain[blk] = acc * scale[blk]
    assert not np.array_equal(plain.view(np.uint32), one_run.view(np.uint32))
