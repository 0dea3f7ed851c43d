"""
CPU check of the formulation the InverseMelScale WAVE kernel uses (csrc/rfx_imel.hip::imel_wave_kernel, round 4): one wave per
frame, the 512 groups dealt to 64 lanes in eight chunks (even chunks in lane order, odd chunks reversed), neighbours through
wave shifts whose end lane takes the adjacent chunk's value of the lane itself, weights as a LINE per group
(w0 = a0 + s0 i, w1 = a1 + s1 i) so that A = a0 S + s0 Q with S = sum x, Q = sum i x; the momentum buffer of a group's bins is
then a line in i as well (two scalars per group instead of a value per bin), and the step of a padding slot is multiplied by
a zero mask.  A float32 numpy restatement of exactly that data flow - lanes, chunks, shifts, pairs -
runs next to the oracle's SGD (torchaudio 0.13 InverseMelScale as restated in oracle/riffusion_oracle.py).  The kernel itself
is pinned on the GPU (tests/test_gpu_round4.py, tests/test_gpu_full_parity.py).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import riffusion_oracle as O  # noqa: E402

f32 = np.float32
PAIRS = (2, 2, 3, 3, 5, 6, 8, 12)  # rfx_kernels.h::kImelWavePairs
FULL = (0, 1, 1, 2, 3, 4, 5, 8)    # rfx_kernels.h::kImelWaveFullPairs: leading pairs that every lane of the chunk fills


def _bank(p):
    fb = O.mel_filterbank(p).numpy()
    nz = fb != 0
    act = np.where(nz.any(1))[0]
    first = nz.argmax(1)
    w0 = np.zeros(fb.shape[0], f32)
    w1 = np.zeros(fb.shape[0], f32)
    for f in act:
        w0[f] = fb[f, first[f]]
        if nz[f].sum() == 2:
            w1[f] = fb[f, first[f] + 1]
    M = fb.shape[1]
    start = np.zeros(M + 1, int)
    cnt = np.bincount(first[act], minlength=M)
    start[0] = act[0]
    start[1:] = act[0] + np.cumsum(cnt)
    return fb, act, w0, w1, start


def _line(w):
    """least-squares line in double, rounded to float32: what rfx_plan_create fits and checks"""
    n = len(w)
    i = np.arange(n, dtype=np.float64)
    if n == 1:
        return f32(w[0]), f32(0)
    sx, sy, sxx, sxy = i.sum(), w.astype(np.float64).sum(), (i * i).sum(), (i * w.astype(np.float64)).sum()
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    return f32((sy - slope * sx) / n), f32(slope)


def group_of(c, lane):
    return 64 * c + 63 - lane if c & 1 else 64 * c + lane


def test_constants_match_the_header_and_the_dealing_is_a_bijection():
    import re

    hdr = open(os.path.join(ROOT, "riffusion-hobby_amd", "csrc", "rfx_kernels.h")).read()
    pairs = tuple(int(x) for x in re.search(r"kImelWavePairs\[kImelWaveChunks\] = \{([^}]*)\}", hdr).group(1).split(","))
    full = tuple(int(x) for x in re.search(r"kImelWaveFullPairs\[kImelWaveChunks\] = \{([^}]*)\}", hdr).group(1).split(","))
    assert pairs == PAIRS and full == FULL and all(f < p for f, p in zip(full, pairs))
    assert "return (chunk & 1) ? 64 * chunk + 63 - lane : 64 * chunk + lane;" in hdr  # group_of() below
    seen = sorted(group_of(c, lane) for c in range(8) for lane in range(64))
    assert seen == list(range(512))
    for c in range(7):  # where two chunks meet, the neighbouring groups sit in the SAME lane (the DPP shift's `old` operand)
        end = 63 if c % 2 == 0 else 0
        assert group_of(c + 1, end) == group_of(c, end) + 1
    for c in range(8):  # inside a chunk the successor is the next lane (even chunks) or the previous one (odd chunks)
        for lane in range(63):
            assert group_of(c, lane + 1) - group_of(c, lane) == (-1 if c & 1 else 1)


def test_default_bank_fits_the_chunk_budgets_and_its_weights_are_lines():
    p = O.OracleParams()
    fb, act, w0, w1, start = _bank(p)
    M = fb.shape[1]
    assert M == 512
    lanes = np.zeros(64, int)
    for c in range(8):
        for lane in range(64):
            g = group_of(c, lane)
            n = start[g + 1] - start[g]
            assert 0 < n <= 2 * PAIRS[c], (c, lane, n)
            lanes[lane] += n
    assert lanes.sum() == len(act) == 4000 and lanes.max() <= 67 and lanes.min() >= 60  # every lane within 8 % of 62.5 bins
    for g in range(M):
        for w in (w0, w1):
            seg = w[start[g]:start[g + 1]]
            a, s = _line(seg)
            assert np.abs(a.astype(np.float64) + s.astype(np.float64) * np.arange(len(seg)) - seg).max() <= 1e-6
    # a bank whose weights are NOT lines per group must be refused (mel_scale_type "slaney": the group kernels keep it)
    fbs, acts, w0s, w1s, starts = _bank(O.OracleParams(mel_scale_type="slaney"))
    worst = 0.0
    for g in range(M):
        seg = w0s[starts[g]:starts[g + 1]]
        if len(seg):
            a, s = _line(seg)
            worst = max(worst, float(np.abs(a.astype(np.float64) + s.astype(np.float64) * np.arange(len(seg)) - seg).max()))
    assert worst > 1e-3


def _shift_from_prev(old, src):  # wave_shr:1 - lane i takes src of lane i - 1, lane 0 keeps `old`
    out = np.empty_like(src)
    out[1:] = src[:-1]
    out[0] = old[0]
    return out


def _shift_from_next(old, src):  # wave_shl:1 - lane i takes src of lane i + 1, lane 63 keeps `old`
    out = np.empty_like(src)
    out[:-1] = src[1:]
    out[-1] = old[-1]
    return out


@pytest.mark.parametrize("norm", [None, "slaney"])
def test_wave_formulation_tracks_the_oracle(norm):
    """norm None: unit form in the upper four chunks (B = S - A, one weight line); "slaney" (area-normalised triangles: still lines
    per group, but the two weights of a bin no longer sum to one): both weight lines in every chunk."""
    p = O.OracleParams(max_mel_iters=120, mel_scale_norm=norm)
    uf = norm is None
    fb, act, w0, w1, start = _bank(p)
    F, M = fb.shape
    T = 4
    gen = torch.Generator().manual_seed(5)
    mel = (torch.rand(1, M, T, generator=gen) ** 3 * 2e7).numpy().astype(f32)
    spec0 = torch.rand(1, T, F, generator=gen)
    ref = O.inverse_mel_scale_sgd(torch.from_numpy(mel), p, spec0=spec0).numpy()[0]
    SC = f32(2.0 ** -60)
    nl = f32(-(f32(0.1) * f32(-2.0 / T)))
    mom = f32(0.9)
    lane = np.arange(64)
    G = [np.array([group_of(c, l) for l in lane]) for c in range(8)]
    A0 = [np.array([_line(w0[start[g]:start[g + 1]]) for g in G[c]], f32) for c in range(8)]  # [64][2] = (a0, s0)
    A1 = [np.array([_line(w1[start[g]:start[g + 1]]) for g in G[c]], f32) for c in range(8)]
    out = np.zeros((F, T), f32)
    zero = np.zeros(64, f32)
    for t in range(T):
        x, msk, m0 = [], [], []
        for c in range(8):
            n = start[G[c] + 1] - start[G[c]]
            assert n.min() >= 2 * FULL[c]  # the leading pairs every lane fills need no mask
            i = np.arange(2 * PAIRS[c])[None, :]
            ok = i < n[:, None]
            fidx = np.where(ok, start[G[c]][:, None] + i, 0)
            x.append(np.where(ok, spec0[0, t].numpy()[fidx] * SC, 0).astype(f32))
            msk.append(ok.astype(f32))
            m0.append((mel[0, G[c], t] * SC).astype(f32))
        Cc = [np.zeros(64, f32) for _ in range(8)]  # the momentum buffer of a group's bin i is the LINE Cc + Gg i
        Gg = [np.zeros(64, f32) for _ in range(8)]
        for _ in range(p.max_mel_iters):
            A, B = [], []
            for c in range(8):
                xs = x[c]
                S = xs[:, 0:2].copy()
                Q = xs[:, 2:4].copy()
                S = (S + xs[:, 2:4]).astype(f32)
                for pp in range(2, PAIRS[c]):
                    S = (S + xs[:, 2 * pp:2 * pp + 2]).astype(f32)
                    Q = (f32(pp) * xs[:, 2 * pp:2 * pp + 2] + Q).astype(f32)
                q = (f32(2) * (Q[:, 0] + Q[:, 1]).astype(f32) + S[:, 1]).astype(f32)
                s = (S[:, 0] + S[:, 1]).astype(f32)
                a = (A0[c][:, 1] * q + (A0[c][:, 0] * s).astype(f32)).astype(f32)
                b = (s - a).astype(f32) if uf and c >= 4 else (A1[c][:, 1] * q + (A1[c][:, 0] * s).astype(f32)).astype(f32)
                A.append(a)
                B.append(b)
            d0 = []
            for c in range(8):
                old = B[c - 1] if c else zero
                bp = _shift_from_next(old, B[c]) if c & 1 else _shift_from_prev(old, B[c])
                d0.append(((m0[c] - A[c]).astype(f32) - bp).astype(f32))
            n0 = [(nl * d0[c]).astype(f32) for c in range(8)]  # residuals in units of the step -lr g: everything below is linear in them
            n1 = []
            for c in range(8):
                old = n0[c + 1] if c < 7 else zero
                n1.append(_shift_from_prev(old, n0[c]) if c & 1 else _shift_from_next(old, n0[c]))
            for c in range(8):
                if uf and c >= 4:
                    dd = (n0[c] - n1[c]).astype(f32)
                    cc = (dd * A0[c][:, 0] + n1[c]).astype(f32)
                    st = (dd * A0[c][:, 1]).astype(f32)
                else:
                    cc = (n1[c] * A1[c][:, 0] + (n0[c] * A0[c][:, 0]).astype(f32)).astype(f32)
                    st = (n1[c] * A1[c][:, 1] + (n0[c] * A0[c][:, 1]).astype(f32)).astype(f32)
                Cc[c] = (mom * Cc[c] + cc).astype(f32)  # torch.optim.SGD: buf.mul_(momentum).add_(grad), for the whole line at once
                Gg[c] = (mom * Gg[c] + st).astype(f32)
                vx, h = Cc[c], Gg[c]
                base = np.stack([vx, (vx + h).astype(f32)], 1)
                w2 = (h + h).astype(f32)[:, None]
                for pp in range(PAIRS[c]):
                    v = base if pp == 0 else (f32(pp) * w2 + base).astype(f32)
                    sl = slice(2 * pp, 2 * pp + 2)
                    if pp < FULL[c]:
                        x[c][:, sl] = np.clip((x[c][:, sl] + v).astype(f32), 0, 1)
                    else:
                        x[c][:, sl] = np.clip((v * msk[c][:, sl] + x[c][:, sl]).astype(f32), 0, 1)
        for c in range(8):
            n = start[G[c] + 1] - start[G[c]]
            for l in range(64):
                out[start[G[c][l]]:start[G[c][l]] + n[l], t] = x[c][l, :n[l]] / SC
                assert not x[c][l, n[l]:].any()  # padding slots stayed at exactly zero through every step
    rel = float(np.linalg.norm(out[act] - ref[act]) / np.linalg.norm(ref[act]))
    print(f"wave formulation, mel_scale_norm={norm!r}: rel-L2 {rel:.2e} vs the oracle after {p.max_mel_iters} steps")
    assert rel <= 2e-6
