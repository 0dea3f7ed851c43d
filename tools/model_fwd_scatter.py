"""LDS bank model of stft_mel2_kernel's product exchange (no GPU): per wave-instruction conflict cycles of the scatter
(`ds_write_b32`: two groups of 32 lanes, bank = dword address mod 32) and of the segment reads (`ds_read_b128`: four groups of
16 lanes, bank = dword address mod 64, four banks per lane), for a given placement G[g] of the filter groups in the product
array (rfx_api.hip: `G`, `tab_at`, `seg`).  MI355X_MICROARCH.md, section LDS, is the rule book.

  python tools/model_fwd_scatter.py             # the plan's dense placement against the searched one
"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))

N_FFT, HOP, KQ = 17640, 441, 448
B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]


def bank_groups(fb: np.ndarray):
    """bin -> first filter (group), per-group counts and first bins, the way rfx_plan_create derives them."""
    F, M = fb.shape
    nz = fb > 0
    active = np.where(nz.any(1))[0]
    f_lo, f_hi = int(active[0]), int(active[-1]) + 1
    m0 = np.array([int(np.argmax(nz[f])) if nz[f].any() else -1 for f in range(F)])
    cnt = np.zeros(M, int)
    for f in range(f_lo, f_hi):
        cnt[m0[f]] += 1
    gfirst = f_lo + np.concatenate([[0], np.cumsum(cnt)[:-1]])
    return f_lo, f_hi, m0, cnt, gfirst


def slot_bin(k1, ka, kb):
    k = k1 + 40 * (ka + 21 * kb)
    return N_FFT - k if k > N_FFT // 2 else k


def scatter_instructions(f_lo, f_hi, m0):
    """One entry per (wave, kb) store instruction that has at least one contributing lane: list of (lane, bin)."""
    seen = set()
    owner = {}
    for k1 in range(21):
        for ka in range(21):
            for kb in range(21):
                b = slot_bin(k1, ka, kb)
                if b in seen:
                    continue
                seen.add(b)
                if f_lo <= b < f_hi:
                    owner[(k1, ka, kb)] = b
    out = []
    for w in range(7):
        for kb in range(21):
            lanes = []
            for lane in range(63):
                k1, ka = 3 * w + lane // 21, lane % 21
                if (k1, ka, kb) in owner:
                    lanes.append((lane, owner[(k1, ka, kb)]))
            if lanes:
                out.append((w, kb, lanes))
    return out


def write_cycles(instrs, pos, dump0=0):
    """LDS-array cycles of the scatter per frame (one array; the second array shifts every address by `arr`)."""
    total, ideal = 0, 0
    for w, kb, lanes in instrs:
        addr = {lane: dump0 + w * 64 + lane for lane in range(63)}  # non-contributing lanes: their dump float ([0, 448) since round 5)
        for lane, b in lanes:
            addr[lane] = pos[b]
        for grp in (range(0, 32), range(32, 63)):
            per_bank = {}
            for lane in grp:
                per_bank.setdefault(addr[lane] % 32, set()).add(addr[lane])
            total += max(len(s) for s in per_bank.values())
            ideal += 1
    return total, ideal


def read_cycles(G, cnt, arr, M):
    """LDS-array cycles of the first four 16-byte reads of both segments of every filter (thread m reads filter m)."""
    total, ideal = 0, 0
    for base_wave in range(0, 448, 64):
        for which in (0, 1):  # rising (array 1, group m-1) / falling (array 0, group m)
            for j in range(4):
                for grp in B128_GROUPS:
                    per_bank = {}
                    for l in grp:
                        m = base_wave + l
                        if m >= M or (which == 0 and m == 0):
                            a = 0
                        else:
                            a = (arr + G[m - 1] if which == 0 else G[m]) + 4 * j
                        for d in range(4):
                            per_bank.setdefault((a + d) % 64, set()).add(a + d)
                    total += max(len(s) for s in per_bank.values())
                    ideal += 1
    return total, ideal


def positions(G, gfirst, cnt):
    pos = {}
    for g in range(len(cnt)):
        for i in range(cnt[g]):
            pos[gfirst[g] + i] = G[g] + i
    return pos


def dense_placement(cnt):
    G = KQ + np.concatenate([[0], np.cumsum((cnt + 3) // 4 * 4)])  # the groups follow the dump floats (rfx_api.hip: G[0] = kQPad)
    return G[:-1], int(G[-1])


def searched_placement(cnt, gfirst, instrs, budget, seed=0, sweeps=6):
    """Groups keep their order; each may be pushed back by 0..7 quads (a gap of zero padding nobody reads).  Coordinate descent
    over the gaps, cost = scatter cycles; total length bounded by `budget` floats."""
    M = len(cnt)
    gaps = np.zeros(M, int)
    size = (cnt + 3) // 4 * 4

    def build(gaps):
        G = np.zeros(M, int)
        acc = 0
        for g in range(M):
            acc += 4 * gaps[g]
            G[g] = acc
            acc += size[g]
        return G, acc

    def cost(gaps):
        G, arr = build(gaps)
        if arr > budget:
            return 1 << 30
        return write_cycles(instrs, positions(G, gfirst, cnt))[0]

    best = cost(gaps)
    rng = np.random.default_rng(seed)
    for _ in range(sweeps):
        improved = False
        for g in rng.permutation(M):
            keep = gaps[g]
            for cand in range(8):
                if cand == keep:
                    continue
                gaps[g] = cand
                c = cost(gaps)
                if c < best:
                    best, keep, improved = c, cand, True
            gaps[g] = keep
        if not improved:
            break
    G, arr = build(gaps)
    return G, arr, gaps


if __name__ == "__main__":
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams

    p = SpectrogramParams()
    fb = _hip.mel_filterbank(p.n_fft // 2 + 1, p.min_frequency, p.max_frequency, p.num_frequencies, p.sample_rate, p.mel_scale_norm, p.mel_scale_type).numpy()
    f_lo, f_hi, m0, cnt, gfirst = bank_groups(fb)
    instrs = scatter_instructions(f_lo, f_hi, m0)
    G, arr = dense_placement(cnt)
    pos = positions(G, gfirst, cnt)
    w, wi = write_cycles(instrs, pos)
    r, ri = read_cycles(G, cnt, arr, len(cnt))
    print(f"bins {f_lo}..{f_hi}, groups {len(cnt)} (longest {cnt.max()}), store instructions with work: {len(instrs)}")
    print(f"dense placement: first array ends at {arr} floats; scatter {w} cycles (conflict-free {wi}); segment reads {r} (conflict-free {ri})")
