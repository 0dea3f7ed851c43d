"""
CPU check of the arithmetic the InverseMelScale group kernel uses since round 3 (csrc/rfx_imel.hip): state scaled by 2^-60 so
that the clamp at zero is the FMA's [0, 1] output clamp, and the gradient of the long groups in unit form
d0 w0 + d1 w1 = d1 + (d0 - d1) w0, valid where a bin's two filterbank weights sum to one.  A float32 numpy restatement of the
kernel's step (group sums A / B, residuals from the neighbours' sums, momentum buffer in units of the gradient scale) is run
next to the oracle's SGD (torchaudio 0.13 InverseMelScale as restated in oracle/riffusion_oracle.py): both forms must sit at
the same distance from it.  The kernel itself is pinned on the GPU (tests/test_gpu_full_parity.py, rel-L2 9e-8 at T = 512).
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
    return fb, act, first, w0, w1


def test_weights_of_the_long_groups_sum_to_one_for_the_default_bank_only():
    """What rfx_plan_create checks before it enables the unit form (rfx_plan_imel_unit_form)."""
    for norm, expect in ((None, True), ("slaney", False)):
        p = O.OracleParams(mel_scale_norm=norm)
        fb, act, first, w0, w1 = _bank(p)
        M = fb.shape[1]
        ok = True
        for f in act:
            g = first[f]
            if g < M - 256:
                continue
            ok = ok and ((w1[f] == 0) if g == M - 1 else abs(float(w0[f]) + float(w1[f]) - 1.0) <= 1e-6)
        assert ok == expect, norm


@pytest.mark.parametrize("unit_form", [False, True])
def test_scaled_unit_form_step_tracks_the_oracle(unit_form):
    p = O.OracleParams(max_mel_iters=120)
    fb, act, first, w0, w1 = _bank(p)
    F, M = fb.shape
    T = 6
    g = torch.Generator().manual_seed(5)
    mel = (torch.rand(1, M, T, generator=g) ** 3 * 2e7).numpy().astype(f32)
    spec0 = torch.rand(1, T, F, generator=g)
    ref = O.inverse_mel_scale_sgd(torch.from_numpy(mel), p, spec0=spec0).numpy()[0]
    grp = first[act]
    u, v = w0[act], w1[act]
    hi = grp >= M - 256
    SC = f32(2.0 ** -60)
    lrg = f32(0.1) * f32(-2.0 / T)   # the step in units of the gradient scale -2 / (C T): spec = fma(-lr g, buf'', spec)
    mom = f32(0.9)
    out = np.zeros((F, T), f32)
    for t in range(T):
        s = (spec0[0, t].numpy()[act] * SC).astype(f32)
        m = (mel[0, :, t] * SC).astype(f32)
        buf = np.zeros_like(s)
        for _ in range(p.max_mel_iters):
            A = np.zeros(M + 1, f32)
            B = np.zeros(M + 1, f32)
            np.add.at(A, grp, u * s)
            np.add.at(B, grp, v * s)
            Bm1 = np.concatenate([[f32(0)], B[:-1]]).astype(f32)
            d0 = np.concatenate([(m - A[:M]) - Bm1[:M], [f32(0)]]).astype(f32)
            d1 = np.zeros(M + 1, f32)
            d1[: M - 1] = (m[1:] - A[1:M]) - B[: M - 1]
            d1[M - 1] = 0.0  # no filter M: forced in the unit form, multiplied by w1 == 0 otherwise
            D0, D1 = d0[grp], d1[grp]
            full = ((mom * buf + D0 * u) + D1 * v).astype(f32)
            if unit_form:
                unit = ((mom * buf + D1) + (D0 - D1) * u).astype(f32)
                bn = np.where(hi, unit, full)
            else:
                bn = full
            s = np.clip(s - lrg * bn, 0, 1).astype(f32)  # the [0, 1] clamp of the scaled state
            buf = bn
        out[act, t] = s / SC
    rel = float(np.linalg.norm(out[act] - ref[act]) / np.linalg.norm(ref[act]))
    print(f"unit_form={unit_form}: rel-L2 {rel:.2e} vs the oracle after {p.max_mel_iters} steps")
    assert rel <= 2e-6
