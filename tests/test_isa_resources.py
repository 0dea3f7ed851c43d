"""
What the compiler made of the hot kernels, read from the ISA hipcc emits for gfx950 (no GPU needed; tools/isa_resources.py and
tools/isa_mix.py do the reading).  A frame loop that touches scratch memory has lost the register allocation it was written for:
the round-4 review found spills nobody had looked for.  Held here:
  * the default-bank forward kernel, the InverseMelScale wave kernel: no scratch at all;
  * the 44.1 kHz Griffin-Lim kernels: no scratch instruction inside the frame loop (their 16 B belong to the per-segment prologue);
  * the 48 kHz row-family kernels: at most one scratch store inside the frame loop (the forward kernel parks one prefetched
    twiddle of the NEXT frame across its mel loop; what else is left are re-loads of loop-invariant addresses).
One compile of four translation units in parallel (~40 s).
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "riffusion-hobby_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILES = ("rfx_stft.hip", "rfx_gl.hip", "rfx_imel.hip", "rfx_fam.hip")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


@pytest.fixture(scope="module")
def asm():
    import isa_resources

    def one(name):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run([HIPCC, *isa_resources.FLAGS, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, name)], check=True,
                           capture_output=True, cwd=CSRC)
            return open(out).read()

    with ThreadPoolExecutor(len(FILES)) as ex:
        return dict(zip(FILES, ex.map(one, FILES)))


def kernel_symbols(text, pattern):
    return [m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S+)", text) if re.search(pattern, m.group(1))]


def scratch_bytes(text, symbol):
    body = re.search(r"\.amdhsa_kernel " + re.escape(symbol) + r"(.*?)\.end_amdhsa_kernel", text, re.S).group(1)
    return int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body).group(1))


def frame_loop(text, symbol):
    """the instruction lines of the kernel's frame loop, found the way tools/isa_mix.py finds it"""
    s = text.index(symbol + ":")
    body = text[s:text.index("s_endpgm", s)].split("\n")
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    best = max(loops, key=lambda ab: ab[1] - ab[0])
    while True:
        inner = [ab for ab in loops if best[0] <= ab[0] and ab[1] <= best[1] and ab != best and 2 * (ab[1] - ab[0]) >= best[1] - best[0]]
        if not inner:
            break
        best = max(inner, key=lambda ab: ab[1] - ab[0])
    return body[best[0]:best[1] + 1]


def test_forward_and_sgd_kernels_hold_no_scratch(asm):
    fwd = kernel_symbols(asm["rfx_stft.hip"], r"stft_mel2_kernelILj2031647ELb1")
    assert len(fwd) == 1
    assert scratch_bytes(asm["rfx_stft.hip"], fwd[0]) == 0
    # ... and fetches 29 values per thread and frame, 30 in the wave that carries the second filters (66 until round 5, when those fetches - not the exchange - turned out to be what
    # the mel half of the kernel cost: profiles/r05_forward_ablation.txt)
    loads = [l for l in frame_loop(asm["rfx_stft.hip"], fwd[0]) if re.match(r"\s+buffer_load", l)]
    assert 26 <= len(loads) <= 30, len(loads)
    wave = kernel_symbols(asm["rfx_imel.hip"], r"imel_wave_kernel")
    assert len(wave) == 2
    for sym in wave:
        assert scratch_bytes(asm["rfx_imel.hip"], sym) == 0, sym


def test_griffinlim_frame_loops_touch_no_scratch(asm):
    syms = kernel_symbols(asm["rfx_gl.hip"], r"gl_iter_kernelILi[12]E")
    assert len(syms) == 2
    for sym in syms:
        loop = frame_loop(asm["rfx_gl.hip"], sym)
        assert len(loop) > 1000  # (the frame loop, not some small inner loop)
        assert not [l for l in loop if re.match(r"\s+scratch_", l)], sym


def test_48khz_frame_loops_keep_their_butterflies_out_of_scratch(asm):
    syms = kernel_symbols(asm["rfx_fam.hip"], r"fam_(gl_kernelILi1E|fwd_kernelILi[012]E)Li24ELi20E")
    assert len(syms) == 4, syms
    for sym in syms:
        loop = frame_loop(asm["rfx_fam.hip"], sym)
        assert len(loop) > 1000
        stores = [l for l in loop if re.match(r"\s+scratch_store", l)]
        loads = [l for l in loop if re.match(r"\s+scratch_load", l)]
        print(sym, "frame loop:", len(loop), "lines,", len(loads), "scratch loads,", len(stores), "scratch stores; kernel", scratch_bytes(asm["rfx_fam.hip"], sym), "B")
        assert len(stores) <= 1 and len(loads) <= 8, sym
        assert scratch_bytes(asm["rfx_fam.hip"], sym) <= 48, sym  # (88 - 132 B before the radix-24 butterfly was streamed)
