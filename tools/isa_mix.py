"""
Instruction mix of the frame loop of the dominant kernels, counted in the ISA hipcc emits for gfx950 (no GPU needed):
rfx::gl_iter_kernel<2> (csrc/rfx_gl.hip), rfx::stft_mel2_kernel (csrc/rfx_stft.hip) and the SGD step loop of rfx::imel_wave_kernel
(csrc/rfx_imel.hip).  Since round 4 most butterfly
arithmetic is packed (v_pk_*_f32 on (re, im) pairs): a packed instruction is ONE issue slot but occupies the SIMD's fp32 pipe
for 4 cycles where a plain one takes 2 (MI355X_MICROARCH.md; quarter-rate v_rsq / v_sqrt / v_rcp: 8), so the count of wave
instructions no longer measures the VALU time a kernel needs.  bench.py's `binding` roofline multiplies the PMC count of VALU
instructions per launch by the average pipe cycles per VALU instruction found here.

    python tools/isa_mix.py            -> profiles/isa_mix_latest.json (and a line per kernel on stdout)
"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "riffusion-hobby_amd", "csrc")
KERNELS = [("rfx_gl.hip", "_ZN3rfx14gl_iter_kernelILi2EEEvNS_6GlArgsE", "rfx::gl_iter_kernel<2>"),
           ("rfx_stft.hip", "_ZN3rfx16stft_mel2_kernelILj2031647ELb1EEEvNS_11StftMelArgsE", "rfx::stft_mel2_kernel<0x1F001F>"),
           # one trip = one SGD step of one frame (one wave); its 16 DPP wave shifts count as plain here (measured 2.1 ns each)
           ("rfx_imel.hip", "_ZN3rfx16imel_wave_kernelILb1EEEvNS_8ImelArgsE", "rfx::imel_wave_kernel")]
TRANS = {"v_rsq_f32_e32", "v_rcp_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e64", "v_rcp_f32_e64", "v_sqrt_f32_e64"}


def loop_mix(asm: str, symbol: str, once_every: int = 16):
    s = asm.index(symbol + ":")
    body = asm[s:asm.index("s_endpgm", s)].split("\n")
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # the frame loop: the longest backward branch - or, when that one is an outer loop around it (round 5: gl_iter_kernel walks the
    # segments of a run), the loop nested inside it that makes up most of its body
    # Round 6: "longest" is not enough - branchy epilogues are laid out with their taken paths behind the fall-through code, and the
    # jumps back look like loops longer than the real one.  The loop wanted is the one that holds the packed arithmetic: among the
    # backward branches with the most v_pk_* instructions, the shortest.
    def packed(ab):
        return sum(1 for l in body[ab[0]:ab[1] + 1] if re.match(r"\s+v_pk", l))
    most = max(packed(ab) for ab in loops)
    best = min((ab for ab in loops if packed(ab) == most), key=lambda ab: ab[1] - ab[0])
    # a block between the markers `; RFX_ONCE_PER_GROUP_BEGIN / _END` (rfx_gl.hip: the group boundary inside a run) runs once per
    # `once_every` trips: its instructions count with that weight
    mix = collections.Counter()
    weight = 1.0
    for l in body[best[0]:best[1] + 1]:
        if "RFX_ONCE_PER_GROUP_BEGIN" in l:
            weight = 1.0 / once_every
        elif "RFX_ONCE_PER_GROUP_END" in l:
            weight = 1.0
        m = re.match(r"\s+([a-z_0-9]+)\s", l)
        if not m:
            continue
        k = m.group(1)
        key = ("valu_packed" if k.startswith("v_pk") else "valu_quarter_rate" if k in TRANS else "valu_plain" if k.startswith("v_") else
               "lds" if k.startswith("ds_") else "vmem" if k.startswith(("buffer", "global")) else "scratch" if k.startswith("scratch") else
               "s_nop" if k == "s_nop" else "s_waitcnt" if k == "s_waitcnt" else "s_barrier" if k == "s_barrier" else "salu")
        mix[key] += weight
    return {k: round(v, 1) if v != int(v) else int(v) for k, v in mix.items()}


def main():
    out = {"note": "instructions of ONE trip of the frame loop per wave (one frame), counted in `hipcc -S --offload-arch=gfx950` output; "
                   "pipe cycles per wave instruction: plain fp32 2, packed fp32 4, quarter-rate transcendental 8 (MI355X_MICROARCH.md)",
           "kernels": {}}
    try:
        out["git"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        out["git"] = "not recorded"
    with tempfile.TemporaryDirectory() as tmp:
        for src, sym, name in KERNELS:
            s_path = os.path.join(tmp, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"),
                            "--cuda-device-only", "-S", "-o", s_path, os.path.join(CSRC, src)], check=True, capture_output=True)
            mix = loop_mix(open(s_path).read(), sym)
            valu = mix.get("valu_plain", 0) + mix.get("valu_packed", 0) + mix.get("valu_quarter_rate", 0)
            cycles = 2 * mix.get("valu_plain", 0) + 4 * mix.get("valu_packed", 0) + 8 * mix.get("valu_quarter_rate", 0)
            out["kernels"][name] = {"loop_mix": mix, "valu_instructions": valu, "valu_pipe_cycles": cycles, "pipe_cycles_per_valu_instruction": round(cycles / valu, 4),
                                    "issue_slots": sum(mix.values())}
            print(name, out["kernels"][name])
    json.dump(out, open(os.path.join(ROOT, "profiles", "isa_mix_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
