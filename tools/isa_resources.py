"""
Register / scratch / LDS use of every kernel of the library, read from the ISA hipcc emits for gfx950 (no GPU needed):

    python tools/isa_resources.py [file.hip ...]     -> one line per kernel, and profiles/isa_resources_latest.json

A kernel with private_segment_fixed_size > 0 spills (or indexes a local array dynamically): the round-4 review listed the
48 kHz row-family kernels and the one-tile Griffin-Lim kernels; tests/test_isa_resources.py holds the hot kernels' frame loops
free of scratch traffic (and the forward kernel's per-frame load count).
"""
import json, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "riffusion-hobby_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include")]  # as __graft_entry__.build()


def kernels_of(src: str, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, *extra, "-S", "--cuda-device-only", "-o", out, src],
                       check=True, capture_output=True, cwd=CSRC)
        asm = open(out).read()
    res = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        body = m.group(2)
        def field(k):
            mm = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
            return int(mm.group(1)) if mm and mm.group(1).isdigit() else None
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        start = asm.index(m.group(1) + ":")
        code = asm[start:asm.index("s_endpgm", start)].split("\n")
        n_instr = sum(1 for l in code if re.match(r"\s+[sv]_|\s+ds_|\s+buffer_|\s+global_|\s+scratch_|\s+flat_", l))
        n_scratch = sum(1 for l in code if re.match(r"\s+scratch_", l))
        res.append({"file": os.path.basename(src), "kernel": name, "vgpr": field("next_free_vgpr"), "agpr_offset": field("accum_offset"),
                    "scratch_bytes": field("private_segment_fixed_size"), "static_lds_bytes": field("group_segment_fixed_size"),
                    "instructions_static": n_instr, "scratch_instructions_static": n_scratch})
    return res


def main(argv):
    srcs = [os.path.join(CSRC, a) if not os.path.isabs(a) else a for a in argv] or sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(kernels_of, srcs) for r in rs]
    for r in rows:
        print(f"{r['file']:16s} vgpr {r['vgpr']:>4} scratch {r['scratch_bytes']:>4} B ({r['scratch_instructions_static']:>3} of {r['instructions_static']:>5} instructions) lds {r['static_lds_bytes']:>6}  {r['kernel'][:130]}")
    if not argv:
        with open(os.path.join(ROOT, "profiles", "isa_resources_latest.json"), "w") as fh:
            json.dump(rows, fh, indent=1)
    return rows


if __name__ == "__main__":
    main(sys.argv[1:])
