#!/usr/bin/env python3
"""
bench.py - headline benchmark of the spectrogram -> audio hot path on MI355X.

Workload (BASELINE.json configs[1]): B = 64 synthetic 512x512 mono spectrogram tiles (uint8, already
resident in HBM) -> image decode -> InverseMelScale (SGD 200) -> Griffin-Lim 32 -> int16 PCM, all in
HIP kernels through librfx.so.  One "step" = one such batch.  value = tiles/s over all ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --workload forward        # configs[2] alone: audio -> mel images (MFMA roofline)
    python bench.py --workload decode-stereo64 --gpus 8   # configs[3]: 512 stereo tiles, Griffin-Lim 64, sharded
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 outside a torch.distributed launch re-launches this script under
`torch.distributed.run` with N ranks (one process per GPU, backend nccl = RCCL); `n_gpus` on the JSON
line is the number of ranks that answered an RCCL all_reduce, not the flag.  Clips are independent:
the headline workload gives every GPU its own 64 tiles (weak scaling, no data-path collective; the only
collectives are the timing barrier and the max-over-ranks reduction); `decode-stereo64` shards a FIXED
batch of 512 stereo clips over the ranks with `riffusion.batch_shard.shard_range` (strong scaling).
The headline line also carries the configs[2] forward measurement as a `"forward"` object.

Extra objects on the JSON line:
  roofline     - the dominant kernel (rfx::gl_iter_kernel<2>, one Griffin-Lim iteration over the
                 batch): algorithmic bytes per launch = 20 B x 8821 bins x 512 frames x 64 tiles
                 (|S| 4 B + tprev 8 B read + 8 B written per bin and iteration)
                 divided by the launch duration measured with HIP events on the launch stream.
  cpu_baseline - the CPU oracle (oracle/riffusion_oracle.py, a torch-CPU port of the reference's
                 torchaudio path) timed on this host on ONE tile of the same workload.
  other_configs - `stereo64`: one GPU's share of BASELINE.json configs[3] (64 stereo clips, Griffin-Lim 64) through the product entry
                 point with its own canonical-bytes roofline; `batch_sweep`: the decode step at B = 16 ... 128 (the run partition of
                 rfx_griffinlim keeps every resident workgroup slot busy whatever B is).  After the timed region; context only.
  ms_per_step_min / median / max, shader_clock - the K timed steps one by one and the sclk the chip held meanwhile.
  other_sample_rates - the same decode step, and the forward path, at 48 kHz (n_fft 19200 / win 4800 / hop 480) and
                 22.05 kHz (8820 / 2205 / 220): the row-family kernels of csrc/rfx_fam.hip, measured after the
                 timed region; context only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "riffusion-hobby_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_BINS, N_FRAMES, N_MELS = 8821, 512, 512
HOP, SR = 441, 44100
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU per step")
    ap.add_argument("--iters", type=int, default=32, help="Griffin-Lim iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["decode", "forward", "decode-stereo64"], default="decode",
                    help="decode = the headline (configs[1], tiles -> audio); forward = configs[2] (audio -> mel images); "
                         "decode-stereo64 = configs[3] (512 stereo tiles, Griffin-Lim 64, sharded over the ranks)")
    ap.add_argument("--global-clips", type=int, default=512, help="decode-stereo64: clips in the sharded batch")
    ap.add_argument("--no-forward", action="store_true", help="skip the embedded configs[2] forward measurement")
    ap.add_argument("--no-other-rates", action="store_true", help="skip the embedded 48 kHz decode measurement (row-family Griffin-Lim engine)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the embedded configs[3] share (64 stereo clips, Griffin-Lim 64) and the batch-shape sweep")
    ap.add_argument("--host-input", action="store_true",
                    help="decode-stereo64: the timed calls take the tiles from HOST memory (the reference's API is host images in, host "
                         "audio out); uploads run chunk by chunk through pinned memory on a side stream (batch_shard.ChunkSource)")
    ap.add_argument("--n1-ms", type=float, default=None,
                    help="ms_per_step of the same command at --gpus 1: an N > 1 line then carries `vs_n1` (weak scaling: t1 / tN)")
    ap.add_argument("--gather", choices=["none", "rank0", "all"], default="none",
                    help="decode-stereo64: which clips a rank returns (own shard / everything on rank 0 / everything everywhere)")
    return ap.parse_args()


def relaunch_distributed(n_gpus: int) -> int:
    """`python bench.py --gpus N` (N > 1, no RANK in the environment): run N ranks of this script under
    torch.distributed.run on this node, one per GPU, rendezvous on 127.0.0.1."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    return subprocess.run(cmd, env=env).returncode


class stdout_to_stderr:
    """RCCL prints a start-up banner (version, hostname, library path) straight to file descriptor 1 when a communicator is
    created; the contract is ONE JSON line on stdout, so fd 1 points at stderr while the process group comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def profile_rev(summary: dict, path: str) -> str:
    """Commit the binaries were built from when the PMC summary `path` was collected.  The hash is written INTO the summary
    by the script that generates it (tools/profile_round.sh reads .git_rev, which tools/gpu.sh ships to the GPU box - the
    pushed snapshot has no .git); older summaries fall back to the last commit that touched the file."""
    if summary.get("git"):
        return str(summary["git"])
    import subprocess

    try:
        return subprocess.run(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, capture_output=True,
                              text=True, timeout=10).stdout.strip() or "not recorded"
    except Exception:
        return "not recorded"


def pmc_summary():
    """Committed rocprofv3 PMC summary of the dominant kernel (HBM bytes and VALU instructions per launch)."""
    path = os.path.join(ROOT, "profiles", "gl_iter_pmc_latest.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def isa_mix(kernel_prefix: str):
    """Committed instruction mix of a kernel's frame loop (tools/isa_mix.py -> profiles/isa_mix_latest.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "isa_mix_latest.json")) as f:
            d = json.load(f)
        for name, v in d.get("kernels", {}).items():
            if name.startswith(kernel_prefix):
                return dict(v, git=d.get("git", "not recorded"))
    except Exception:
        pass
    return {}


def valu_binding(valu_instr_per_launch: float, launch_ms: float, mix: dict, pmc: dict, pmc_src: str, rev: str):
    """
    The roofline that bounds the transform kernels since round 4: occupancy of the SIMDs' fp32 pipes.  A wave64 fp32
    instruction holds its SIMD-32's pipe for 2 cycles, a packed one (v_pk_*_f32, two results per lane) for 4, a quarter-rate
    transcendental for 8 (MI355X_MICROARCH.md), so pipe cycles per launch = VALU wave-instructions per launch (PMC
    SQ_INSTS_VALU) x the average pipe cycles per VALU instruction of the kernel's frame loop (tools/isa_mix.py), against
    1024 SIMDs x 2.4 GHz.  `frac_at_measured_clock` uses the clock the chip really ran the profiled launches at
    (GRBM_GUI_ACTIVE / 8 XCDs / launch time of the PMC run: it clocks down to its power budget under this load).
    (Rounds 1-3 priced wave-instructions per second against the issue rate: after packing, a third of the issue slots are gone
    at unchanged pipe time - the kernel was never bound by issue)
    """
    cpi = mix.get("pipe_cycles_per_valu_instruction")
    if not (valu_instr_per_launch and cpi):
        return None
    cycles = valu_instr_per_launch * cpi
    got = cycles / (launch_ms * 1e-3) / 1e9
    peak = 1024 * 2.4
    out = {"bound": "valu_pipe", "unit": "G SIMD-cycles/s", "valu_wave_instructions_per_launch": valu_instr_per_launch,
           "pipe_cycles_per_valu_instruction": cpi, "valu_pipe_cycles_per_launch": cycles, "achieved": round(got, 1), "peak": round(peak, 1),
           "frac": round(got / peak, 4), "loop_mix": mix.get("loop_mix"), "source": pmc_src, "git": rev, "isa_mix_git": mix.get("git")}
    gui, busy_ms = pmc.get("GRBM_GUI_ACTIVE_per_launch"), pmc.get("profiled_launch_ms")
    if gui and busy_ms:
        clk = gui / 8.0 / (busy_ms * 1e-3) / 1e9
        out["measured_clock_ghz"] = round(clk, 3)
        out["frac_at_measured_clock"] = round(got / (1024 * clk), 4)
    # Third reading (round 4, late): what the SIMDs were MEASURED to sustain per instruction kind with nothing else in their way
    # (tools/ubench/valu_rate.hip, profiles/r04_valu_rate_ubench.txt, four waves per SIMD): v_fma / v_add 1.47 ns, v_pk_* 2.28 ns,
    # quarter-rate 3.5 ns per wave-instruction and SIMD - a packed instruction costs 1.55 plain ones, not 2 x 2 cycles / 2.4 GHz.
    # `frac_of_measured_instruction_rate` = the launch's VALU work priced that way / launch time, averaged over the 1024 SIMDs (the
    # SIMDs that carry four of a CU's fourteen waves sit 14 % above the average).
    lm = mix.get("loop_mix") or {}
    n_loop = lm.get("valu_plain", 0) + lm.get("valu_packed", 0) + lm.get("valu_quarter_rate", 0)
    if n_loop:
        ns_per_instr = (1.47 * lm.get("valu_plain", 0) + 2.28 * lm.get("valu_packed", 0) + 3.5 * lm.get("valu_quarter_rate", 0)) / n_loop
        out["measured_ns_per_valu_instruction"] = round(ns_per_instr, 3)
        out["frac_of_measured_instruction_rate"] = round(valu_instr_per_launch / 1024.0 * ns_per_instr * 1e-6 / launch_ms, 4)
    return out


class ClockSampler:
    """Shader clock the chip sustains DURING the timed region: a host thread reads the driver's current-sclk file every few
    milliseconds while the steps run (no device work, no sync).  The kernels of this path are VALU-bound and the chip clocks down
    to its power budget under them (1.93 - 2.06 GHz depending on the box), which is most of the box-to-box spread
    of `value`.  Reports nothing (source: null) where the driver exposes no such file."""

    def __init__(self, index: int):
        import glob
        import threading

        self.samples, self.path, self.kind = [], None, None
        self._stop = threading.Event()
        self._thread = None
        # the node's other GPUs show up in sysfs too (busy with other people's work): take the card whose PCI address is this device's
        want = None
        try:
            pr = torch.cuda.get_device_properties(index or 0)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        mine = [c for c in cards if want and os.path.basename(os.path.realpath(os.path.dirname(c))) == want]
        if mine:
            self.path, self.kind = mine[0], f"pp_dpm_sclk of {want}"
        elif len(cards) == 1:
            self.path, self.kind = cards[0], "pp_dpm_sclk (the only card)"
        # socket power of the same device (round 5: is the clock a POWER limit?): hwmon power1_average / power1_input in microwatts,
        # power1_cap = the limit the driver enforces
        self.power, self.power_path, self.power_cap_w = [], None, None
        if self.path:
            dev_dir = os.path.dirname(self.path)
            for name in ("power1_average", "power1_input"):
                hit = sorted(glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*", name)))
                if hit:
                    self.power_path = hit[0]
                    break
            try:
                cap = sorted(glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*", "power1_cap")))
                if cap:
                    self.power_cap_w = int(open(cap[0]).read().strip()) / 1e6
            except Exception:
                pass

    def _read(self):
        try:
            with open(self.path) as f:
                for line in f:
                    if line.rstrip().endswith("*"):
                        return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def _read_power(self):
        try:
            with open(self.power_path) as f:
                return int(f.read().strip()) / 1e6
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v:
                self.samples.append(v)
            if self.power_path:
                w = self._read_power()
                if w:
                    self.power.append(w)
            self._stop.wait(0.004)

    def start(self):
        import threading

        if self.path:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=1.0)

    def summary(self):
        if not self.samples:
            return {"source": None, "note": "no current-sclk file readable on this host"}
        xs = sorted(self.samples)
        out = {"source": self.kind, "samples": len(xs), "mhz_min": xs[0], "mhz_median": xs[len(xs) // 2], "mhz_max": xs[-1],
               "note": "current sclk level of the driver, polled by a host thread during the timed steps"}
        if self.power:
            ws = sorted(self.power)
            out["socket_power_w"] = {"min": round(ws[0], 1), "median": round(ws[len(ws) // 2], 1), "max": round(ws[-1], 1), "cap": self.power_cap_w,
                                     "source": os.path.basename(self.power_path),
                                     "note": "the driver's averaged socket power during the timed steps (it lags: a 0.6 s timed region shows the ramp)"}
        return out


def strip_comments_and_space(src: str) -> str:
    """C / C++ source without comments and without white space (string and character literals kept as they are): what is left
    changes only when the code does."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == "/" and i + 1 < n and src[i + 1] == "/":
            while i < n and src[i] != "\n":
                i += 1
        elif c == "/" and i + 1 < n and src[i + 1] == "*":
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i : j + 1])
            i = j + 1
        elif c.isspace():
            i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def kernel_source_fingerprint(files=("rfx_gl.hip", "rfx_core.h", "rfx_frame.hip.h", "rfx_kernels.h")) -> str:
    """sha1 over the CODE of the sources the dominant kernel is compiled from - comments and white space stripped (round 6: two
    comment-only commits had flipped the `stale` flag of round 5's line) - the GPU box and the driver's checkout have no .git:
    the PMC summaries carry the fingerprint of the sources they were collected on (tools/profile_round.sh calls this function),
    and a line whose sources differ says `stale`."""
    import hashlib

    h = hashlib.sha1()
    for name in files:
        try:
            with open(os.path.join(ROOT, "riffusion-hobby_amd", "csrc", name), "r", encoding="utf-8", errors="replace") as f:
                h.update(strip_comments_and_space(f.read()).encode())
        except OSError:
            h.update(b"?")
        h.update(b"\0")
    return h.hexdigest()[:12]


def cpu_baseline(iters: int, threads_cap: int = 16, min_seconds: float = 10.0, max_tiles: int = 8):
    """
    The oracle (a torch-CPU port of the reference's torchaudio path) on the host cores, on a BOUNDED
    sample of the same workload: whole synthetic mono 512x512 tiles, one after the other (one call of
    the reference per tile), until at least `min_seconds` of CPU work have been timed (at most
    `max_tiles` tiles).  Threads are capped at 16, the fastest setting measured on the GPU box (one tile: 5.4 / 4.4 / 5.8 /
    9.5 / 17.6 s on 8 / 16 / 32 / 64 / 128 threads, profiles/r02_cpu_oracle_thread_scaling.txt; all 256: 424 s).
    """
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import riffusion_oracle as O  # checker / reported baseline only

    threads = max(1, min(os.cpu_count() or 1, threads_cap))
    torch.set_num_threads(threads)
    p = O.OracleParams(num_griffin_lim_iters=iters)
    rng = np.random.default_rng(20240807)
    g = torch.Generator().manual_seed(1234)
    t_imel = t_gl = 0.0
    n = 0
    t_start = time.time()
    while n < max_tiles and (n == 0 or time.time() - t_start < min_seconds):
        tile = rng.integers(0, 256, size=(N_MELS, N_FRAMES, 3), dtype=np.uint8)
        t0 = time.time()
        mel = torch.from_numpy(O.spectrogram_from_image_u8(tile, 0.25, False, 30e6))
        lin = O.inverse_mel_scale_sgd(mel, p, generator=g)
        t1 = time.time()
        wave = O.griffinlim(lin, p, generator=g)
        O.pcm16_from_waveform(wave.numpy(), normalize=True)
        t2 = time.time()
        t_imel += t1 - t0
        t_gl += t2 - t1
        n += 1
    total = t_imel + t_gl
    # BASELINE.json configs[0] on the CPU: the oracle on og_beat.png (image decode -> InverseMelScale -> Griffin-Lim -> int16), one call
    og_s = None
    try:
        from PIL import Image

        og = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "og_beat.png")).convert("RGB"))
        t0 = time.time()
        mel = torch.from_numpy(O.spectrogram_from_image_u8(og, 0.25, False, 30e6))
        wave = O.griffinlim(O.inverse_mel_scale_sgd(mel, p, generator=g), p, generator=g)
        O.pcm16_from_waveform(wave.numpy(), normalize=True)
        og_s = round(time.time() - t0, 3)
    except Exception as exc:  # (a baseline figure, not worth failing the line for)
        og_s = f"failed: {exc}"
    return {
        "og_beat_s": og_s,
        "value": round(n / total, 5),
        "unit": "tiles/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n} synthetic mono 512x512 tile(s), one reference call each: InverseMelScale SGD-200 {t_imel:.1f} s + "
        f"Griffin-Lim {iters} {t_gl:.1f} s (torch {torch.__version__} CPU, {threads} threads of {os.cpu_count()} logical cores)",
        "griffinlim_only_tiles_per_s": round(n / t_gl, 5),
    }


FORWARD_KERNEL_SOURCES = ("rfx_stft.hip", "rfx_core.h", "rfx_frame.hip.h", "rfx_kernels.h", "rfx_codec.hip")
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def executed_mel_k(n_mels: int = N_MELS) -> int:
    """K the mel GEMM really multiplies: 32-position blocks of the slot-ordered filterbank that hold a non-zero
    row (the list rfx_plan_create builds; positions follow slot_pos_f of csrc/rfx_core.h)."""
    from riffusion import _hip

    fb = np.asarray(_hip.mel_filterbank(N_BINS, 0.0, 10000.0, n_mels, SR, None, "htk"))
    live_bin = np.abs(fb).sum(1) > 0
    live = np.zeros(9408, bool)
    for k1 in range(21):
        for kp in range(441):
            k = k1 + 40 * kp
            q, kb = k1 * 21 + kp % 21, kp // 21
            qp = q + q // 63
            pos = ((kb >> 2) * 448 + qp) * 4 + (kb & 3) if kb < 20 else 20 * 448 + qp
            live[pos] = live_bin[k if k <= 8820 else 17640 - k]
    return 32 * int(live.reshape(-1, 32).any(1).sum())


def forward_cpu_baseline(threads_cap: int = 16, min_seconds: float = 10.0):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import riffusion_oracle as O  # reported baseline only

    threads = max(1, min(os.cpu_count() or 1, threads_cap))
    torch.set_num_threads(threads)
    p = O.OracleParams()
    rng = np.random.default_rng(20240807)
    n, t0 = 0, time.time()
    while n == 0 or time.time() - t0 < min_seconds:
        wave = torch.from_numpy((rng.standard_normal((1, HOP * (N_FRAMES - 1))) * 8000).astype(np.float32))
        mel = O.mel_amplitudes_from_waveform(wave, p)
        O.image_u8_from_spectrogram(mel.numpy(), 0.25)
        n += 1
    dt = time.time() - t0
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic waveforms of {HOP * (N_FRAMES - 1)} samples, one reference call each (torch.stft + dense "
                      f"mel matmul + uint8 quantisation), {dt:.1f} s, {threads} threads of {os.cpu_count()} logical cores"}


def forward_measure(args, world, rank, dev, distributed, with_cpu):
    """BASELINE.json configs[2]: B waveforms -> STFT -> MFMA mel GEMM -> uint8 image, per rank.
    Returns the result object on rank 0 (None elsewhere); every rank takes part in the timed region."""
    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    if distributed:
        import torch.distributed as dist
    plan = _hip.get_plan(SpectrogramParams(), dev)
    B, L = args.batch, HOP * (N_FRAMES - 1)
    rng = np.random.default_rng(20240807 + rank)
    wave = torch.from_numpy((rng.standard_normal((B, L)) * 8000).astype(np.float32)).to(dev)
    thr = torch.from_numpy(image_util.encode_thresholds(0.25)).to(dev)

    def step():  # spectrogram_image_from_audio's device half in one call (rfx_image_from_waveform)
        return plan.image_from_waveform(wave, False, thr)[0]

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert img.shape == (B, N_MELS, N_FRAMES, 3)
    if rank == 0:
        def timed(fn, reps=5):  # one call per measurement, drained before and after: no overlap between launches
            tot = 0.0
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                res = fn()
                e1.record()
                torch.cuda.synchronize(dev)
                tot += e0.elapsed_time(e1)
            return tot / reps, res

        mel_ms, mel = timed(lambda: plan.mel_from_waveform(wave))   # ONE launch: rfx::stft_mel_kernel
        enc_ms, _ = timed(lambda: plan.image_encode(mel, False, thr))
        one_call_ms, _ = timed(lambda: plan.image_from_waveform(wave, False, thr))  # what a timed step runs
        stft_ms, (mag, _, _) = timed(lambda: plan.stft(wave, want_mag=True, want_spec=False))  # standalone Spectrogram member
        lin = plan.unpack_magnitudes(mag, B, N_FRAMES)
        del mag
        melscale_ms, _ = timed(lambda: plan.mel_scale(lin))          # standalone MelScale member: pack + MFMA GEMM
        del lin
        images_per_s = world * B * args.steps / elapsed
        # the fused kernel reads the waveform and writes the mel amplitudes: its algorithmic HBM bytes are tiny, the
        # transform itself (fp32 VALU butterflies) is what takes the time
        alg_bytes = 4.0 * B * (L + N_MELS * N_FRAMES)
        dense_flop = B * (N_FRAMES * 2.5 * 17640 * np.log2(17640) + 2.0 * N_MELS * N_BINS * N_FRAMES)  # 4.94 GFLOP per tile
        pmc_src = "profiles/forward_pmc_latest.json"
        try:
            with open(os.path.join(ROOT, pmc_src)) as fh:
                fpmc = json.load(fh)
        except Exception:
            fpmc = {}
        fscale = B / float(fpmc.get("batch_tiles", 64))
        # what the fused kernel really computes per launch: the pruned real FFTs (2.5 N log2 N per frame) and the BANDED mel
        # projection (2 x 7 976 flops per frame); against the fp32 vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s)
        true_flop = B * N_FRAMES * (2.5 * 17640 * np.log2(17640) + 2.0 * 7976)
        if "stft_mel2_kernel" not in str(fpmc.get("kernel", "")):
            fpmc = {}  # a summary of another kernel (an older round's) says nothing about this one
        roof = {"kernel": "rfx::stft_mel2_kernel", "bound": "hbm", "achieved": round(alg_bytes / (mel_ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / (mel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic": fpmc.get("hbm_bytes_per_launch") * fscale if fpmc.get("hbm_bytes_per_launch") else None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(mel_ms, 4),
                "true_flops_per_launch": true_flop, "true_tflops": round(true_flop / (mel_ms * 1e-3) / 1e12, 2),
                "true_flops_frac": round(true_flop / (mel_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                "note": "fused framed transform -> |X| -> banded mel projection in one launch: the mel GEMM of the reference (4.6 of its "
                        "4.94 GFLOP per tile) multiplies a banded filterbank (7 976 non-zeros of 4.5 M) and is evaluated as such on chip, so "
                        "neither HBM nor the MFMA pipes bound this kernel: `frac` (HBM) is small by construction, true_flops_frac prices the "
                        "flops actually executed against the 157.3 TFLOP/s fp32 peak, and `binding` (occupancy of the fp32 VALU pipes) is the resource that bounds it; "
                        "avg_launch_ms is one launch between two stream drains (HIP events)"}
        fp_here = kernel_source_fingerprint(FORWARD_KERNEL_SOURCES)
        roof["from_profiles"] = {"source": pmc_src, "git": profile_rev(fpmc, pmc_src) if fpmc else None, "kernel_sources_of_summary": fpmc.get("src_sha"),
                                 "kernel_sources_here": fp_here, "stale": fpmc.get("src_sha") != fp_here}
        valu = fpmc.get("SQ_INSTS_VALU_per_launch")
        if valu:
            b = valu_binding(valu * fscale, mel_ms, isa_mix("rfx::stft_mel2_kernel"), fpmc, pmc_src, profile_rev(fpmc, pmc_src))
            if b:
                roof["binding"] = b
        k_exec = executed_mel_k()
        out = {
            "metric": "spectrogram_images_per_sec_forward",
            "value": round(images_per_s, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"batch={B} synthetic waveforms of {L} samples -> fused STFT / |X| / banded mel kernel -> uint8 image "
                                   "(BASELINE.json configs[2]); waveforms resident in HBM",
                       "batch_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"clips sharded over {world} GPU(s), no data-path collective"},
            "audio_sec_per_sec": round(images_per_s * L / SR, 1),
            "roofline": roof,
            "stages": {"image_from_waveform_ms": round(one_call_ms, 3), "stft_mel_fused_ms": round(mel_ms, 3), "image_encode_ms": round(enc_ms, 3),
                       "note": "a timed step is ONE rfx_image_from_waveform call (forward kernel with the maximum taken on the fly -> encoder reading "
                               "its frame-major scratch); the other two figures are the same work as the two calls rfx_mel_from_waveform (with the "
                               "transpose to (B, M, T)) and rfx_image_encode_u8 (with its pass for the maximum), byte-identical output"},
            # the reference's members are also available one by one (spectrogram_func, mel_scaler): unfused they cost
            "standalone_members": {"spectrogram_stft_ms": round(stft_ms, 3), "mel_scale_pack_plus_mfma_gemm_ms": round(melscale_ms, 3),
                                   "mfma_gemm_executed_k": k_exec,
                                   "note": "MelScale on (B, n_stft, T) input = layout pack + fp32-MFMA GEMM over the non-zero K blocks"},
        }
        if with_cpu:
            out["cpu_baseline"] = forward_cpu_baseline()
        return out
    return None


def og_beat_cli(dev, reps: int = 5):
    """BASELINE.json configs[0], the GPU half: what `python -m riffusion.cli image-to-audio --image seed_images/og_beat.png --audio x.wav`
    does (reference cli.py:73-95), in process: PNG file -> EXIF -> SpectrogramParams -> SpectrogramImageConverter ->
    audio_from_spectrogram_image(apply_filters=True) -> WAV bytes.  Median of `reps` after one warm-up (plan and arena exist, as
    in a server; the very first call of a process also builds the plan: reported separately)."""
    import io

    from PIL import Image

    from riffusion import cli
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter

    path = os.path.join(ROOT, "tests", "golden", "og_beat.png")

    def once():
        t0 = time.perf_counter()
        pil = Image.open(path)
        params = cli._params_from_image(pil)
        converter = SpectrogramImageConverter(params=params, device=str(dev))
        segment = converter.audio_from_spectrogram_image(pil, apply_filters=True)
        buf = io.BytesIO()
        segment.export(buf, format="wav")
        return time.perf_counter() - t0, len(buf.getvalue()), segment.duration_seconds

    first, nbytes, seconds = once()
    times = sorted(once()[0] for _ in range(reps))
    return {"ms": round(times[len(times) // 2] * 1e3, 3), "ms_min": round(times[0] * 1e3, 3), "ms_max": round(times[-1] * 1e3, 3),
            "first_call_ms": round(first * 1e3, 3), "reps": reps, "wav_bytes": nbytes, "audio_seconds": round(seconds, 3),
            "audio_sec_per_sec": round(seconds / times[len(times) // 2], 1),
            "workload": "riffusion.cli image-to-audio on tests/golden/og_beat.png (= the reference's seed_images/og_beat.png), in process: PNG decode -> "
                        "EXIF params -> audio_from_spectrogram_image(apply_filters=True) -> WAV bytes; mono 44.1 kHz, Griffin-Lim 32, one tile "
                        "(BASELINE.json configs[0] on the GPU; the oracle's CPU time for the same image is cpu_baseline.og_beat_s)"}


def stereo64_main(args, world, rank, dev, distributed):
    """BASELINE.json configs[3]: a fixed batch of `--global-clips` stereo 512x512 tiles, Griffin-Lim 64, sharded over the
    ranks through the product entry point (SpectrogramImageConverter.audio_from_spectrogram_images(group=..., gather=...)):
    every rank decodes shard_range(N, world, rank) and copies ITS clips to (pinned) host memory chunk by chunk behind the
    compute (`--gather none`, the default: no data-path collective); `--gather rank0` / `all` add one RCCL
    gather / all_gather_into_tensor of the int16 PCM.  After the timed region a stage leg splits one step into
    compute / device-to-host / collective and predicts the 8-GPU speed-up from those parts."""
    import torch.distributed as dist

    from riffusion import batch_shard
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams

    params = SpectrogramParams(stereo=True, num_griffin_lim_iters=64)
    conv = SpectrogramImageConverter(params, device=str(dev))
    N = args.global_clips
    rng = np.random.default_rng(20240807)  # the SAME full batch on every rank
    tiles_host = rng.integers(0, 256, size=(N, N_MELS, N_FRAMES, 3), dtype=np.uint8)  # pageable, as a caller's array would be
    tiles = torch.from_numpy(tiles_host).to(dev)
    timed_input = tiles_host if args.host_input else tiles
    group = dist.group.WORLD if distributed else None
    L = HOP * (N_FRAMES - 1)

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        conv.audio_from_spectrogram_images(timed_input, seed=w, group=group, gather=args.gather)  # same shapes as the timed steps (pinned blocks cached)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pcm = conv.audio_from_spectrogram_images(timed_input, seed=100 + k, group=group, gather=args.gather)
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    lo, hi = batch_shard.result_rows(N, group, args.gather)
    assert pcm.shape == (hi - lo, L, 2) and pcm.dtype == np.int16
    del pcm

    # ---- stage leg (outside the timed region, every rank takes part in the collective) ----------------------------------
    def wall(fn, reps=2):
        best = float("inf")
        for _ in range(reps):
            sync_all()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t)
        return best * 1e3

    mylo, myhi = batch_shard.shard_range(N, world, rank)
    n_mine = myhi - mylo
    compute_ms = wall(lambda: conv.audio_from_spectrogram_images(tiles, seed=7, group=group, gather="none", return_device=True))
    host_ms = wall(lambda: conv.audio_from_spectrogram_images(tiles, seed=7, group=group, gather="none"))
    host_in_ms = wall(lambda: conv.audio_from_spectrogram_images(tiles_host, seed=7, group=group, gather="none"))
    shard = torch.zeros((n_mine, L, 2), dtype=torch.int16, device=dev)
    pinned = torch.empty(shard.shape, dtype=torch.int16, pin_memory=True)
    d2h_raw_ms = wall(lambda: pinned.copy_(shard, non_blocking=True))
    gather_ms = {}
    if distributed:
        gather_ms["all_gather_into_tensor_ms"] = wall(lambda: batch_shard.gather_clips(shard, N, group))
        gather_ms["gather_rank0_ms"] = wall(lambda: batch_shard.gather_clips(shard, N, group, dst=0))
    # what one eighth of the batch costs on this GPU: the per-rank compute of an 8-GPU run of the same N
    n8 = -(-N // 8)
    eighth_ms = wall(lambda: conv.audio_from_spectrogram_images(tiles[:n8], seed=9, gather="none")) if world == 1 and N >= 8 else None
    del shard, pinned
    if rank != 0:
        return None
    ms_per_step = elapsed / args.steps * 1e3
    tiles_per_s = N * args.steps / elapsed
    pcm_mb = N * L * 2 * 2 / 1e6
    stages = {"compute_ms": round(compute_ms, 3), "compute_plus_d2h_ms": round(host_ms, 3),
              "d2h_exposed_ms": round(max(0.0, host_ms - compute_ms), 3),
              "host_in_host_out_ms": round(host_in_ms, 3), "h2d_exposed_ms": round(max(0.0, host_in_ms - host_ms), 3),
              "own_shard_in_mb": round(n_mine * N_MELS * N_FRAMES * 3 / 1e6, 1),
              "d2h_own_shard_raw_ms": round(d2h_raw_ms, 3), "own_shard_mb": round(n_mine * L * 4 / 1e6, 1), **{k: round(v, 3) for k, v in gather_ms.items()},
              "note": "rank 0, best of 2, wall clock between device syncs: compute = own shard with return_device=True; compute_plus_d2h = the same "
                      "call returning host PCM (chunk copies on a side stream behind the compute, pinned memory); host_in_host_out = the same call "
                      "fed from a pageable HOST array (uploads staged through pinned memory on a side stream, chunk k+1 under chunk k); d2h_own_shard_raw = the whole "
                      "shard in one un-overlapped pinned copy; collectives move the own shard (int16, as bytes) into one preallocated tensor"}
    out = {
        "metric": "stereo_spectrogram_tiles_per_sec_griffinlim64", "value": round(tiles_per_s, 2), "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{N} synthetic stereo 512x512 uint8 tiles -> image decode -> InverseMelScale SGD-200 -> Griffin-Lim 64 "
                               f"-> int16 PCM in host memory (BASELINE.json configs[3]), tiles {'in HOST memory' if args.host_input else 'resident in HBM'} when the timed region starts, "
                               f"clips sharded over the ranks with shard_range, gather={args.gather!r}"
                               + {"none": " (every rank returns its own clips; no data-path collective)",
                                  "rank0": f" (one RCCL gather of {pcm_mb:.0f} MB int16 PCM to rank 0)",
                                  "all": f" (one RCCL all_gather_into_tensor of {pcm_mb:.0f} MB int16 PCM)"}[args.gather],
                   "global_batch": N, "clips_per_gpu": -(-N // world), "griffin_lim_iters": 64, "gather": args.gather,
                   "parallelism": f"clips sharded over {world} GPU(s)"},
        "audio_sec_per_sec": round(tiles_per_s * L / SR, 1),
        "stages": stages,
    }
    if eighth_ms is not None:
        # 8-GPU step = the slowest rank's shard (N/8 clips incl. its overlapped D2H) + the collective, if one was asked for.
        # The collective is priced from the xGMI figures of MI355X_MICROARCH.md (7 links x ~153 GB/s per GPU, ~75 % achievable):
        # all_gather: every rank receives 7 shards over 7 links in parallel; gather: rank 0 receives 7 shards, one per link.
        link_gbs = 153.0 * 0.75
        shard_mb = n8 * L * 4 / 1e6
        coll_ms = 0.0 if args.gather == "none" else shard_mb / link_gbs
        full_d2h_ms = 0.0 if args.gather == "none" else d2h_raw_ms * (7.0 / 8.0)  # the 7 foreign shards cross PCIe after the collective
        pred8 = eighth_ms + coll_ms + full_d2h_ms
        out["predicted_speedup_8"] = {"value": round(ms_per_step / pred8, 2), "t1_ms": round(ms_per_step, 3), "t8_ms": round(pred8, 3),
                                      "per_rank_compute_plus_own_d2h_ms": round(eighth_ms, 3), "collective_ms_model": round(coll_ms, 3),
                                      "foreign_shards_d2h_ms_model": round(full_d2h_ms, 3),
                                      "note": f"t8 = this GPU running {n8} of the {N} clips through the same entry point (measured) + the gather={args.gather!r} "
                                              "collective at 0.75 x 153 GB/s per xGMI link (model) + the pinned copy of the foreign shards (measured rate); "
                                              "the driver's SCALE run is the measurement, this is the budget"}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_distributed(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ  # launched by torch.distributed.run (also exercised with one rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank if distributed else 0)
    torch.cuda.set_device(dev)
    n_ranks = 1
    if distributed:  # n_gpus = ranks that answered an RCCL all_reduce
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev)
            ones = torch.ones(1, dtype=torch.int32, device=dev)
            dist.all_reduce(ones)
            n_ranks = int(ones.item())
            torch.cuda.synchronize(dev)
        assert n_ranks == world

    def finish(out):
        if rank == 0 and out is not None:
            out["n_gpus"] = n_ranks
            print(json.dumps(out), flush=True)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()

    if args.workload == "forward":
        return finish(forward_measure(args, world, rank, dev, distributed, world == 1 and not args.no_cpu_baseline))
    if args.workload == "decode-stereo64":
        return finish(stereo64_main(args, world, rank, dev, distributed))

    from riffusion import _hip
    from riffusion.spectrogram_image_converter import SpectrogramImageConverter
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    params = SpectrogramParams(num_griffin_lim_iters=args.iters)
    conv = SpectrogramImageConverter(params, device=str(dev))
    plan = _hip.get_plan(params, dev)
    B, T = args.batch, N_FRAMES

    # synthetic tiles of SURVEY.md 8(d), a different seed per rank, resident in HBM before timing
    rng = np.random.default_rng(20240807 + rank)
    tiles = torch.from_numpy(rng.integers(0, 256, size=(B, N_MELS, T, 3), dtype=np.uint8)).to(dev)
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)
    gl_ws = torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, B, T), dtype=torch.uint8, device=dev)

    def step(seed):
        # THE PRODUCT ENTRY POINT (round 6): what a caller of the reference's SpectrogramImageConverter.audio_from_spectrogram_image
        # (spectrogram_image_converter.py:65-91) reaches for a batch - one Python call, one C call (rfx_audio_from_image_u8_ex),
        # uint8 tiles in HBM -> int16 PCM in HBM, scratch space from the plan's arena (no allocator traffic in steady state)
        return conv.audio_from_spectrogram_images(tiles, seed=seed, return_device=True, tiles_per_call=B)

    def four_calls(seed, launch_ms=None):
        # the same work as four C calls with a Griffin-Lim workspace hoisted out of the step: the headline of rounds 1-5, kept as
        # stages.c_abi_four_calls_ms (the product call must not cost more)
        mel = plan.image_decode(tiles, False, lut)                       # (B, 512, T) float32
        lin = plan.inverse_mel(mel, 1, seed=seed)                        # slots
        wave = plan.griffinlim(lin, B, T, args.iters, 0.99, seed=seed + 1, workspace=gl_ws, launch_ms=launch_ms)
        pcm, _ = plan.pcm16(wave, channels=1, normalize=True)            # (B, L, 1) int16
        return pcm

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        step(w)
    clock = ClockSampler(dev.index)
    # one event per step on the launch stream (no host sync inside the timed region): the spread of the K steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync_all()
    clock.start()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        pcm = step(100 + k)
        marks[k + 1].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    clock.stop()
    per_rank_ms = None
    if distributed:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # each rank's own wall time of the K steps (the value uses the MAX, per the contract)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)
    assert pcm.shape == (B, HOP * (T - 1), 1) and pcm.dtype == torch.int16 and int(pcm.abs().max()) > 30000
    step_series = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]  # in order: a trend is the chip warming up, an outlier is not
    per_step = sorted(step_series)

    # ---- roofline leg (outside the timed region): per-launch durations from HIP events on the stream
    roofline = None
    extra = {}
    if rank == 0:
        ms = (ctypes.c_float * (args.iters + 1))()
        four_calls(999, launch_ms=ms)
        steady = [ms[i] for i in range(2, args.iters + 1)] or [ms[-1]]
        avg_ms = sum(steady) / len(steady)
        alg_bytes = 20.0 * N_BINS * T * B
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        pmc = pmc_summary()
        pmc_src = "profiles/gl_iter_pmc_latest.json"
        # the counters were collected at the headline batch; per-launch figures scale with the frames a launch processes
        pmc_scale = (B * T) / float(pmc.get("batch_tiles", 64) * pmc.get("frames_per_tile", 512))
        traffic = pmc.get("hbm_bytes_per_launch")
        traffic = traffic * pmc_scale if traffic else traffic
        roofline = {
            "kernel": "rfx::gl_iter_kernel<2>",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": round(avg_ms, 4),
            "measured_in_this_run": ["avg_launch_ms", "achieved", "frac"],
            "avg_launch_ms_note": "mean of the steady-state launches of one rfx_griffinlim_timed call, each bracketed by its own pair of HIP events on the "
                                  "launch stream: consecutive launches cannot overlap their tails there, so iterations x avg_launch_ms exceeds stages.griffinlim_ms slightly",
            # counters cannot be collected from inside the timed process: these fields are read from the committed
            # rocprofv3 --pmc summary of the same kernel (re-collected whenever the kernel changes)
            "from_profiles": {"fields": ["traffic", "actual_hbm_gbs", "actual_hbm_frac", "binding.valu_wave_instructions_per_launch", "binding.measured_clock_ghz"],
                              "source": pmc_src, "git": profile_rev(pmc, pmc_src),
                              # the summary names the kernel sources it was collected on; `stale` = this checkout's differ
                              "kernel_sources_of_summary": pmc.get("src_sha"), "kernel_sources_here": kernel_source_fingerprint(),
                              "stale": pmc.get("src_sha") != kernel_source_fingerprint()},
            # `achieved` prices the kernel against the CANONICAL fused formulation of SURVEY 8(d) (|S| 4 B + tprev
            # 8 B read + 8 B written per bin and iteration: an HBM-bound kernel).  The shipped kernel applies the
            # momentum in the time domain (STFT linearity), streams only |S| (`traffic` is what it really moves)
            # and is bound by the fp32 VALU pipes: `binding` is the roofline that tracks progress from here.
            "formulation": "time-domain momentum: rebuilt - m*tprev = STFT(x_k - m*x_{k-1}); 4 B/bin/iteration streamed",
        }
        if traffic:
            roofline["actual_hbm_gbs"] = round(traffic / (avg_ms * 1e-3) / 1e9, 1)
            roofline["actual_hbm_frac"] = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        valu = pmc.get("SQ_INSTS_VALU_per_launch")
        valu = valu * pmc_scale if valu else valu
        if valu:
            b = valu_binding(valu, avg_ms, isa_mix("rfx::gl_iter_kernel<2>"), pmc, pmc_src, profile_rev(pmc, pmc_src))
            if b:
                roofline["binding"] = b
        # the three readings side by side: `frac` prices the canonical 20 B per bin and iteration of SURVEY 8(d) (bytes the shipped
        # formulation does not move), `actual_hbm_frac` what it really streams, `binding_frac` the fp32 pipes that bound it
        roofline["summary"] = {"frac_canonical_bytes": roofline["frac"], "actual_hbm_frac": roofline.get("actual_hbm_frac"),
                               "binding_frac_valu_pipe": (roofline.get("binding") or {}).get("frac"),
                               "binding_frac_of_measured_instruction_rate": (roofline.get("binding") or {}).get("frac_of_measured_instruction_rate")}
        # stage split of one step (events through torch on the current stream = the launch stream)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        evs[0].record()
        mel = plan.image_decode(tiles, False, lut)
        evs[1].record()
        lin = plan.inverse_mel(mel, 1, seed=5)
        evs[2].record()
        wave = plan.griffinlim(lin, B, T, args.iters, 0.99, seed=6, workspace=gl_ws)
        evs[3].record()
        plan.pcm16(wave, channels=1, normalize=True)
        evs[4].record()
        torch.cuda.synchronize(dev)
        names = ["image_decode_ms", "inverse_mel_ms", "griffinlim_ms", "pcm16_ms"]
        extra = {n: round(evs[i].elapsed_time(evs[i + 1]), 3) for i, n in enumerate(names)}
        extra["griffinlim_only_tiles_per_s"] = round(B / (extra["griffinlim_ms"] * 1e-3), 1)
        # the four-call form of the step (rounds 1-5's headline) and the product call, side by side, same box, same minute
        n4 = max(3, min(args.steps, 10))
        for fn, key in ((four_calls, "c_abi_four_calls_ms"), (step, "product_call_ms")):
            fn(0)
            torch.cuda.synchronize(dev)
            t4 = time.perf_counter()
            for k in range(n4):
                fn(200 + k)
            torch.cuda.synchronize(dev)
            extra[key] = round((time.perf_counter() - t4) / n4 * 1e3, 3)
        extra["note"] = ("the timed step is ONE SpectrogramImageConverter.audio_from_spectrogram_images(tiles, return_device=True) call = one "
                         "rfx_audio_from_image_u8_ex; c_abi_four_calls_ms is the same work as rfx_image_decode_u8 + rfx_inverse_mel + rfx_griffinlim + "
                         f"rfx_pcm16 with a workspace hoisted out of the step (the headline of rounds 1-5), product_call_ms the timed step again, {n4} steps each")
        arena = plan.arena
        extra["workspace_arena"] = {"buffers_allocated_in_this_process": arena.allocations, "idle_bytes": arena.idle_bytes()}

    ms_per_step = elapsed / args.steps * 1e3
    tiles_per_s = world * B * args.steps / elapsed
    out = None
    if rank == 0:
        out = {
            "metric": "spectrogram_tiles_per_sec_griffinlim32" if args.iters == 32 else f"spectrogram_tiles_per_sec_griffinlim{args.iters}",
            "value": round(tiles_per_s, 2),
            "unit": "tiles/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            # the K timed steps one by one (an event per step on the launch stream, rank 0): boxes of the pool differ by a few per
            # cent (the chip settles at 1.9 - 2.1 GHz under this load), steps on one box by far less - a regression smaller than the
            # spread between boxes shows up here, not in `value`
            "ms_per_step_min": round(per_step[0], 3),
            "ms_per_step_median": round(per_step[len(per_step) // 2], 3),
            "ms_per_step_max": round(per_step[-1], 3),
            "ms_per_step_max_over_min": round(per_step[-1] / per_step[0], 4),
            "ms_per_step_series": [round(v, 3) for v in step_series],
            "shader_clock": clock.summary(),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"batch={B} synthetic 512x512 mono uint8 tiles -> image decode -> InverseMelScale SGD-200 -> "
                f"Griffin-Lim {args.iters} -> int16 PCM (BASELINE.json configs[1]); tiles resident in HBM; one product call per step "
                "(SpectrogramImageConverter.audio_from_spectrogram_images -> rfx_audio_from_image_u8_ex)",
                "batch_per_gpu": B,
                "global_batch": world * B,
                "griffin_lim_iters": args.iters,
                "parallelism": f"clips sharded over {world} GPU(s), no data-path collective",
            },
            "audio_sec_per_sec": round(tiles_per_s * HOP * (T - 1) / SR, 1),
            "roofline": roofline,
            "stages": extra,
        }
        if per_rank_ms is not None:
            out["per_rank_ms"] = {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3), "by_rank": [round(v, 3) for v in per_rank_ms]}
        if args.n1_ms:
            out["vs_n1"] = {"n1_ms": args.n1_ms, "weak_scaling_efficiency": round(args.n1_ms / ms_per_step, 4),
                            "speedup": round(n_ranks * args.n1_ms / ms_per_step, 3),
                            "note": "weak scaling: every rank converts its own batch; speedup = N x t1 / tN"}
    # ---- the one-tile-per-request case (reference server.py:152-164): uint8 tile in HBM -> int16 PCM on the host, one call
    latency = None
    if rank == 0:
        latency = {}
        for name, stereo in (("mono", False), ("stereo", True)):
            conv1 = SpectrogramImageConverter(SpectrogramParams(stereo=stereo, num_griffin_lim_iters=args.iters), device=str(dev))
            one = tiles[:1]
            for r in range(3):
                conv1.audio_from_spectrogram_images(one, seed=r)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for r in range(10):
                conv1.audio_from_spectrogram_images(one, seed=10 + r)
            latency[name + "_ms"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
        latency["note"] = "SpectrogramImageConverter.audio_from_spectrogram_images on ONE 512x512 tile, D2H copy of the PCM included (small-batch Griffin-Lim kernels)"

    # ---- the same decode step at 48 kHz (cli.py:43 takes the sample rate from the input file; spectrogram_params.py:62-81 derives
    # n_fft 19200 / win 4800 / hop 480 from it): Griffin-Lim on the row-family kernels (csrc/rfx_fam.hip).  Context, not the headline.
    other = None
    if rank == 0 and world == 1 and not args.no_other_rates:
        other = {}
        for rate in (48000, 22050):
            p2 = SpectrogramParams(sample_rate=rate, num_griffin_lim_iters=args.iters)
            plan2 = _hip.get_plan(p2, dev)

            def step2(seed):  # the drop-in converter's calls: decode -> rfx_waveform_from_mel (InverseMelScale + Griffin-Lim) -> PCM
                mel2 = plan2.image_decode(tiles, False, lut)
                w2 = plan2.waveform_from_mel(mel2, 1, args.iters, 0.99, seed=seed)
                return plan2.pcm16(w2, channels=1, normalize=True)[0]

            step2(0)
            torch.cuda.synchronize(dev)
            n2 = max(2, min(args.steps, 5))
            t2 = time.perf_counter()
            for k in range(n2):
                pcm2 = step2(50 + k)
            torch.cuda.synchronize(dev)
            dt2 = (time.perf_counter() - t2) / n2
            lin2 = plan2.inverse_mel(plan2.image_decode(tiles, False, lut), 1, seed=49)  # (for the Griffin-Lim stage timer below)
            e2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e2[0].record()
            plan2.griffinlim(lin2, B, T, args.iters, 0.99, seed=3)
            e2[1].record()
            torch.cuda.synchronize(dev)
            # forward path at this rate: B waveforms of T frames -> mel amplitudes -> uint8 images
            wave2 = torch.randn(B, p2.hop_length * (T - 1), device=dev) * 8000
            thr2 = torch.from_numpy(image_util.encode_thresholds(0.25)).to(dev)
            for _ in range(2):
                plan2.image_from_waveform(wave2, False, thr2)
            e3 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e3[0].record()
            for _ in range(5):
                plan2.image_from_waveform(wave2, False, thr2)
            e3[1].record()
            torch.cuda.synchronize(dev)
            fwd_ms = e3[0].elapsed_time(e3[1]) / 5
            other[str(rate)] = {"tiles_per_s": round(B / dt2, 1), "ms_per_step": round(dt2 * 1e3, 3), "steps": n2,
                                "forward_images_per_s": round(B / fwd_ms * 1e3, 0), "forward_ms": round(fwd_ms, 3),
                                "griffinlim_ms": round(e2[0].elapsed_time(e2[1]), 3), "griffinlim_engine": plan2.griffinlim_engine,
                                "n_fft": p2.n_fft, "hop_length": p2.hop_length, "finite": bool(torch.isfinite(pcm2.float()).all()),
                                "workload": f"batch={B} synthetic 512x512 mono uint8 tiles -> audio at {rate} Hz, Griffin-Lim {args.iters}"}

    # ---- BASELINE.json configs[3], one GPU's share: 64 stereo clips (128 tile-channels), Griffin-Lim 64, through the product
    # entry point (uint8 tiles resident in HBM -> int16 PCM on the device), and the decode step over batch shapes that do not
    # divide the chip's 512 resident Griffin-Lim workgroups (cli.py:172-204 feeds arbitrary file counts per batch)
    other_cfg = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        from riffusion.spectrogram_image_converter import SpectrogramImageConverter as _SIC

        other_cfg = {}
        conv64 = _SIC(SpectrogramParams(stereo=True, num_griffin_lim_iters=64), device=str(dev))
        tiles64 = torch.from_numpy(np.random.default_rng(7).integers(0, 256, size=(64, N_MELS, T, 3), dtype=np.uint8)).to(dev)
        conv64.audio_from_spectrogram_images(tiles64, seed=0, return_device=True)
        torch.cuda.synchronize(dev)
        n64 = 3
        t64 = time.perf_counter()
        for k in range(n64):
            pcm64 = conv64.audio_from_spectrogram_images(tiles64, seed=1 + k, return_device=True)
        torch.cuda.synchronize(dev)
        dt64 = (time.perf_counter() - t64) / n64
        alg64 = 64 * 2 * (20.0 * 64 + 4) * N_BINS * T  # (20 n + 4) F T per tile-channel = 11.60 GB per stereo tile at n = 64
        other_cfg["stereo64"] = {
            "metric": "stereo_spectrogram_tiles_per_sec_griffinlim64", "value": round(64 / dt64, 2), "unit": "tiles/s", "ms_per_step": round(dt64 * 1e3, 3),
            "steps": n64, "finite": bool(torch.isfinite(pcm64.float()).all()),
            "workload": "64 synthetic stereo 512x512 uint8 tiles (128 tile-channels; one GPU's share of BASELINE.json configs[3]: 512 stereo tiles over "
                        "8 GPUs) -> SpectrogramImageConverter.audio_from_spectrogram_images -> int16 PCM, Griffin-Lim 64; tiles and PCM resident in HBM",
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": alg64, "achieved": round(alg64 / dt64 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg64 / dt64 / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "canonical bytes of SURVEY 8(d) for the WHOLE step (InverseMelScale and the codecs included in the time, not in the bytes)"}}
        with stdout_to_stderr():  # (the CLI prints its "no EXIF, using defaults" warning to stdout, like the reference's)
            other_cfg["og_beat_cli"] = og_beat_cli(dev)
        sweep = {}
        for Bs in (16, 48, 64, 65, 96, 100, 128):
            tl = torch.from_numpy(np.random.default_rng(Bs).integers(0, 256, size=(Bs, N_MELS, T, 3), dtype=np.uint8)).to(dev)
            ws = torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, Bs, T), dtype=torch.uint8, device=dev)

            def step_b(seed):
                m_ = plan.image_decode(tl, False, lut)
                l_ = plan.inverse_mel(m_, 1, seed=seed)
                w_ = plan.griffinlim(l_, Bs, T, args.iters, 0.99, seed=seed + 1, workspace=ws)
                return plan.pcm16(w_, channels=1, normalize=True)[0]

            step_b(0)
            torch.cuda.synchronize(dev)
            tb = time.perf_counter()
            for k in range(3):
                step_b(5 + k)
            torch.cuda.synchronize(dev)
            dtb = (time.perf_counter() - tb) / 3
            sweep[str(Bs)] = {"tiles_per_s": round(Bs / dtb, 1), "ms_per_step": round(dtb * 1e3, 3)}
            del tl, ws
        # ---- the reference's own round-trip test parameters (test/spectrogram_converter_test.py:46-53: 20 Hz .. 20 kHz, 512 filters):
        # groups of up to 54 bins - InverseMelScale ran on the general LDS kernel (169 ms per 64 tiles) until round 5's line-form
        # group kernel; the decode step of 64 mono tiles and the forward path with that bank
        p_fb = SpectrogramParams(min_frequency=20, max_frequency=20000, num_griffin_lim_iters=args.iters)
        plan_fb = _hip.get_plan(p_fb, dev)
        lut_fb = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)

        def step_fb(seed):
            m_ = plan_fb.image_decode(tiles, False, lut_fb)
            l_ = plan_fb.inverse_mel(m_, 1, seed=seed)
            w_ = plan_fb.griffinlim(l_, B, T, args.iters, 0.99, seed=seed + 1)
            return plan_fb.pcm16(w_, channels=1, normalize=True)[0], m_

        _, mel_fb = step_fb(0)
        torch.cuda.synchronize(dev)
        tfb = time.perf_counter()
        for k in range(3):
            pcm_fb, _ = step_fb(3 + k)
        torch.cuda.synchronize(dev)
        dtfb = (time.perf_counter() - tfb) / 3
        efb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        efb[0].record()
        plan_fb.inverse_mel(mel_fb, 1, seed=9)
        efb[1].record()
        torch.cuda.synchronize(dev)
        wave_fb = torch.randn(B, HOP * (T - 1), device=dev) * 8000
        plan_fb.mel_from_waveform(wave_fb)
        efw = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        efw[0].record()
        for _ in range(5):
            plan_fb.mel_from_waveform(wave_fb)
        efw[1].record()
        torch.cuda.synchronize(dev)
        other_cfg["full_band_20hz_20khz"] = {
            "tiles_per_s": round(B / dtfb, 1), "ms_per_step": round(dtfb * 1e3, 3), "steps": 3, "inverse_mel_ms": round(efb[0].elapsed_time(efb[1]), 3),
            "forward_mel_ms": round(efw[0].elapsed_time(efw[1]) / 5, 3),
            "inverse_mel_kernel": int(plan_fb.lib.rfx_plan_imel_kernel(plan_fb.handle)), "finite": bool(torch.isfinite(pcm_fb.float()).all()),
            "workload": f"batch={B} synthetic 512x512 mono uint8 tiles -> audio with min_frequency=20, max_frequency=20000 (the reference's "
                        f"round-trip test parameters), Griffin-Lim {args.iters}; inverse_mel_kernel 5 = line-form group kernel, 0 = general LDS kernel"}
        del plan_fb, mel_fb, pcm_fb, wave_fb
        for Bs in (65, 96, 100):  # distance from the straight line between B = 64 and B = 128 (the run partition has no cliff: <= 5 %)
            line = sweep["64"]["ms_per_step"] + (sweep["128"]["ms_per_step"] - sweep["64"]["ms_per_step"]) * (Bs - 64) / 64.0
            sweep[str(Bs)]["vs_linear_64_128_pct"] = round(100.0 * (sweep[str(Bs)]["ms_per_step"] / line - 1.0), 2)
        other_cfg["batch_sweep"] = {"workload": f"the headline decode step (Griffin-Lim {args.iters}) at other batch sizes, 3 steps each", "by_batch": sweep}

    # ---- configs[2] (audio -> mel image) measured in the same run and carried on the same line
    fwd = None
    if not args.no_forward:
        fwd = forward_measure(args, world, rank, dev, distributed, with_cpu=False)
    if rank == 0:
        out["single_tile_latency"] = latency
        if other is not None:
            out["other_sample_rates"] = other
        if other_cfg is not None:
            out["other_configs"] = other_cfg
        if fwd is not None:
            out["forward"] = {k: fwd[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "stages")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.iters)
    finish(out if rank == 0 else None)


if __name__ == "__main__":
    main()
