#!/usr/bin/env python3
"""
bench.py - headline benchmark of the spectrogram -> audio hot path on MI355X.

Workload (BASELINE.json configs[1]): B = 64 synthetic 512x512 mono spectrogram tiles (uint8, already
resident in HBM) -> image decode -> InverseMelScale (SGD 200) -> Griffin-Lim 32 -> int16 PCM, all in
HIP kernels through librfx.so.  One "step" = one such batch.  value = tiles/s over all ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --workload forward        # secondary line: configs[2], audio -> mel images (MFMA roofline)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Clips are independent, so N GPUs each process their own 64 tiles (weak scaling, no data-path
collective; the only collectives are the timing barrier and the max-over-ranks reduction).

Extra objects on the JSON line:
  roofline     - the dominant kernel (rfx::gl_iter_kernel<2>, one Griffin-Lim iteration over the
                 batch): algorithmic bytes per launch = 20 B x 8821 bins x 512 frames x 64 tiles
                 (|S| 4 B + tprev 8 B read + 8 B written per bin and iteration)
                 divided by the launch duration measured with HIP events on the launch stream.
  cpu_baseline - the CPU oracle (oracle/riffusion_oracle.py, a torch-CPU port of the reference's
                 torchaudio path) timed on this host on ONE tile of the same workload.
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
    ap.add_argument("--workload", choices=["decode", "forward"], default="decode",
                    help="decode = the headline (tiles -> audio); forward = BASELINE.json configs[2] (audio -> mel images)")
    return ap.parse_args()


def pmc_summary():
    """Committed rocprofv3 PMC summary of the dominant kernel (HBM bytes and VALU instructions per launch)."""
    path = os.path.join(ROOT, "profiles", "gl_iter_pmc_latest.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def cpu_baseline(iters: int, threads_cap: int = 16, min_seconds: float = 10.0, max_tiles: int = 8):
    """
    The oracle (a torch-CPU port of the reference's torchaudio path) on the host cores, on a BOUNDED
    sample of the same workload: whole synthetic mono 512x512 tiles, one after the other (one call of
    the reference per tile), until at least `min_seconds` of CPU work have been timed (at most
    `max_tiles` tiles).  Threads are capped: torch's CPU kernels degrade badly when a 256-thread host
    is oversubscribed (the unbounded run took 424 s for one tile).
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
    return {
        "value": round(n / total, 5),
        "unit": "tiles/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n} synthetic mono 512x512 tile(s), one reference call each: InverseMelScale SGD-200 {t_imel:.1f} s + "
        f"Griffin-Lim {iters} {t_gl:.1f} s (torch {torch.__version__} CPU, {threads} threads of {os.cpu_count()} logical cores)",
        "griffinlim_only_tiles_per_s": round(n / t_gl, 5),
    }


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


def forward_main(args, world, rank, dev, distributed):
    """BASELINE.json configs[2]: B waveforms -> STFT -> MFMA mel GEMM -> uint8 image, per rank."""
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

    def step():
        mel = plan.mel_from_waveform(wave)
        return plan.image_encode(mel, False, thr)[0]

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

        stft_ms, _ = timed(lambda: plan.stft(wave, want_mag=True, want_spec=False))
        mel_ms, mel = timed(lambda: plan.mel_from_waveform(wave))
        enc_ms, _ = timed(lambda: plan.image_encode(mel, False, thr))
        gemm_ms = mel_ms - stft_ms  # mel_from_waveform = stft_kernel + mel_gemm_kernel on one stream
        k_exec = executed_mel_k()
        tflops = 2.0 * N_MELS * k_exec * B * N_FRAMES / (gemm_ms * 1e-3) / 1e12
        images_per_s = world * B * args.steps / elapsed
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
            "config": {"workload": f"batch={B} synthetic waveforms of {L} samples -> STFT -> MFMA mel GEMM -> uint8 image "
                                   "(BASELINE.json configs[2]); waveforms resident in HBM",
                       "batch_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"clips sharded over {world} GPU(s), no data-path collective"},
            "audio_sec_per_sec": round(images_per_s * L / SR, 1),
            "roofline": {"kernel": "rfx::mel_gemm_kernel", "bound": "mfma", "achieved": round(tflops, 1),
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": None, "executed_k": k_exec, "avg_launch_ms": round(gemm_ms, 4),
                         "dense_equivalent_tflops": round(2.0 * N_MELS * N_BINS * B * N_FRAMES / (gemm_ms * 1e-3) / 1e12, 1),
                         "note": "flops = 2 x 512 mel x executed K x frames (all-zero filterbank blocks are skipped); launch time = "
                                 "mel_from_waveform minus stft on the same stream (torch events on the launch stream)"},
            "stages": {"stft_ms": round(stft_ms, 3), "mel_gemm_ms": round(gemm_ms, 3), "image_encode_ms": round(enc_ms, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = forward_cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ  # launched by torch.distributed.run (also exercised with one rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if distributed else 0)
    torch.cuda.set_device(dev)
    if args.workload == "forward":
        return forward_main(args, world, rank, dev, distributed)

    from riffusion import _hip
    from riffusion.spectrogram_params import SpectrogramParams
    from riffusion.util import image_util

    params = SpectrogramParams(num_griffin_lim_iters=args.iters)
    plan = _hip.get_plan(params, dev)
    B, T = args.batch, N_FRAMES

    # synthetic tiles of SURVEY.md 8(d), a different seed per rank, resident in HBM before timing
    rng = np.random.default_rng(20240807 + rank)
    tiles = torch.from_numpy(rng.integers(0, 256, size=(B, N_MELS, T, 3), dtype=np.uint8)).to(dev)
    lut = torch.from_numpy(image_util.decode_lut(0.25, 30e6)).to(dev)
    gl_ws = torch.empty(plan.lib.rfx_griffinlim_workspace_bytes(plan.handle, B, T), dtype=torch.uint8, device=dev)

    def step(seed, launch_ms=None):
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
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pcm = step(100 + k)
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(pcm.float()).all())

    # ---- roofline leg (outside the timed region): per-launch durations from HIP events on the stream
    roofline = None
    extra = {}
    if rank == 0:
        ms = (ctypes.c_float * (args.iters + 1))()
        step(999, launch_ms=ms)
        steady = [ms[i] for i in range(2, args.iters + 1)] or [ms[-1]]
        avg_ms = sum(steady) / len(steady)
        alg_bytes = 20.0 * N_BINS * T * B
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        pmc = pmc_summary()
        traffic = pmc.get("hbm_bytes_per_launch")
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
            # The figure above prices the kernel against the CANONICAL fused formulation of SURVEY 8(d)
            # (|S| 4 B + tprev 8 B read + 8 B written per bin and iteration).  The shipped kernel applies the
            # momentum in the time domain (STFT linearity) and streams only |S|: `traffic` (PMC) is what it
            # really moves, and its binding resource is fp32 VALU issue, reported next.
            "formulation": "time-domain momentum: rebuilt - m*tprev = STFT(x_k - m*x_{k-1}); 4 B/bin/iteration streamed",
        }
        if traffic:
            roofline["actual_hbm_gbs"] = round(traffic / (avg_ms * 1e-3) / 1e9, 1)
        valu = pmc.get("SQ_INSTS_VALU_per_launch")
        if valu:
            # issue ceiling measured by tools/ubench/valu.hip: 1.13 ns per plain fp32 wave-instruction per SIMD, 1024 SIMDs
            peak_ginstr = 1024 / 1.13
            got = valu / (avg_ms * 1e-3) / 1e9
            roofline["valu_issue"] = {"wave_instructions_per_launch": valu, "achieved_ginstr_s": round(got, 1),
                                      "peak_ginstr_s": round(peak_ginstr, 1), "frac": round(got / peak_ginstr, 4)}
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

    ms_per_step = elapsed / args.steps * 1e3
    tiles_per_s = world * B * args.steps / elapsed
    if rank == 0:
        out = {
            "metric": "spectrogram_tiles_per_sec_griffinlim32" if args.iters == 32 else f"spectrogram_tiles_per_sec_griffinlim{args.iters}",
            "value": round(tiles_per_s, 2),
            "unit": "tiles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"batch={B} synthetic 512x512 mono uint8 tiles -> image decode -> InverseMelScale SGD-200 -> "
                f"Griffin-Lim {args.iters} -> int16 PCM (BASELINE.json configs[1]); tiles resident in HBM",
                "batch_per_gpu": B,
                "global_batch": world * B,
                "griffin_lim_iters": args.iters,
                "parallelism": f"clips sharded over {world} GPU(s), no data-path collective",
            },
            "audio_sec_per_sec": round(tiles_per_s * HOP * (T - 1) / SR, 1),
            "roofline": roofline,
            "stages": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.iters)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
