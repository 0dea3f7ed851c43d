"""
bench.py's output contract, exercised on the GPU with a small batch: exactly ONE line on stdout, valid JSON with the keys the
driver reads, the `roofline` / `forward` objects, and the same through a torch.distributed.run launch (RCCL, one rank) - whose
communicator start-up banner must not reach stdout.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def _one_json_line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {[ln[:60] for ln in lines]}"
    return json.loads(lines[0])


def test_headline_line_small_batch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "4", "--iters", "32",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    d = _one_json_line(p)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["unit"] == "tiles/s" and d["higher_is_better"] is True
    assert d["metric"] == "spectrogram_tiles_per_sec_griffinlim32" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    b = r["binding"]  # round 4: occupancy of the fp32 pipes (PMC instruction count x pipe cycles per instruction of the ISA mix)
    assert b["bound"] == "valu_pipe" and 0 < b["frac"] < 1.0 and b["frac"] <= b.get("frac_at_measured_clock", 1.0) and "source" in r["from_profiles"]
    assert b["loop_mix"]["valu_packed"] > b["loop_mix"]["valu_plain"] > 0 and 2.0 <= b["pipe_cycles_per_valu_instruction"] <= 4.0
    f = d["forward"]
    assert f["unit"] == "images/s" and f["value"] > 0 and f["roofline"]["kernel"] == "rfx::stft_mel2_kernel" and 0 < f["roofline"]["true_flops_frac"] < 1
    assert {"image_decode_ms", "inverse_mel_ms", "griffinlim_ms", "pcm16_ms"} <= set(d["stages"])
    assert 0 < d["single_tile_latency"]["mono_ms"] < d["single_tile_latency"]["stereo_ms"] * 1.5
    o = d["other_sample_rates"]["48000"]
    assert o["griffinlim_engine"] == "row-family" and o["n_fft"] == 19200 and o["tiles_per_s"] > 0 and o["finite"] is True
    o2 = d["other_sample_rates"]["22050"]
    assert o2["griffinlim_engine"] == "row-family" and o2["n_fft"] == 8820 and o2["tiles_per_s"] > 0 and o2["forward_images_per_s"] > 0
    # round 5: the spread of the timed steps and the clock, the three roofline readings by name, provenance of the counters,
    # configs[3]'s one-GPU share and the batch-shape sweep on the same line
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"] and "source" in d["shader_clock"]
    assert set(r["summary"]) == {"frac_canonical_bytes", "actual_hbm_frac", "binding_frac_valu_pipe", "binding_frac_of_measured_instruction_rate"}
    assert isinstance(r["from_profiles"]["stale"], bool) and len(r["from_profiles"]["kernel_sources_here"]) == 12
    s64 = d["other_configs"]["stereo64"]
    assert s64["unit"] == "tiles/s" and s64["value"] > 0 and s64["finite"] is True and 0 < s64["roofline"]["frac"] < 1.5
    sweep = d["other_configs"]["batch_sweep"]["by_batch"]
    assert set(sweep) == {"16", "48", "64", "65", "96", "100", "128"}
    # no run-partition cliff (rounds 1-4: +26 % at B = 65).  Round 6: runs are whole groups of 16 frames (what makes a clip's bits
    # independent of its batch), so a batch that is not a multiple of 16 tiles rounds the longest run up to the next group:
    # measured +7.0 % at B = 65 and +6.3 % at B = 100 (the few long runs finish alone on their CUs, faster), 0 at multiples of 16
    for b_ in ("65", "96", "100"):
        assert abs(sweep[b_]["vs_linear_64_128_pct"]) <= (3.0 if b_ == "96" else 14.0), sweep
    # round 6: the timed step is the product call; the four-call form of rounds 1-5 rides along; configs[0]'s GPU half
    st = d["stages"]
    assert st["product_call_ms"] > 0 and st["c_abi_four_calls_ms"] > 0 and st["workspace_arena"]["buffers_allocated_in_this_process"] >= 1
    og = d["other_configs"]["og_beat_cli"]
    assert og["ms"] > 0 and og["wav_bytes"] > 400000 and 5.0 < og["audio_seconds"] < 5.2


def test_distributed_launch_one_rank_keeps_stdout_clean():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-forward", "--n1-ms", "10.0"]
    d = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=600))
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4
    # round 6: what an N > 1 line carries - every rank's own time (all_gather over RCCL) and, with --n1-ms, the weak-scaling figure
    assert d["per_rank_ms"]["by_rank"] == [d["per_rank_ms"]["min"]] == [d["per_rank_ms"]["max"]] and d["vs_n1"]["n1_ms"] == 10.0
    assert abs(d["vs_n1"]["speedup"] - 10.0 / d["ms_per_step"]) < 0.01 * d["vs_n1"]["speedup"] + 0.01


def test_sharded_stereo_workload_small():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "decode-stereo64", "--global-clips", "4", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600)
    d = _one_json_line(p)
    assert d["scaling"] == "strong" and d["config"]["griffin_lim_iters"] == 64 and d["config"]["global_batch"] == 4 and d["value"] > 0
    assert d["config"]["gather"] == "none" and {"compute_ms", "compute_plus_d2h_ms", "d2h_exposed_ms", "d2h_own_shard_raw_ms",
                                                 "host_in_host_out_ms", "h2d_exposed_ms"} <= set(d["stages"])
    # the same workload with the tiles in host memory when the timed region starts (the reference's API: host in, host out)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "decode-stereo64", "--global-clips", "4", "--steps", "1",
                        "--warmup", "0", "--host-input"], capture_output=True, text=True, timeout=600)
    d = _one_json_line(p)
    assert "HOST memory" in d["config"]["workload"] and d["value"] > 0
