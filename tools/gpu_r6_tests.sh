#!/bin/bash
# Round 6: the new tests first (chunking / sharding invariance, arena, numeric range), then the whole GPU suite without -x, then the bench line.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6t; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round6_range.py -m gpu -q -s > $OUT/pytest_r6.log 2>&1; echo "r6 rc=$?"; grep -E "max_value|Hz \(|^FAILED|^ERROR|passed|failed|Error" $OUT/pytest_r6.log | cut -c1-300 | tail -60
if [ "$1" != "quick" ]; then
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round6.py --deselect tests/test_gpu_round6_range.py > $OUT/pytest_gpu.log 2>&1; echo "all rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -30 | cut -c1-300
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_median", "ms_per_step_max", "ms_per_step_max_over_min")})
print({k: v for k, v in d["stages"].items() if k != "note"}); print({k: v for k, v in d["other_configs"]["og_beat_cli"].items() if k != "workload"}); print(d["other_configs"]["batch_sweep"]["by_batch"]); print({k: v for k, v in d["cpu_baseline"].items() if k != "sample"}); print(d["single_tile_latency"])
print(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["from_profiles"]["stale"]); print(d["forward"]["value"], {k: v for k, v in d["forward"]["stages"].items() if k != "note"})
print(d["other_configs"]["stereo64"]["value"], d["shader_clock"]["mhz_median"], {k: (v["tiles_per_s"], v["forward_images_per_s"]) for k, v in d["other_sample_rates"].items()})
PY
