#!/bin/bash
# round 5: the whole GPU suite, the bench line, kernel stats and counters on HEAD (OUT = gpurun_out/r5g)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5g; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
{ ls /sys/class/drm/; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done; rocm-smi --showclocks 2>&1 | head -30; } > $OUT/clock_probe.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E " passed| failed" $OUT/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
grep -E "dB|rel-L2|convergence|re-projection|plan cache" $OUT/pytest_gpu.log > $OUT/gpu_parity_figures.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_median"], d["ms_per_step_max"], d["shader_clock"])
print(d["stages"], d["roofline"]["summary"], d["roofline"]["from_profiles"])
print(d["other_configs"]["stereo64"]["value"], d["other_configs"]["stereo64"]["ms_per_step"], d["other_configs"]["stereo64"]["roofline"]["frac"])
print(d["other_configs"]["batch_sweep"]["by_batch"])
print(d["forward"]["value"], {k: (v.get("tiles_per_s"), v.get("forward_images_per_s")) for k, v in d["other_sample_rates"].items()}, d["cpu_baseline"]["value"])
PY
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; tail -8 $OUT/profile_round.log | cut -c1-300
