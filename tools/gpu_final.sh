#!/bin/bash
# The short end-of-round visit: smoke, the whole GPU suite, the bench line, rocprofv3 kernel stats of the bench command and the
# InverseMelScale counters.  (tools/gpu_round.sh is the long one with every probe and counter pass.)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed on this box: giving up"; tail -3 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout 900 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E " passed| failed" $OUT/pytest_gpu.log | tail -2
grep -E "dB|rel-L2|convergence|re-projection|plan cache" $OUT/pytest_gpu.log > $OUT/gpu_parity_figures.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["stages"], d["roofline"]["achieved"], d["forward"]["value"])
print({k: (v.get("tiles_per_s"), v.get("forward_images_per_s"), v.get("griffinlim_ms")) for k, v in d["other_sample_rates"].items()})
print(d["roofline"].get("binding", {}).get("frac_of_measured_instruction_rate"), d["cpu_baseline"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/stats.log 2>&1
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; head -6 $OUT/kernel_stats.csv | cut -c1-160
cd $R; [ -n "$NO_PMC" ] || { bash tools/pmc_imel.sh > $OUT/pmc_imel.log 2>&1; tail -1 $OUT/pmc_imel.log | cut -c1-400; }
