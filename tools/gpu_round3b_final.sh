#!/bin/bash
# Round 3, second session: the whole GPU suite, smoke, bench lines, rocprofv3 kernel stats + PMC summaries, probes.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round3b; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
grep -E "dB|rel-L2|convergence|re-projection|SGD kernel|rel err" $OUT/pytest_gpu.log > $OUT/parity_figures.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-300 $OUT/bench.json
timeout 600 python bench.py --workload forward --steps 20 --warmup 5 > $OUT/bench_forward.json 2>/dev/null; cut -c1-200 $OUT/bench_forward.json
timeout 600 python bench.py --workload decode-stereo64 --global-clips 512 --steps 2 --warmup 1 > $OUT/bench_stereo64_512clips_1gpu.json 2> $OUT/bench_stereo64.err; cut -c1-200 $OUT/bench_stereo64_512clips_1gpu.json
bash tools/profile_round.sh 2>&1 | tail -12 | cut -c1-400
bash tools/pmc_imel.sh 2>&1 | tail -1 | cut -c1-600
bash tools/pmc_fam.sh 2>&1 | tail -1 | cut -c1-900
RATES=48000,32000,24000,22050,16000,8000,44100 timeout 300 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | tee $OUT/engine_probe.txt
timeout 300 python tools/probe_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/latency_small_batches.txt | tail -4
