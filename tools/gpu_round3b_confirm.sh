#!/bin/bash
# Round 3, second session, last visit: the whole GPU suite, smoke, the bench lines and the engine probe on the final binaries
# (the PMC summaries of the unchanged kernels are the ones of tools/gpu_round3b_final.sh).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round3c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
grep -E "dB|rel-L2|convergence|re-projection|SGD kernel|rel err" $OUT/pytest_gpu.log > $OUT/parity_figures.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-260 $OUT/bench.json
RATES=48000,32000,24000,22050,16000,8000,44100 timeout 200 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | tee $OUT/engine_probe.txt
