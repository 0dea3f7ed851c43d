#!/bin/bash
# One GPU visit (round 4): the GPU test-suite, smoke, the bench lines (headline, 1-rank RCCL launch, configs[3] with host input),
# the engine probe over the sample rates, then the profiles (tools/profile_round.sh, tools/pmc_imel.sh, tools/pmc_fam.sh).
# Everything lands in gpurun_out/round4/ (+ gpurun_out/prof, pmc_imel, pmc_fam); copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round4; mkdir -p $OUT; cd $R
unset RFX_LIB_PATH
# a box that faults on the smoke test faults on everything after it too (one visit of round 4 burned 19 GPU-minutes that way): stop at once
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed on this box: giving up"; tail -3 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
grep -E "dB|rel-L2|convergence|re-projection|plan cache" $OUT/pytest_gpu.log > $OUT/gpu_parity_figures.txt; wc -l $OUT/gpu_parity_figures.txt
grep -q "rc=0" $OUT/pytest_gpu.log || { echo "GPU tests failed: no bench, no profiles"; tail -5 $OUT/pytest_gpu.log; exit 1; }
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-400 $OUT/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-forward --no-other-rates > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err; cut -c1-200 $OUT/bench_rccl1.json
timeout 600 python bench.py --workload decode-stereo64 --global-clips 512 --steps 1 --warmup 1 --host-input > $OUT/bench_stereo64_host_input.json 2> $OUT/bench_stereo64.err; cut -c1-300 $OUT/bench_stereo64_host_input.json
RATES=${RATES:-48000,32000,24000,22050,16000,11025,8000,44100} timeout 600 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids > $OUT/engine_probe.txt; cat $OUT/engine_probe.txt
RATES=48000,32000,24000,22050,16000,8000 timeout 300 python tools/probe_fwd_rate.py 2>&1 | tail -1 > $OUT/forward_rates.txt; cat $OUT/forward_rates.txt
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; tail -14 $OUT/profile_round.log | cut -c1-260
bash tools/pmc_imel.sh > $OUT/pmc_imel.log 2>&1; tail -1 $OUT/pmc_imel.log | cut -c1-300
bash tools/pmc_fam.sh > $OUT/pmc_fam.log 2>&1; tail -1 $OUT/pmc_fam.log | cut -c1-600
