#!/bin/bash
# Round 6, evidence visit: smoke, the whole GPU suite, the bench lines (headline, 1-rank RCCL launch, forward), then the profiles
# (tools/profile_round.sh: rocprofv3 kernel stats + PMC passes of the two dominant kernels; tools/pmc_imel.sh).
# Everything lands in gpurun_out/round6/ (+ gpurun_out/prof, pmc_imel); what is to be judged is copied into profiles/ afterwards.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round6; mkdir -p $OUT; cd $R
unset RFX_LIB_PATH
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed on this box: giving up"; tail -3 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q -s -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -E "^SKIPPED" $OUT/pytest_gpu.log | cut -c1-200
grep -E "dB|rel-L2|convergence|re-projection|plan cache|max_value|differ" $OUT/pytest_gpu.log > $OUT/gpu_parity_figures.txt; wc -l $OUT/gpu_parity_figures.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-330 $OUT/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-forward --no-other-rates --no-other-configs --n1-ms 29.5 > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err; cut -c1-200 $OUT/bench_rccl1.json
timeout 300 python bench.py --workload forward --steps 20 --warmup 5 > $OUT/bench_forward.json 2>/dev/null; cut -c1-200 $OUT/bench_forward.json
timeout 600 python bench.py --workload decode-stereo64 --global-clips 512 --steps 1 --warmup 1 --host-input > $OUT/bench_stereo64_host_input.json 2> $OUT/bench_stereo64.err; cut -c1-300 $OUT/bench_stereo64_host_input.json
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; tail -14 $OUT/profile_round.log | cut -c1-260
bash tools/pmc_imel.sh > $OUT/pmc_imel.log 2>&1; tail -1 $OUT/pmc_imel.log | cut -c1-300
