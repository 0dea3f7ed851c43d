#!/bin/bash
# Round 3, one GPU visit: the GPU test-suite, smoke, the bench lines (headline, configs[3] on one GPU with its stage split, the
# same through a 1-rank RCCL launch with gather=all), and the InverseMelScale PMC summary restricted to 64-tile launches.
# usage: gpu_round3.sh [tests|notests]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round3; mkdir -p $OUT; cd $R
if [ "${1:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  grep -E "dB|rel-L2|convergence|re-projection|SGD kernel" $OUT/pytest_gpu.log > $OUT/parity_figures.txt; tail -40 $OUT/parity_figures.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
timeout 600 python bench.py --steps 10 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cut -c1-1500 $OUT/bench.json
timeout 900 python bench.py --workload decode-stereo64 --global-clips 512 --steps 2 --warmup 1 > $OUT/bench_stereo64_512clips_1gpu.json 2> $OUT/bench_stereo64.err; tail -3 $OUT/bench_stereo64.err; cat $OUT/bench_stereo64_512clips_1gpu.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload decode-stereo64 --global-clips 128 --gather all --steps 1 --warmup 1 > $OUT/bench_stereo64_rccl1_gather_all.json 2> $OUT/bench_rccl1.err; tail -2 $OUT/bench_rccl1.err; cut -c1-600 $OUT/bench_stereo64_rccl1_gather_all.json
bash tools/pmc_imel.sh 2>&1 | tail -3
