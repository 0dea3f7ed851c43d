#!/bin/bash
# One GPU visit: the GPU test-suite, smoke, the bench lines (headline, 1-rank RCCL launch, configs[3] sample).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round2; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
grep -E "dB|rel-L2|convergence|re-projection" $OUT/pytest_gpu.log | head -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
unset RFX_LIB_PATH
timeout 600 python bench.py --steps 10 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cat $OUT/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-forward > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err; cut -c1-300 $OUT/bench_rccl1.json
timeout 600 python bench.py --workload decode-stereo64 --global-clips 128 --steps 1 --warmup 1 > $OUT/bench_stereo64.json 2> $OUT/bench_stereo64.err; cut -c1-300 $OUT/bench_stereo64.json
