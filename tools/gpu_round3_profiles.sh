#!/bin/bash
# Round 3, the profile visit: bench lines, rocprofv3 kernel stats + PMC summaries (tools/profile_round.sh, tools/pmc_imel.sh),
# generic-engine probe.  Everything lands under gpurun_out/; the summaries to be judged are copied into profiles/ afterwards.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round3; mkdir -p $OUT; cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-400 $OUT/bench.json
timeout 900 python bench.py --workload decode-stereo64 --global-clips 512 --steps 2 --warmup 1 > $OUT/bench_stereo64_512clips_1gpu.json 2> $OUT/bench_stereo64.err; cut -c1-300 $OUT/bench_stereo64_512clips_1gpu.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload decode-stereo64 --global-clips 128 --gather all --steps 1 --warmup 1 > $OUT/bench_stereo64_rccl1_gather_all.json 2> $OUT/bench_rccl1.err; cut -c1-200 $OUT/bench_stereo64_rccl1_gather_all.json
timeout 600 python bench.py --workload forward --steps 20 --warmup 5 > $OUT/bench_forward.json 2>/dev/null; cut -c1-300 $OUT/bench_forward.json
bash tools/profile_round.sh 2>&1 | tail -14
bash tools/pmc_imel.sh 2>&1 | tail -2
python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | tee $OUT/generic_engine_probe.txt
python tools/probe_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/latency_small_batches.txt | tail -8
