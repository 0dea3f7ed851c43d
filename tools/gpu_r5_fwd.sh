#!/bin/bash
# round 5: the forward path after the table-traffic work: its GPU tests, the configs[2] bench line, kernel stats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5fwd; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_mel_codec.py tests/test_gpu_boundary_round2.py tests/test_gpu_api_contract.py tests/test_gpu_round4.py -m gpu -q -x -s -k "image or forward or mel or codec or audio or spectrogram or cli" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED|^ERROR|Error" $OUT/pytest.log | head; grep "forward kernel" $OUT/pytest.log
timeout 300 python bench.py --workload forward --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_forward.json 2> $OUT/bench_forward.err; tail -2 $OUT/bench_forward.err
python -c "
import json; d=json.load(open('$OUT/bench_forward.json')); print(d['value'], d['unit'], d['ms_per_step'], d['stages'])"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fwd -- python $R/bench.py --workload forward --no-cpu-baseline --steps 10 --warmup 2 > $OUT/stats.log 2>&1
cp $OUT/stats/*kernel_stats.csv $OUT/forward_kernel_stats.csv 2>/dev/null; head -8 $OUT/forward_kernel_stats.csv | cut -c1-160
