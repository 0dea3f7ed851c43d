#!/bin/bash
# Round 6: forward path after the dispatch-order skew and the encoder's estimate: tests that hold its bits, the bench line, kernel stats.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6f; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_round5.py tests/test_gpu_full_size.py tests/test_gpu_api_contract.py -m gpu -q -x > $OUT/pytest_fwd.log 2>&1; echo "fwd tests rc=$?"; tail -2 $OUT/pytest_fwd.log
for i in 1 2; do timeout 300 python bench.py --workload forward --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $OUT/bench_forward_$i.json; python - <<PY
import json
d = json.load(open("$OUT/bench_forward_$i.json")); print(d["value"], d["ms_per_step"], {k: v for k, v in d["stages"].items() if k != "note"})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fwd -- python $R/bench.py --workload forward --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/*kernel_stats.csv $OUT/forward_kernel_stats.csv 2>/dev/null; head -8 $OUT/forward_kernel_stats.csv | cut -c1-200
