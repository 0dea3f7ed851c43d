#!/bin/bash
# Round 6: profiles only (kernel stats + PMC passes of the two dominant kernels on HEAD's sources) and the forward bench line.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round6c; mkdir -p $OUT; cd $R
timeout 300 python bench.py --workload forward --steps 20 --warmup 5 > $OUT/bench_forward.json 2>/dev/null; cut -c1-200 $OUT/bench_forward.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; tail -14 $OUT/profile_round.log | cut -c1-200
