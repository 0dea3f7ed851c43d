#!/bin/bash
# Round 6, first visit: does the tree as inherited still pass on today's box, and what does the headline read there?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6base; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed"; tail -3 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-600 $OUT/bench.json
