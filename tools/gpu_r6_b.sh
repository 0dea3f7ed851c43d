#!/bin/bash
# Round 6, second kind of visit: the codec and 48 kHz tests (encoder estimate, masked 32-iteration gate), then the forward bench.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_round5.py tests/test_gpu_round3_parity.py -m gpu -q -x > $OUT/pytest_codec.log 2>&1; echo "codec rc=$?"; tail -3 $OUT/pytest_codec.log
timeout 1200 python -m pytest tests/test_gpu_generic_geometry.py -m gpu -q -s -k griffinlim_matches_oracle > $OUT/pytest_gen.log 2>&1; echo "gen rc=$?"; grep -E "n_iter=32|passed|failed|^FAILED" $OUT/pytest_gen.log | cut -c1-400
timeout 300 python bench.py --workload forward --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_forward.json 2>/dev/null; python - <<PY
import json
d = json.load(open("$OUT/bench_forward.json")); print(d["value"], d["ms_per_step"], {k: v for k, v in d["stages"].items() if k != "note"})
PY
