cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_boundary_round2.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3b_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages'], d['roofline']['avg_launch_ms'], d['forward']['value'], d.get('other_sample_rates'))"
