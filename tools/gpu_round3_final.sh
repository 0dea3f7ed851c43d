cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/round3_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/round3_pytest_gpu.log | tail -2
mkdir -p gpurun_out/round3; grep -E "dB|rel-L2|convergence|re-projection|SGD kernel|max\|d\|" gpurun_out/round3_pytest_gpu.log > gpurun_out/round3/parity_figures.txt
bash tools/gpu_round3_profiles.sh 2>&1 | cut -c1-300 | tail -40
