#!/bin/bash
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stft_gl.py -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_first.log
cat gpurun_out/pytest_first.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gl -o gl -- python $GRAFT_REPO_ROOT/tools/probe_gl.py > $GRAFT_REPO_ROOT/gpurun_out/prof_gl.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_gl | head -30
