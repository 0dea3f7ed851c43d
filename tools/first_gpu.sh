#!/bin/bash
# first GPU contact: parity tests, then a timing probe
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stft_gl.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_first.log
cat gpurun_out/pytest_first.log
python tools/probe_gl.py 2>&1 | tee gpurun_out/probe_gl.log
