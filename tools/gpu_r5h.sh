#!/bin/bash
# round 5: clock and socket power under each kernel of the path
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5h; mkdir -p $OUT; cd $R
timeout 200 python tools/probe_power.py 2>&1 | grep -v amdgpu.ids > $OUT/power.txt; cat $OUT/power.txt | cut -c1-300
rocm-smi --showpower --showmaxpower 2>&1 | grep -i "power" | head -5
