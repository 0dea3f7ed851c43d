#!/bin/bash
# Clock and power of the chip while one kernel family runs in a loop (is the headline kernel power-capped?): rocm-smi samples
# next to tools/probe_gl.py (Griffin-Lim) and tools/probe_imel.py (InverseMelScale) -> gpurun_out/power.log
mkdir -p gpurun_out
{
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | head -6
for w in gl imel; do
  if [ $w = gl ]; then REPS=40 python tools/probe_gl.py > /dev/null 2>&1 & else (for i in 1 2 3 4 5 6; do python tools/probe_imel.py; done) > /dev/null 2>&1 & fi
  pid=$!
  sleep 12
  for i in 1 2 3 4 5; do echo "== under $w load, sample $i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3; sleep 1; done
  wait $pid
done
} 2>&1 | tee gpurun_out/power.log
