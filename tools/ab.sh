#!/bin/bash
# A/B visit: for every variant library build_var/librfx_NAME.so run the Griffin-Lim probe, the forward line of bench.py and
# (RATES set) the row-family probe.  tools/ab.sh NAME...   -> gpurun_out/ab.log
mkdir -p gpurun_out
for v in "$@"; do
  export RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so
  echo "=== $v"
  python tools/probe_gl.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-2}
  [ -n "$FWD" ] && python bench.py --workload forward --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forward', d['value'], d['unit'], d.get('stages'))"
  [ -n "$RATES" ] && TAG=$v python tools/probe_fam.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/ab.log
