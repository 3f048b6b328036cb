#!/bin/bash
# A/B of library builds on the resident leg only: bash tools/ab_main_dp.sh <preset> <lib or ->...
# prints ms per step and the main_dp stage of each build ("-" = the tree's libtombo_amd.so)
preset=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset TBA_LIB_PATH; else export TBA_LIB_PATH=$lib; fi
  timeout 240 python bench.py --preset $preset --steps 6 --warmup 1 --e2e none --api-reads 0 --no-pmc --no-cpu-baseline 2>gpurun_out/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$preset', '$lib', 'ms_per_step %.2f' % d['ms_per_step'], 'value %.0f' % d['value'], 'stage_ms', {k: round(v, 2) for k, v in d['config'].get('stage_ms', {}).items() if v > 0.3})
" || tail -5 gpurun_out/ab_err.log
done
