#!/bin/bash
# bash tools/ab_coresidency.sh <preset> <lib or ->...   one resident batch against two batches alternating on two
# engines (streams), per build of the library: ms per step, main_dp stage, the alternating figure
preset=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset TBA_LIB_PATH; else export TBA_LIB_PATH=$PWD/$lib; fi
  timeout 400 python bench.py --preset $preset --steps 8 --warmup 2 --interleaved --e2e none --api-reads 0 --no-pmc --no-cpu-baseline 2>gpurun_out/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']; t = c.get('two_resident_batches_alternating') or {}
s = c.get('stage_ms', {})
print('$preset', '$lib', '| one batch: %.2f ms/step = %.0f reads/s, main_dp %.2f, other stages %.2f | two batches alternating: %s reads/s (%s ms per batch)' % (
    d['ms_per_step'], d['value'], s.get('main_dp', 0), s.get('total', 0) - s.get('main_dp', 0), t.get('reads_per_s'), t.get('ms_per_step')),
    '| digest_stable', c.get('digest_stable'), 'verify_fail', c.get('tb_verify_fail_rows'))
" || tail -5 gpurun_out/ab_err.log
done
