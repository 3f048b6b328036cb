# Round 4 A/B helper (GPU box): bash tools/r04_ab.sh <out-name> <preset>[:extra bench args] ... -- <lib or -> ...
# runs the -m gpu tests once on the tree's library, then bench.py per preset and library
# ("-" = tombo_amd/libtombo_amd.so), resident + end-to-end legs only; one line per run in gpurun_out/<out-name>.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/$1.txt; shift
presets=(); while [ "$1" != "--" ]; do presets+=("$1"); shift; done; shift
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/$(basename $out .txt)_pytest.txt
tail -3 gpurun_out/$(basename $out .txt)_pytest.txt
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset TBA_LIB_PATH; else export TBA_LIB_PATH=$R/$lib; fi
  for pa in "${presets[@]}"; do
    cfg=${pa%%:*}; extra=""; [ "$pa" != "$cfg" ] && extra=${pa#*:}
    timeout 600 python bench.py --preset $cfg --steps 8 --warmup 1 --no-pmc --no-cpu-baseline --api-reads 0 $extra 2>gpurun_out/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$extra', '$lib', 'ms_per_step %.2f' % d['ms_per_step'], 'resident %.0f' % d['value'], 'e2e', (d.get('end_to_end') or {}).get('value'), 'stage_ms', {k: round(v, 2) for k, v in d['config'].get('stage_ms', {}).items() if v > 0.3})
" >> $out 2>&1 || tail -5 gpurun_out/ab_err.log >> $out
  done
done
unset TBA_LIB_PATH
cat $out
