# Round 4, first GPU call (VERDICT r3 "next" #1): the evidence of the final k_dp tree, and the
# register-budget A/B of k_dp<8>.  gpurun -- 'bash tools/r04_measure_first.sh'; results in gpurun_out/r04a/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
export TMPDIR=/tmp
Q="--no-pmc --no-cpu-baseline --api-reads 0"
# kernel table of the cfg2 bench command (rocprofv3 --kernel-trace --stats)
cd /tmp
rm -rf /tmp/st2; rocprofv3 --kernel-trace --stats -d /tmp/st2 -- python $R/bench.py --steps 3 --warmup 1 $Q --e2e none > $O/bench_line_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/st2 -name "*.db" | head -1) > $O/kernel_stats_10k_reads.txt
# SQ counters of the DP kernels
for cfg in cfg2 cfg3 cfg1; do
  rm -rf /tmp/sq
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d /tmp/sq -- python $R/bench.py --preset $cfg --steps 1 --warmup 0 $Q --e2e none > /dev/null 2>&1
  python $R/tools/pmc_sq.py $(find /tmp/sq -name "*.db" | head -1) 10000 $([ $cfg = cfg1 ] && echo 2000 || echo 10000) > $O/pmc_sq_counters_$cfg.txt 2>&1
done
cd $R
# A/B: register budget of k_dp<8> (128 = tree) on cfg2 resident and cfg4 resident + end to end
for lib in - alt_builds/lib_vgpr120.so alt_builds/lib_vgpr112.so; do
  if [ "$lib" = "-" ]; then unset TBA_LIB_PATH; else export TBA_LIB_PATH=$R/$lib; fi
  for cfg in cfg2 cfg4; do
    timeout 400 python bench.py --preset $cfg --steps 8 --warmup 1 $Q 2>$O/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$lib', 'ms_per_step %.2f' % d['ms_per_step'], 'resident %.0f' % d['value'], 'e2e', d.get('end_to_end', {}).get('value'), 'stage_ms', {k: round(v, 2) for k, v in d['config'].get('stage_ms', {}).items() if v > 0.3})
" >> $O/ab_vgpr.txt 2>&1 || tail -5 $O/ab_err.log >> $O/ab_vgpr.txt
  done
done
unset TBA_LIB_PATH
cat $O/ab_vgpr.txt
head -30 $O/kernel_stats_10k_reads.txt
cat $O/pmc_sq_counters_cfg2.txt
