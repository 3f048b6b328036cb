# Round-6 measurement set (GPU box: gpurun -- 'bash tools/r06_measure.sh [quick]'); results in gpurun_out/r06/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
for cfg in cfg2 cfg3 cfg1 cfg4; do
  python bench.py --preset $cfg --steps 20 --interleaved > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
python bench.py --preset longtail --steps 4 --no-pmc --api-reads 0 > $O/bench_longtail.json 2> $O/bench_longtail.err
[ "$1" = "quick" ] && exit 0
python bench.py --gpus 2 --steps 6 > $O/bench_default_2ranks_on_1gpu.json 2> $O/bench_n2.err
python bench.py --gpus 2 --preset cfg5 --job-reads 200000 > $O/bench_cfg5_job_200k_2ranks_on_1gpu.json 2> $O/bench_cfg5_n2.err
# kernel stats of the cfg2 run (rocprofv3 --kernel-trace --stats), same command as the bench line
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r06_stats
rocprofv3 --kernel-trace --stats -d /tmp/r06_stats -- python $R/bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --e2e none --api-reads 0 > $O/bench_line_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/r06_stats -name "*.db" | head -1) > $O/kernel_stats_10k_reads.txt
rm -rf /tmp/r06_stats4
rocprofv3 --kernel-trace --stats -d /tmp/r06_stats4 -- python $R/bench.py --preset cfg4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --e2e none --api-reads 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/r06_stats4 -name "*.db" | head -1) > $O/kernel_stats_cfg4_rna.txt
# SQ counters of the DP kernels per preset
for cfg in cfg2 cfg3 cfg1; do
  rm -rf /tmp/r06_sq
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d /tmp/r06_sq -- python $R/bench.py --preset $cfg --steps 1 --warmup 0 --no-pmc --no-cpu-baseline --e2e none --api-reads 0 > /dev/null 2>&1
  python $R/tools/pmc_sq.py $(find /tmp/r06_sq -name "*.db" | head -1) 10000 $([ $cfg = cfg1 ] && echo 2000 || echo 10000) > $O/pmc_sq_counters_$cfg.txt 2>&1
done
# full-batch HBM traffic per kernel (two separate counter passes over the whole 10 000-read batch)
python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e none --api-reads 0 > $O/bench_cfg2_pmc_at_10k_reads.json 2> $O/bench_pmc10k.err
python $R/tools/kernel_gbps.py $O/bench_cfg2_pmc_at_10k_reads.json $O/kernel_stats_10k_reads.txt > $O/kernel_GBps_cfg2.txt 2>&1
cd $R
[ -x alt_builds/occupancy ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o alt_builds/occupancy tools/occupancy.hip
alt_builds/occupancy > $O/occupancy.txt 2>&1
ls -la $O
