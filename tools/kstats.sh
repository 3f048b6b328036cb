# Per-kernel averages of one bench preset under rocprofv3 (GPU box): bash tools/kstats.sh <preset> <out-name> [min pct]
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats -d /tmp/kst -- python $R/bench.py --preset $1 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --e2e none --api-reads 0 > $R/gpurun_out/$2.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/kst -name "*.db" | head -1) > $R/gpurun_out/$2.txt
head -${3:-24} $R/gpurun_out/$2.txt | cut -c1-120
