# SQ / LDS / wait counters of every kernel of one preset (GPU box): bash tools/pmc_kernel_counters.sh <preset> <tag>
# -> gpurun_out/<tag>/counters_<preset>.txt (one tools/pmc_sq.py table per counter group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PRESET=${1:-cfg4}; TAG=${2:-pmc}
mkdir -p $R/gpurun_out/$TAG
OUT=$R/gpurun_out/$TAG/counters_$PRESET.txt
: > $OUT
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_k
  rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_k -- python $R/bench.py --preset $PRESET --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --e2e none --api-reads 0 > /dev/null 2>&1
  db=$(find /tmp/pmc_k -name "*.db" | head -1)
  python $R/tools/pmc_sq.py $db 10000 10000 2>&1 | grep -v "e+04 *$\|e+05 *$\|^$\|SQ_INSTS_VALU [0-9]" >> $OUT
  echo >> $OUT
done
cat $OUT
