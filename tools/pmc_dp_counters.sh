# SQ counters of the DP kernels of one preset (run on the GPU box through gpurun):
#   bash tools/pmc_dp_counters.sh <preset> <tag> [lib.so]
# writes gpurun_out/<tag>/*.txt (tools/pmc_sq.py tables, k_dp rows only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PRESET=${1:-cfg2}; TAG=${2:-pmc}; LIB=${3:-}
mkdir -p $R/gpurun_out/$TAG
[ -n "$LIB" ] && export TBA_LIB_PATH=$R/$LIB
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $ctrs | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$tag -- python $R/bench.py --preset $PRESET --reads 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --e2e none --api-reads 0 > /dev/null 2>&1
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  python $R/tools/pmc_sq.py $db 4096 10000 2>&1 | grep -E "^kernel|k_dp" | grep -v "e+04\|e+05 *$" > $R/gpurun_out/$TAG/$tag.txt
  head -4 $R/gpurun_out/$TAG/$tag.txt
done
