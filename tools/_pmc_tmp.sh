cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3e
for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  tag=$(echo $ctrs | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$tag -- python $R/bench.py --bases 10000 --bandwidth 200 --reads 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --e2e none --api-reads 0 > /dev/null 2>&1
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  python $R/tools/pmc_sq.py $db 4096 10000 2>&1 | grep -E "^kernel|k_dp_multi|k_dp<4" 
done
