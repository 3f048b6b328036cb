R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
B="--no-pmc --no-cpu-baseline --api-reads 0"
for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --preset longtail --steps 4 $B > $O/lt_q$q.json 2>/dev/null
done
GPU_MAX_HW_QUEUES=16 TBA_SIDE_STREAM_MAX_ENGINES=64 python bench.py --preset longtail --steps 4 $B > $O/lt_q16_side.json 2>/dev/null
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --preset cfg2 --steps 10 --no-pmc --no-cpu-baseline --interleaved > $O/cfg2_q$q.json 2>/dev/null
GPU_MAX_HW_QUEUES=$q python bench.py --preset cfg4 --steps 10 --no-pmc --no-cpu-baseline --interleaved > $O/cfg4_q$q.json 2>/dev/null
done
for f in $O/lt_q*.json $O/cfg2_q*.json $O/cfg4_q*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], (d['config'].get('two_resident_batches_alternating') or {}).get('reads_per_s'), (d.get('end_to_end') or {}).get('value'), {k:(v['reads_per_s'] if isinstance(v,dict) and 'reads_per_s' in v else v) for k,v in (d.get('api') or {}).items() if k.startswith('resquiggle_batch')})"; done
