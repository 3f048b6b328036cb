R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
python bench.py --preset longtail --steps 4 --no-pmc --api-reads 0 --no-cpu-baseline > $O/longtail.json 2> $O/longtail.err
python bench.py --preset cfg2 --steps 10 --no-pmc --no-cpu-baseline > $O/cfg2_api.json 2> $O/cfg2_api.err
for f in $O/longtail.json $O/cfg2_api.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('end_to_end') or {}).get('value'), {k:(v['reads_per_s'] if isinstance(v,dict) and 'reads_per_s' in v else v) for k,v in (d.get('api') or {}).items() if k.startswith('resq')})"; done
