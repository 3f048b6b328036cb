R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
B="--no-pmc --no-cpu-baseline --e2e none --api-reads 0 --steps 20"
for i in 1 2; do
python bench.py --preset cfg4 $B > $O/cfg4_side_$i.json 2>/dev/null
TBA_NO_SIDE_STREAM=1 python bench.py --preset cfg4 $B > $O/cfg4_noside_$i.json 2>/dev/null
python bench.py --preset cfg2 $B > $O/cfg2_side_$i.json 2>/dev/null
TBA_NO_SIDE_STREAM=1 python bench.py --preset cfg2 $B > $O/cfg2_noside_$i.json 2>/dev/null
done
for f in $O/cfg*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
