R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
B="--no-pmc --no-cpu-baseline --e2e none --api-reads 0 --steps 20"
for i in 1 2; do
python bench.py --preset cfg4 $B > $O/cfg4_pipe_$i.json 2>/dev/null
python bench.py --preset cfg2 $B > $O/cfg2_pipe_$i.json 2>/dev/null
done
python bench.py --preset cfg1 $B > $O/cfg1_pipe_1.json 2>/dev/null
for f in $O/cfg*_pipe_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['stage_ms'])"; done
