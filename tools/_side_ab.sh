R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
python tools/stage_times.py --reads 10000 --libs alt_builds/lib_emit1.so,tombo_amd/libtombo_amd.so > $O/emit_ab_f64.txt 2>&1; tail -12 $O/emit_ab_f64.txt
python tools/stage_times.py --reads 10000 --dac --dtype i16 --libs alt_builds/lib_emit1.so,tombo_amd/libtombo_amd.so > $O/emit_ab_i16.txt 2>&1; tail -12 $O/emit_ab_i16.txt
B="--no-pmc --no-cpu-baseline --e2e none --api-reads 0 --steps 20"
for cfg in cfg2 cfg4 cfg3 cfg1; do
python bench.py --preset $cfg $B > $O/${cfg}_emit.json 2>$O/${cfg}_emit.err
done
for f in $O/cfg*_emit.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['two_resident_batches_alternating'], d['config']['stage_ms'])"; done
