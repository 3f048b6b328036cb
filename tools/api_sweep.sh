# the no-signal leg of the API (5 000 reads) against the number of sub-batches and of packing threads:
#   gpurun -- 'bash tools/api_sweep.sh'
nproc
for t in 8 16 32 64; do
  export TBA_PACK_THREADS=$t
  echo "== pack threads $t"; NO_SIGNAL=1 API_ONE_CALL=1 python tools/api_profile.py 5000 2>&1 | grep "wall"
done
unset TBA_PACK_THREADS
for c in 0 2 3 4 6; do
  if [ $c = 0 ]; then export TBA_API_STREAM=0; unset TBA_API_CUTS; else export TBA_API_STREAM=1; export TBA_API_CUTS=$c; fi
  echo "== cuts $c"; NO_SIGNAL=1 API_ONE_CALL=1 python tools/api_profile.py 5000 2>&1 | grep "wall"
done
