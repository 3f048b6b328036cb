#!/bin/bash
# bash tools/run_hunt.sh <outdir> <runs> <lib>...   tools/tb_hunt.py over several builds of the library, one result
# file and one HUNT line each ("-" = the tree's libtombo_amd.so; a name containing b2 = a -DTBA_TB_B2 build).
# HUNT_KIND: the read set (default "--rna": 10 000 x 3 kb RNA; "--bases 10000": 10 000 x 10 kb DNA)
out=$1; runs=$2; shift 2
mkdir -p $out
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = "-" ]; then unset TBA_LIB_PATH; name=tree; else export TBA_LIB_PATH=$PWD/$lib; fi
  extra=""
  case $name in *b2*) extra="$extra --aux b2";; esac
  case $name in *times*) extra="$extra --aux times";; esac
  timeout 900 python tools/tb_hunt.py ${HUNT_KIND---rna} --runs $runs --tag $name $extra > $out/hunt_$name.txt 2>&1
  tail -1 $out/hunt_$name.txt
done
