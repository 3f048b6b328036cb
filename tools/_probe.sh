cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
python tools/determinism_probe.py --rna --runs 16 > $O/probe_repair.txt 2>&1
echo repair same: $(grep -c "same as" $O/probe_repair.txt); grep -v "same as" $O/probe_repair.txt | cut -c1-150 | head -6
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
