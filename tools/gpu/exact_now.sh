cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_exact.sh xd0 xd1 | grep -v "^step 0"
python tools/exact_sweep.py 2>&1 | grep -v EXACT_JSON | tail -4 | cut -c1-420
