cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_exact.sh xd2 xd3 | grep -v "^step 0"
python tools/gpu/exact_pack.py 2>&1 | tail -3
python tools/exact_sweep.py 2>&1 | grep -v EXACT_JSON | tail -4 | cut -c1-420
