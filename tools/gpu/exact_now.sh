cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_exact.sh xd1 xd2 | grep -v "^step 0"
python -m pytest tests -m gpu -x -q -k "exact or dropin" 2>&1 | tail -2
