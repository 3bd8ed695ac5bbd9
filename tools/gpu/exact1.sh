cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/exact_rate.py 2>&1 | tail -12
