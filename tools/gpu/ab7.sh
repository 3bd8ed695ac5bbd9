cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/gpu/ab_lib.sh old new rowm
