cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_lib.sh new gs th th_nogs
